// ref_binding_main.cpp -- TEST DRIVER for integration/gpu_q_gram_mapper.h: one translation unit made of
//   * the two typedefs of /root/reference/bucket_map/utils.h:309-311 (utils.h itself needs SeqAn3, absent here),
//   * the reference's REAL bucket_map/mapper/mapper.h (found through -I/root/reference/bucket_map),
//   * the binding class,
// used through a `mapper*` the way bucket_locator uses its `_m` (bucket_locator.h:449 load, :624 map, :627 reset,
// :642 num_records).  Prints every per-bucket list so that tests can compare it with the C-ABI results.
//   ref_binding <NB> <bucket_len> <read_len> <k> <q> <samples> <fault> <index_dir> <indicator> <reads.fastq>
#include <filesystem>
#include <string>
#include <utility>
#include <vector>

typedef std::pair<unsigned int, int> segment_info_t;            // utils.h:309
typedef std::vector<std::vector<segment_info_t>> segments_t;    // utils.h:311

#include "mapper/mapper.h"
#include "gpu_q_gram_mapper.h"

#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv) {
    if (argc != 11) {
        std::fprintf(stderr, "usage: %s NB bucket_len read_len k q samples fault index_dir indicator reads.fastq\n", argv[0]);
        return 2;
    }
    auto u = [&](int i) { return static_cast<unsigned int>(std::strtoul(argv[i], nullptr, 10)); };
    try {
        gpu_q_gram_mapper map(u(1), u(2), u(3), static_cast<uint8_t>(u(4)), static_cast<uint8_t>(u(5)), u(6), u(7), 0.5f);
        mapper *_m = &map;
        _m->load(argv[8], argv[9]);
        auto [sequence_ids_orig, sequence_ids_rev_comp] = _m->map(argv[10]);
        _m->reset();
        std::printf("num_records %u\n", _m->num_records);
        for (std::size_t b = 0; b < sequence_ids_orig.size(); b++)
            for (auto &s : sequence_ids_orig[b]) std::printf("o %zu %u %d\n", b, s.first, s.second);
        for (std::size_t b = 0; b < sequence_ids_rev_comp.size(); b++)
            for (auto &s : sequence_ids_rev_comp[b]) std::printf("r %zu %u %d\n", b, s.first, s.second);
        // a second map() after reset(): the index is gone, every list must be empty (q_gram_mapper.h:389-393)
        auto again = _m->map(argv[10]);
        std::size_t left = 0;
        for (auto &v : again.first) left += v.size();
        for (auto &v : again.second) left += v.size();
        std::printf("after_reset %zu\n", left);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[ERROR]\t\t%s\n", e.what());
        return 1;
    }
    return 0;
}
