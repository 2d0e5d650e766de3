// bm_indexer.h -- host q-gram x bucket index builder and the three index files.
//
// Restates bucket_indexer<NB> (bucket_map/indexer/bucket_indexer.h:49-127,138-216) and the
// FracMinHash row selection driven by hash_function_generator
// (bucket_map/tools/hash_function_generator.h:91-117, bucket_map/main.cpp:176-185).
// File formats: SURVEY.md Appendix B.2.  Stays on the host by design (north_star: "indexer ... stay on
// host"); it exists so that the mapper has real .qgram/.kmers_index/.bucket_id files to load.
#pragma once

#include "bm_genome.h"

#include <algorithm>
#include <cstring>
#include <filesystem>
#include <thread>

namespace bm {

// (x*i + y) % p % table_size with p = first listed prime > 10*table_size (116731 for 10000).
// The reference draws x, y from rand() seeded by time(); here they come from a seed so that index
// files are reproducible.  kmer_frac == 1 keeps every q-gram whatever x, y are.
struct FracMinHash {
    uint64_t x = 1, y = 0, p = 116731, table = 10000;
    static FracMinHash from_seed(uint64_t seed) {
        FracMinHash h;
        h.x = splitmix64(seed) % (h.p - 1) + 1;
        h.y = splitmix64(seed + 1) % h.p;
        return h;
    }
    uint64_t operator()(int input) const { return (x * static_cast<uint64_t>(input) + y) % p % table; }
};

struct QgramIndex {
    uint32_t num_buckets = 0;                // NB (bitset width)
    uint32_t q = 0;
    uint32_t row_bytes = 0;                  // (NB + 7) >> 3
    std::vector<int32_t> kmer_to_index;      // 4^q entries, -1 = not sampled
    uint64_t num_rows = 0;                   // num_valid_kmers
    std::vector<uint8_t> rows;               // num_rows x row_bytes, LSB-first bits (bucket_indexer.h:64-73)
    std::vector<std::string> bucket_id;      // one FASTA header per kept bucket
};

// bucket_indexer ctor (bucket_indexer.h:138-160): kept iff hash(i) <= threshold, rows numbered in
// ascending q-gram hash.  threshold = (unsigned)(10000 * kmer_frac) in float32 (main.cpp:185).
inline void select_qgrams(QgramIndex &ix, uint32_t q, const FracMinHash &h, float kmer_frac) {
    ix.q = q;
    const uint32_t threshold = trunc_mul_f32(kmer_frac, static_cast<uint32_t>(h.table));
    const uint64_t n = 1ull << (2 * q);
    ix.kmer_to_index.assign(n, -1);
    int32_t index = 0;
    for (uint64_t i = 0; i < n; i++)
        if (h(static_cast<int>(i)) <= threshold) ix.kmer_to_index[i] = index++;
    ix.num_rows = static_cast<uint64_t>(index);
}

// bucket_indexer::index + _insert_into_bucket (bucket_indexer.h:49-61,170-216): set bit
// (row of q-gram, bucket) for every q-gram of every kept bucket.  Threads own bucket ranges aligned to
// 8 buckets, i.e. whole bytes of every row, so no two threads touch the same byte.
inline QgramIndex build_index(const Genome &g, uint32_t num_buckets, int bucket_length, int read_length,
                              uint32_t q, const FracMinHash &h, float kmer_frac, unsigned n_threads = 0) {
    QgramIndex ix;
    ix.num_buckets = num_buckets;
    ix.row_bytes = (num_buckets + 7u) >> 3;
    select_qgrams(ix, q, h, kmer_frac);
    const std::vector<Bucket> buckets = cut_buckets(g, bucket_length, read_length);
    if (buckets.size() > num_buckets)
        throw std::runtime_error("genome cuts into " + std::to_string(buckets.size()) + " buckets, more than NB = " +
                                 std::to_string(num_buckets));
    ix.rows.assign(static_cast<size_t>(ix.num_rows) * ix.row_bytes, 0);
    for (auto &b : buckets) ix.bucket_id.push_back(g.ids[b.record]);

    const uint32_t qmask = static_cast<uint32_t>((1ull << (2 * q)) - 1ull);
    auto work = [&](size_t b0, size_t b1) {
        for (size_t b = b0; b < b1; b++) {
            const Bucket &bk = buckets[b];
            const char *s = g.seqs[bk.record].data() + bk.start;
            const uint32_t len = bk.end - bk.start;
            if (len < q) continue;
            uint8_t *col = ix.rows.data() + (b >> 3);
            const uint8_t bit = static_cast<uint8_t>(1u << (b & 7));
            uint32_t hash = 0;
            for (uint32_t i = 0; i < len; i++) {
                hash = ((hash << 2) | dna4_rank(static_cast<uint8_t>(s[i]))) & qmask;
                if (i + 1 >= q) {
                    const int32_t row = ix.kmer_to_index[hash];
                    if (row >= 0) col[static_cast<size_t>(row) * ix.row_bytes] |= bit;
                }
            }
        }
    };
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    const size_t groups = (buckets.size() + 7) / 8;
    n_threads = static_cast<unsigned>(std::min<size_t>(n_threads, std::max<size_t>(groups, 1)));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; t++) {
        const size_t g0 = groups * t / n_threads, g1 = groups * (t + 1) / n_threads;
        pool.emplace_back(work, std::min(g0 * 8, buckets.size()), std::min(g1 * 8, buckets.size()));
    }
    for (auto &t : pool) t.join();
    return ix;
}

// check_filename_in (utils.h:125-144): refuse to overwrite.
inline bool index_file_exists(const std::filesystem::path &dir, const std::string &name) {
    return std::filesystem::exists(dir / name);
}

// _store_q_gram_index / _store_bucket_ids / _store_sampled_kmers (bucket_indexer.h:76-127).
inline void write_index(const QgramIndex &ix, const std::filesystem::path &dir, const std::string &indicator) {
    std::filesystem::create_directories(dir);
    {
        std::ofstream f(dir / (indicator + ".qgram"), std::ios::binary);
        if (!f) throw std::runtime_error("cannot write " + (dir / (indicator + ".qgram")).string());
        f.write(reinterpret_cast<const char *>(ix.rows.data()), static_cast<std::streamsize>(ix.rows.size()));
    }
    {
        std::ofstream f(dir / (indicator + ".bucket_id"), std::ios::binary);
        for (auto &id : ix.bucket_id) f << id << "\n";
    }
    {
        std::ofstream f(dir / (indicator + ".kmers_index"), std::ios::binary);
        std::string buf;
        buf.reserve(ix.kmer_to_index.size() * 7);
        for (int32_t v : ix.kmer_to_index) {
            buf += std::to_string(v);
            buf.push_back('\n');
        }
        f.write(buf.data(), static_cast<std::streamsize>(buf.size()));
    }
}

}  // namespace bm
