// make_mapper_gpu.cpp -- the product's factories: the MI355X filter behind bm::mapper and the MI355X
// locator scan behind bm::offset_scanner.
// (main.cpp:202-209: q_gram_mapper<BM_BUCKET_NUM> map(BM_BUCKET_LEN, read_len, k, q, S, fault, d, b))
#include "bm_indexer.h"
#include "cli.h"
#include "gpu_alignment_verifier.h"
#include "gpu_offset_scanner.h"
#include "gpu_q_gram_mapper.h"

#include <sys/resource.h>

#include <memory>
#include <thread>

// HIP's start-up and the loading of the library's code objects take 0.2-0.4 s in a fresh process: a throw-away filter
// context on every device of --gpus pays for them while main() reads the FASTA file (tools/init_probe.py).
std::thread bm_warm_up(const bm::cmd_arguments &args) {
    std::vector<int> devices = args.gpus.empty() ? std::vector<int>{0} : args.gpus;
    std::sort(devices.begin(), devices.end());
    devices.erase(std::unique(devices.begin(), devices.end()), devices.end());
    return std::thread([devices] {
        for (int d : devices) {
            bmf_params p{};
            p.num_buckets = 64;
            p.q = 4;
            p.k = 4;
            p.num_samples = 1;
            p.num_fault = 1;
            p.max_candidates = 30;
            p.read_len = 32;
            p.num_segment_samples = 5;
            p.device = d;
            bmf_ctx *ctx = nullptr;
            if (bmf_create(&p, &ctx) == BMF_OK) bmf_destroy(ctx);
        }
    });
}

std::unique_ptr<bm::mapper> bm_make_mapper(const bm::cmd_arguments &args, unsigned int num_buckets, unsigned int fault) {
    return std::make_unique<bm::gpu_q_gram_mapper>(num_buckets, args.bucket_len, args.max_read_length,
                                                   args.query_seed_length, args.index_seed_length,
                                                   args.mapper_sample_size, fault,
                                                   args.mapper_distinguishability_threshold, args.average_base_quality,
                                                   30, 5, args.gpus, args.early_exit ? BMF_FLAG_EARLY_EXIT : 0u);
}

// --gpu-index: bucket_indexer::index on the device (bmf_build_index), rows copied back for the files.
bool bm_gpu_index(const bm::cmd_arguments &args, const bm::Genome &genome, unsigned int num_buckets, bm::QgramIndex &ix) {
    ix = bm::QgramIndex();
    ix.num_buckets = num_buckets;
    ix.row_bytes = (num_buckets + 7u) >> 3;
    bm::select_qgrams(ix, args.index_seed_length, bm::FracMinHash::from_seed(args.hash_seed), args.frac_min_hash);
    const std::vector<bm::Bucket> buckets = bm::cut_buckets(genome, static_cast<int>(args.bucket_len), static_cast<int>(args.max_read_length));
    std::vector<uint64_t> rec_off(genome.seqs.size() + 1, 0);
    for (size_t r = 0; r < genome.seqs.size(); r++) rec_off[r + 1] = rec_off[r] + genome.seqs[r].size();
    std::vector<uint8_t> flat(rec_off.back());
    for (size_t r = 0; r < genome.seqs.size(); r++)
        std::copy(genome.seqs[r].begin(), genome.seqs[r].end(), flat.begin() + static_cast<std::ptrdiff_t>(rec_off[r]));
    std::vector<uint64_t> bstart(buckets.size());
    std::vector<uint32_t> blen(buckets.size());
    for (size_t b = 0; b < buckets.size(); b++) {
        bstart[b] = rec_off[buckets[b].record] + buckets[b].start;
        blen[b] = buckets[b].end - buckets[b].start;
        ix.bucket_id.push_back(genome.ids[buckets[b].record]);
    }
    bmf_params p{};
    p.num_buckets = num_buckets;
    p.q = args.index_seed_length;
    // the index rows depend on q and the buckets only: the mapper's own limits (read_len, k - q + 1 ...) must not
    // decide whether an index can be BUILT, so the build context asks for the smallest mapper there is
    p.k = args.index_seed_length;
    p.num_samples = 1;
    p.num_fault = 1;
    p.max_candidates = 1;
    p.read_len = args.index_seed_length;
    p.num_segment_samples = 5;
    p.device = args.gpus.front();
    bmf_ctx *c = nullptr;
    int rc = bmf_create(&p, &c);
    if (rc == BMF_ERR_UNSUPPORTED || rc == BMF_ERR_ARG) return false;   // not a geometry the device build covers: host indexer
    if (rc != BMF_OK) throw std::runtime_error(std::string("--gpu-index: ") + bmf_last_error());
    rc = bmf_build_index(c, flat.data(), flat.size(), bstart.data(), blen.data(), static_cast<uint32_t>(buckets.size()),
                         ix.kmer_to_index.data(), ix.kmer_to_index.size());
    if (rc == BMF_ERR_UNSUPPORTED) {
        bmf_destroy(c);
        return false;
    }
    if (rc == BMF_OK) {
        ix.rows.assign(static_cast<size_t>(ix.num_rows) * ix.row_bytes, 0);
        rc = bmf_index_download(c, ix.rows.data(), nullptr);
    }
    const std::string why = rc == BMF_OK ? "" : bmf_last_error();
    bmf_destroy(c);
    if (rc != BMF_OK) throw std::runtime_error("--gpu-index: " + why);
    return true;
}

std::unique_ptr<bm::offset_scanner> bm_make_scanner(const bm::cmd_arguments &args, int allowed_mismatch, int allowed_indel) {
    return std::make_unique<bm::gpu_offset_scanner>(args.query_seed_length, static_cast<uint32_t>(args.locator_sample_size),
                                                    allowed_mismatch, allowed_indel,
                                                    args.bucket_len + args.max_read_length, args.gpus);
}

std::unique_ptr<bm::alignment_verifier> bm_make_verifier(const bm::cmd_arguments &args) {
    return std::make_unique<bm::gpu_alignment_verifier>(args.gpus);
}

void bm_report_resources(const bm::cmd_arguments &args) {
    struct rusage ru {};
    getrusage(RUSAGE_SELF, &ru);
    std::vector<int> seen;
    for (int dev : args.gpus) {
        if (std::find(seen.begin(), seen.end(), dev) != seen.end()) continue;
        seen.push_back(dev);
        uint64_t free_b = 0, total_b = 0;
        if (bmf_device_memory(dev, &free_b, &total_b) != BMF_OK) continue;
        std::cerr << "[INFO]\t\tDevice " << dev << ": " << static_cast<double>(total_b - free_b) / (1u << 30) << " GiB of "
                  << static_cast<double>(total_b) / (1u << 30) << " GiB in use at the end of the run.\n";
    }
    std::cerr << "[INFO]\t\tHost peak resident set: " << static_cast<double>(ru.ru_maxrss) / (1u << 20) << " GiB.\n";
}
