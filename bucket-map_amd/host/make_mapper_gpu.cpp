// make_mapper_gpu.cpp -- the product's factories: the MI355X filter behind bm::mapper and the MI355X
// locator scan behind bm::offset_scanner.
// (main.cpp:202-209: q_gram_mapper<BM_BUCKET_NUM> map(BM_BUCKET_LEN, read_len, k, q, S, fault, d, b))
#include "cli.h"
#include "gpu_offset_scanner.h"
#include "gpu_q_gram_mapper.h"

#include <memory>

std::unique_ptr<bm::mapper> bm_make_mapper(const bm::cmd_arguments &args, unsigned int num_buckets, unsigned int fault) {
    return std::make_unique<bm::gpu_q_gram_mapper>(num_buckets, args.bucket_len, args.max_read_length,
                                                   args.query_seed_length, args.index_seed_length,
                                                   args.mapper_sample_size, fault,
                                                   args.mapper_distinguishability_threshold, args.average_base_quality,
                                                   30, 5, args.gpus, args.early_exit ? BMF_FLAG_EARLY_EXIT : 0u);
}

std::unique_ptr<bm::offset_scanner> bm_make_scanner(const bm::cmd_arguments &args, int allowed_mismatch, int allowed_indel) {
    return std::make_unique<bm::gpu_offset_scanner>(args.query_seed_length, static_cast<uint32_t>(args.locator_sample_size),
                                                    allowed_mismatch, allowed_indel,
                                                    args.bucket_len + args.max_read_length, args.gpus.front());
}
