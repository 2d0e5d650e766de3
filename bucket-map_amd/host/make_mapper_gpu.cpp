// make_mapper_gpu.cpp -- the product's mapper factory: the MI355X filter behind bm::mapper.
// (main.cpp:202-209: q_gram_mapper<BM_BUCKET_NUM> map(BM_BUCKET_LEN, read_len, k, q, S, fault, d, b))
#include "cli.h"
#include "gpu_q_gram_mapper.h"

#include <memory>

std::unique_ptr<bm::mapper> bm_make_mapper(const bm::cmd_arguments &args, unsigned int num_buckets, unsigned int fault) {
    return std::make_unique<bm::gpu_q_gram_mapper>(num_buckets, args.bucket_len, args.max_read_length,
                                                   args.query_seed_length, args.index_seed_length,
                                                   args.mapper_sample_size, fault,
                                                   args.mapper_distinguishability_threshold, args.average_base_quality,
                                                   30, 5, args.gpus, args.early_exit ? BMF_FLAG_EARLY_EXIT : 0u);
}
