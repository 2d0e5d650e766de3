// mapper_test.cpp -- the reference's mapper benchmark (bucket_map/mapper_test.cpp:10-68) on the MI355X filter:
// load an index, _query_file a FASTQ file of simulated reads, _check_ground_truth against the simulator's
// .bucket_ground_truth file.  The reference hard-codes its paths and parameters; here they are the `bucketmap`
// flags plus --ground-truth.
//
//   mapper_test -i <name> --genome ref.fa -q reads.fastq --ground-truth reads.bucket_ground_truth [-r 150 ...]
#include "bm_indexer.h"
#include "cli.h"
#include "gpu_q_gram_mapper.h"

#include <iostream>

int main(int argc, char **argv) {
    std::string truth;
    std::vector<char *> rest{argv[0]};
    for (int i = 1; i < argc; i++) {
        if (std::string(argv[i]) == "--ground-truth" && i + 1 < argc) truth = argv[++i];
        else rest.push_back(argv[i]);
    }
    try {
        bm::cmd_arguments args = bm::parse_arguments(static_cast<int>(rest.size()), rest.data());
        if (args.genome_path.empty() || args.fastq_path.empty() || truth.empty()) {
            std::cerr << "[ERROR]\t\tusage: mapper_test -i <name> --genome ref.fa -q reads.fastq --ground-truth <file> [bucketmap flags]\n";
            return 1;
        }
        const bm::Genome genome = bm::read_fasta(args.genome_path.string());
        const unsigned int num_buckets = args.num_buckets ? args.num_buckets : bm::awk_bucket_num(genome, args.bucket_len);
        const unsigned int fault = bm::ceil_mul_f32(args.allowed_seed_miss_rate, args.mapper_sample_size);   // mapper_test.cpp:50
        bm::gpu_q_gram_mapper map(num_buckets, args.bucket_len, args.max_read_length, args.query_seed_length, args.index_seed_length,
                                  args.mapper_sample_size, fault, args.mapper_distinguishability_threshold, args.average_base_quality,
                                  30, 5, args.gpus, args.early_exit ? BMF_FLAG_EARLY_EXIT : 0u);
        map.load(std::filesystem::current_path(), args.index_indicator);                                      // :55
        const auto res = map._query_file(args.fastq_path);                                                    // :61
        map._check_ground_truth(res, truth);                                                                  // :63
    } catch (const std::exception &e) {
        std::cerr << "[ERROR]\t\t" << e.what() << "\n";
        return 2;
    }
    return 0;
}
