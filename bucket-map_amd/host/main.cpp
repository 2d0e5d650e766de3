// main.cpp -- the `bucketmap` command-line tool (bucket_map/main.cpp:135-234), MI355X edition.
// Compiled a second time with -DBM_ALIGN it is `bucketmap_align` (bucket_map/CMakeLists.txt:138): every
// located candidate is verified by a pairwise alignment and written with MAPQ = 60 + score and a CIGAR.
//
//   bucketmap -x -i <name> --genome ref.fa                       index only (writes into the cwd)
//   bucketmap -i <name> -q reads.fq -o out.sam --genome ref.fa    map (indexes first if needed)
//
// Same flags, same stderr tags, same exit codes as the reference; the candidate-bucket filter runs on
// the GPU(s) given by --gpus (default device 0) through bm::mapper.  Which mapper sits behind the
// interface is decided by bm_make_mapper (make_mapper_gpu.cpp for the product).
#include "bucket_locator.h"
#include "cli.h"
#include "bm_indexer.h"
#include "mapper.h"

#include <iostream>
#include <memory>
#include <thread>

// Defined per build: the product links make_mapper_gpu.cpp (MI355X filter + MI355X locator scan).
std::unique_ptr<bm::mapper> bm_make_mapper(const bm::cmd_arguments &args, unsigned int num_buckets, unsigned int fault);
std::unique_ptr<bm::offset_scanner> bm_make_scanner(const bm::cmd_arguments &args, int allowed_mismatch, int allowed_indel);
// bucketmap_align only: where align_pairwise runs (the MI355X verifier in the product).
std::unique_ptr<bm::alignment_verifier> bm_make_verifier(const bm::cmd_arguments &args);
// --gpu-index: fills ix with the rows built on the device; false = not available in this build.
bool bm_gpu_index(const bm::cmd_arguments &args, const bm::Genome &genome, unsigned int num_buckets, bm::QgramIndex &ix);
// Whatever the build's devices need before their first use (HIP start-up and code objects: a third of a second), begun
// on a thread of its own while the genome is read; errors are left for the real calls to report.  main() joins it.
std::thread bm_warm_up(const bm::cmd_arguments &args);

// One [INFO] line when the run is done: device memory in use (the contexts are still alive) and the host's peak RSS.
void bm_report_resources(const bm::cmd_arguments &args);

int main(int argc, char **argv) {
    bm::cmd_arguments args;
    try {
        args = bm::parse_arguments(argc, argv);
    } catch (bm::parser_error const &ext) {
        std::cerr << "[ERROR]\t\t" << ext.what() << "\n";   // main.cpp:148-155
        return -1;
    }
#ifdef BM_ALIGN
    std::cerr << "[INFO]\t\tAllowing Smith-Waterman for alignment verifications.\n";   // main.cpp:159-163
#else
    std::cerr << "[INFO]\t\tNot using Smith-Waterman for alignment verifications.\n";
#endif
    if (args.genome_path.empty()) {
        // the reference prints this when BM_* are not defined at build time (main.cpp:227-231)
        std::cerr << "[ERROR]\t\tThe definition of BM_BUCKET_NUM, BM_BUCKET_LEN or BM_GENOME_FILE is not found. "
                     "Pass --genome (and optionally --bucket-len / --num-buckets).\n";
        return -1;
    }
    struct joiner {
        std::thread t;
        ~joiner() {
            if (t.joinable()) t.join();
        }
    } warm{bm_warm_up(args)};
    try {
        auto t_fa = std::chrono::steady_clock::now();
        bm::Genome genome = bm::read_fasta(args.genome_path.string());
        std::cerr << "[BENCHMARK]\tElapsed time for reading the reference genome: "
                  << std::chrono::duration<float>(std::chrono::steady_clock::now() - t_fa).count() << " s.\n";
        const unsigned int num_buckets = args.num_buckets ? args.num_buckets : bm::awk_bucket_num(genome, args.bucket_len);
        std::cerr << "[INFO]\t\tInitializing indexer and mapper with bucket length: " << args.bucket_len
                  << ", and number of buckets: " << num_buckets << ".\n";
        const std::filesystem::path cwd = std::filesystem::current_path();

        // locator::initialize -> indexer::index (locator.h:33-34, bucket_indexer.h:170-216): build the
        // index files unless they exist already (the reference then prints an error and carries on).
        auto run_indexer = [&]() {
            if (bm::index_file_exists(cwd, args.index_indicator + ".qgram")) {
                std::cerr << "[ERROR]\t\tThe specified file already exists in directory: "
                          << (cwd / (args.index_indicator + ".qgram")) << ".\n";
                return;
            }
            std::cerr << "[INFO]\t\tSet index seed length to be: " << static_cast<int>(args.index_seed_length) << ".\n";
            auto t0 = std::chrono::steady_clock::now();
            // indexing is a host job in the reference; here the same rows are built in HBM (bmf_build_index,
            // byte-identical) and copied back for the files wherever the build's factory offers it and the
            // seed length is one its kernels cover; --host-index / --gpu-index force either way
            bm::QgramIndex ix;
            const bool on_device = args.gpu_index == 1 ||
                                   (args.gpu_index < 0 && args.index_seed_length >= 3 && args.index_seed_length <= 10);
            if (!(on_device && bm_gpu_index(args, genome, num_buckets, ix)))
                ix = bm::build_index(genome, num_buckets, static_cast<int>(args.bucket_len),
                                     static_cast<int>(args.max_read_length), args.index_seed_length,
                                     bm::FracMinHash::from_seed(args.hash_seed), args.frac_min_hash, args.host_threads);
            std::cerr << "[INFO]\t\tNumber of remaining k-mers after FracMinHash is " << ix.num_rows << " out of "
                      << ix.kmer_to_index.size() << " (" << static_cast<float>(ix.num_rows) / ix.kmer_to_index.size() * 100
                      << "%).\n";
            bm::write_index(ix, cwd, args.index_indicator);
            std::cerr << "[INFO]\t\tThe bucket q-gram index is stored in: " << (cwd / (args.index_indicator + ".qgram")) << ".\n";
            std::cerr << "[INFO]\t\tThe number of buckets: " << ix.bucket_id.size() << ".\n";
            std::cerr << "[INFO]\t\tThe bucket ids are stored in: " << (cwd / (args.index_indicator + ".bucket_id")) << ".\n";
            std::cerr << "[INFO]\t\tThe kmer indexes are stored in: " << (cwd / (args.index_indicator + ".kmers_index")) << ".\n";
            std::cerr << "[BENCHMARK]\tElapsed time for creating and storing index files: "
                      << std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count() << " s.\n";
        };

        if (args.only_indexer) {   // main.cpp:186-189
            run_indexer();
            return 0;
        }
        if (args.output_sam_path.empty()) {
            std::cerr << "[ERROR]\t\tThe output sam file is not set. Please set the output path using '-o' option.\n";
            return 1;
        }
        if (args.query_seed_length < args.index_seed_length) {
            std::cerr << "[ERROR]\t\tThe query seed length (currently set to " << static_cast<int>(args.query_seed_length)
                      << ") should be larger thanthe index seed length (currently set to "
                      << static_cast<int>(args.index_seed_length) << ").\n";
            return 1;
        }

        // main.cpp:202-209: fault = ceil(S * e) with a float32 product
        const unsigned int fault = bm::ceil_mul_f32(args.allowed_seed_miss_rate, args.mapper_sample_size);
        std::unique_ptr<bm::mapper> map = bm_make_mapper(args, num_buckets, fault);
        // main.cpp:211-218 (the locator receives -b, not -u, as its quality threshold)
        const unsigned int locator_samples = static_cast<unsigned int>(args.locator_sample_size);
        std::unique_ptr<bm::offset_scanner> scanner =
            bm_make_scanner(args, static_cast<int>(bm::ceil_mul_f32(args.allowed_seed_miss_rate, locator_samples)),   // bucket_locator.h:419
                            static_cast<int>(bm::ceil_mul_f32(args.locator_allowed_indel_rate, args.max_read_length))); // :420
        bm::bucket_locator loc(map.get(), scanner.get(), args.bucket_len, args.max_read_length, args.query_seed_length,
                               args.allowed_seed_miss_rate, args.locator_allowed_indel_rate,
                               static_cast<unsigned int>(args.locator_sample_size), args.average_base_quality);
#ifdef BM_ALIGN
        std::unique_ptr<bm::alignment_verifier> verifier = bm_make_verifier(args);
        loc.set_verifier(verifier.get());
#endif
        run_indexer();
        loc.initialize(genome, cwd, args.index_indicator);                                    // main.cpp:221
        loc.locate(args.fastq_path.string(), cwd / (args.index_indicator + ".bucket_id"),     // main.cpp:224
                   args.output_sam_path, args.locator_quality_threshold);
        bm_report_resources(args);
    } catch (const std::exception &e) {
        std::cerr << "[ERROR]\t\t" << e.what() << "\n";
        return 2;
    }
    return 0;
}
