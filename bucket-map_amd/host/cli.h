// cli.h -- command-line surface of `bucketmap` (bucket_map/main.cpp:12-123), without Sharg.
//
// Same short/long option names and defaults as the reference.  The three values the reference bakes in
// at compile time through CMake (BM_GENOME_PATH, BM_BUCKET_LEN, BM_BUCKET_NUM; CMakeLists.txt:7-58,
// main.cpp:170) are run-time options here (--genome, --bucket-len, --num-buckets), still honouring the
// same -DBM_* definitions as defaults when the tool is built with them.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <filesystem>
#include <stdexcept>
#include <string>
#include <vector>

namespace bm {

struct cmd_arguments {
    bool only_indexer = false;
    std::filesystem::path fastq_path{};
    std::string index_indicator;
    std::filesystem::path output_sam_path{};
    uint8_t query_seed_length = 12;
    uint8_t index_seed_length = 9;
    unsigned int max_read_length = 300;
    unsigned int mapper_sample_size = 15;
    float mapper_distinguishability_threshold = 0.5f;
    unsigned int average_base_quality = 25;
    float allowed_seed_miss_rate = 0.4f;
    float locator_allowed_indel_rate = 0.02f;
    float locator_sample_size = 10;
    unsigned int locator_quality_threshold = 40;
    float frac_min_hash = 0.25f;
    // run-time replacements of the compile-time configuration
#ifdef BM_GENOME_PATH
    std::filesystem::path genome_path = BM_GENOME_PATH;
#else
    std::filesystem::path genome_path{};
#endif
#ifdef BM_BUCKET_LEN
    unsigned int bucket_len = BM_BUCKET_LEN;
#else
    unsigned int bucket_len = 65536;
#endif
#ifdef BM_BUCKET_NUM
    unsigned int num_buckets = BM_BUCKET_NUM;
#else
    unsigned int num_buckets = 0;   // 0 = the CMake awk rule applied to --genome
#endif
    std::vector<int> gpus{0};
    uint64_t hash_seed = 20240004;
    unsigned int host_threads = 0;
    bool early_exit = true;    // BMF_FLAG_EARLY_EXIT (identical output, fewer index rows read); --no-early-exit
                               // makes the filter read every row the reference reads
    int gpu_index = -1;        // index rows built on the device (identical files): 1 = --gpu-index, 0 = --host-index,
                               // -1 = on the device where its kernels cover the seed length (3 <= -k <= 10)
};

struct parser_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline std::vector<int> parse_gpu_list(const std::string &v) {
    std::vector<int> out;
    if (v.find(',') == std::string::npos && !v.empty()) {
        // a single number N means devices 0..N-1 when prefixed with 'n', otherwise device id
        if (v[0] == 'n') {
            for (int i = 0; i < std::atoi(v.c_str() + 1); i++) out.push_back(i);
            return out;
        }
    }
    size_t pos = 0;
    while (pos <= v.size()) {
        size_t c = v.find(',', pos);
        if (c == std::string::npos) c = v.size();
        if (c > pos) out.push_back(std::atoi(v.substr(pos, c - pos).c_str()));
        pos = c + 1;
    }
    if (out.empty()) throw parser_error("--gpus needs a device list such as 0,1,2,3 or n8");
    return out;
}

// Parses argv like sharg does for these options: `-k 9`, `--index-seed 9`, `--index-seed=9`.
inline cmd_arguments parse_arguments(int argc, char **argv) {
    cmd_arguments a;
    bool have_indicator = false;
    auto has_ext = [](const std::filesystem::path &p, std::initializer_list<const char *> exts) {
        std::string e = p.extension().string();
        if (!e.empty()) e.erase(0, 1);
        for (auto x : exts)
            if (e == x) return true;
        return false;
    };
    for (int i = 1; i < argc; i++) {
        std::string opt = argv[i], val;
        bool inline_val = false;
        if (opt.rfind("--", 0) == 0) {
            size_t eq = opt.find('=');
            if (eq != std::string::npos) {
                val = opt.substr(eq + 1);
                opt = opt.substr(0, eq);
                inline_val = true;
            }
        }
        auto value = [&]() -> std::string {
            if (inline_val) return val;
            if (i + 1 >= argc) throw parser_error("Missing value for option " + opt + ".");
            return argv[++i];
        };
        auto as_uint = [&](const std::string &s) -> unsigned long {
            char *end = nullptr;
            unsigned long v = std::strtoul(s.c_str(), &end, 10);
            if (s.empty() || *end) throw parser_error("Value parse failed for " + opt + ": Argument " + s + " could not be parsed as a number.");
            return v;
        };
        auto as_float = [&](const std::string &s) -> float {
            char *end = nullptr;
            float v = std::strtof(s.c_str(), &end);
            if (s.empty() || *end) throw parser_error("Value parse failed for " + opt + ": Argument " + s + " could not be parsed as a number.");
            return v;
        };
        if (opt == "-x" || opt == "--run-index") a.only_indexer = true;
        else if (opt == "-q" || opt == "--query-file") {
            a.fastq_path = value();
            if (!has_ext(a.fastq_path, {"fq", "fastq"}))
                throw parser_error("Validation failed for option -q/--query-file: Expected one of the following valid extensions: [fq, fastq]!");
            if (!std::filesystem::exists(a.fastq_path))
                throw parser_error("Validation failed for option -q/--query-file: The file " + a.fastq_path.string() + " does not exist!");
        } else if (opt == "-i" || opt == "--index-indicator") { a.index_indicator = value(); have_indicator = true; }
        else if (opt == "-o" || opt == "--output-file") {
            a.output_sam_path = value();
            if (!has_ext(a.output_sam_path, {"sam"}))
                throw parser_error("Validation failed for option -o/--output-file: Expected one of the following valid extensions: [sam]!");
            if (std::filesystem::exists(a.output_sam_path))
                throw parser_error("Validation failed for option -o/--output-file: The file " + a.output_sam_path.string() + " already exists!");
        } else if (opt == "-k" || opt == "--index-seed") a.index_seed_length = static_cast<uint8_t>(as_uint(value()));
        else if (opt == "-b" || opt == "--average-base-quality") a.average_base_quality = static_cast<unsigned>(as_uint(value()));
        else if (opt == "-l" || opt == "--query-seed") a.query_seed_length = static_cast<uint8_t>(as_uint(value()));
        else if (opt == "-r" || opt == "--read-len") a.max_read_length = static_cast<unsigned>(as_uint(value()));
        else if (opt == "-s" || opt == "--mapper-samples") a.mapper_sample_size = static_cast<unsigned>(as_uint(value()));
        else if (opt == "-d" || opt == "--distinguishability") a.mapper_distinguishability_threshold = as_float(value());
        else if (opt == "-e" || opt == "--max-error-rate") a.allowed_seed_miss_rate = as_float(value());
        else if (opt == "-n" || opt == "--max-indel-rate") a.locator_allowed_indel_rate = as_float(value());
        else if (opt == "-p" || opt == "--locator-samples") a.locator_sample_size = as_float(value());
        else if (opt == "-u" || opt == "--quality") a.locator_quality_threshold = static_cast<unsigned>(as_uint(value()));
        else if (opt == "-f" || opt == "--kmer-frac") a.frac_min_hash = as_float(value());
        else if (opt == "--version-check") (void)value();   // Sharg built-in used by the benchmark scripts
        else if (opt == "--genome") a.genome_path = value();
        else if (opt == "--bucket-len") a.bucket_len = static_cast<unsigned>(as_uint(value()));
        else if (opt == "--num-buckets") a.num_buckets = static_cast<unsigned>(as_uint(value()));
        else if (opt == "--gpus") a.gpus = parse_gpu_list(value());
        else if (opt == "--hash-seed") a.hash_seed = as_uint(value());
        else if (opt == "--early-exit") a.early_exit = true;
        else if (opt == "--no-early-exit") a.early_exit = false;
        else if (opt == "--gpu-index") a.gpu_index = 1;
        else if (opt == "--host-index") a.gpu_index = 0;
        else if (opt == "--threads") a.host_threads = static_cast<unsigned>(as_uint(value()));
        else throw parser_error("Unknown option " + opt + ". In case this is meant to be a non-option/argument/parameter, please specify the start of non-options with '--'.");
    }
    if (!have_indicator) throw parser_error("Option -i/--index-indicator is required but not set.");
    return a;
}

}  // namespace bm
