// bm_genome.h -- reference genome in memory, FASTA/FASTQ I/O and bucket cutting (host plumbing).
//
// Restates, without SeqAn3: seqan3::sequence_file_input on FASTA/FASTQ and
// iterate_through_buckets (bucket_map/utils.h:60-102).
#pragma once

#include "bm_common.h"

#include <array>
#include <cstdio>
#include <fstream>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace bm {

struct Genome {
    std::vector<std::string> ids;      // full FASTA header lines without '>'
    std::vector<std::string> seqs;     // ACGT only (folded like dna5 -> dna4: everything else -> A)
    uint64_t total_length() const {
        uint64_t t = 0;
        for (auto &s : seqs) t += s.size();
        return t;
    }
};

// FASTA is read as dna5 and converted to dna4 (utils.h:70,91): ambiguity codes become N, then A.
inline char fold_genome_char(char c) {
    switch (c) {
    case 'A': case 'a': return 'A';
    case 'C': case 'c': return 'C';
    case 'G': case 'g': return 'G';
    case 'T': case 't': case 'U': case 'u': return 'T';
    default: return 'A';
    }
}

inline Genome read_fasta(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open FASTA file " + path);
    Genome g;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            g.ids.push_back(line.substr(1));
            g.seqs.emplace_back();
        } else {
            if (g.seqs.empty()) throw std::runtime_error("FASTA file " + path + " does not start with '>'");
            // append the line, then fold it in place through a 256-entry table (white space inside a
            // sequence line is rare: only then fall back to filtering character by character)
            static const auto fold = [] {
                std::array<char, 256> t{};
                for (int c = 0; c < 256; c++) t[static_cast<size_t>(c)] = fold_genome_char(static_cast<char>(c));
                return t;
            }();
            std::string &s = g.seqs.back();
            const size_t at = s.size();
            if (line.find_first_of(" \t") == std::string::npos) {
                s.append(line);
            } else {
                for (char c : line)
                    if (c != ' ' && c != '\t') s.push_back(c);
            }
            for (size_t i = at; i < s.size(); i++) s[i] = fold[static_cast<unsigned char>(s[i])];
        }
    }
    return g;
}

inline void write_fasta(const Genome &g, const std::string &path, size_t width = 60) {
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error("cannot write FASTA file " + path);
    for (size_t r = 0; r < g.ids.size(); r++) {
        out << '>' << g.ids[r] << '\n';
        const std::string &s = g.seqs[r];
        for (size_t i = 0; i < s.size(); i += width) {
            out.write(s.data() + i, std::min(width, s.size() - i));
            out.put('\n');
        }
    }
}

struct FastqRecord {
    std::string id;      // header line without '@', untruncated
    std::string seq;     // as in the file (ASCII); dna4 folding happens where it is hashed
    std::string qual;    // phred+33
};

// Calls op for every record of a 4-line FASTQ file.
inline void for_each_fastq(const std::string &path, const std::function<void(const FastqRecord &)> &op) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open FASTQ file " + path);
    FastqRecord rec;
    std::string plus;
    auto chomp = [](std::string &s) { if (!s.empty() && s.back() == '\r') s.pop_back(); };
    while (std::getline(in, rec.id)) {
        chomp(rec.id);
        if (rec.id.empty()) continue;
        if (rec.id[0] != '@') throw std::runtime_error("FASTQ record does not start with '@' in " + path);
        rec.id.erase(0, 1);
        if (!std::getline(in, rec.seq) || !std::getline(in, plus) || !std::getline(in, rec.qual))
            throw std::runtime_error("truncated FASTQ record in " + path);
        chomp(rec.seq);
        chomp(rec.qual);
        if (rec.seq.size() != rec.qual.size())
            throw std::runtime_error("sequence and quality lengths differ in " + path);
        op(rec);
    }
}

// One kept bucket of iterate_through_buckets (utils.h:72-97).
struct Bucket {
    uint32_t record;     // index of the FASTA record
    uint32_t index;      // i: position of the bucket inside the record (kept AND skipped count)
    uint32_t start, end; // [start, end) in the record; end - start > read_len
};

// utils.h:72-97.  n = ceil(float(len)/bucket_len) in FLOAT32; bucket i = [i*bl, min(i*bl+bl+rl, len));
// buckets with size <= read_len are skipped; global id = running count of kept buckets.
inline std::vector<Bucket> cut_buckets(const Genome &g, int bucket_length, int read_length) {
    std::vector<Bucket> out;
    for (size_t r = 0; r < g.seqs.size(); r++) {
        const int size = static_cast<int>(g.seqs[r].size());
        volatile float total_length = static_cast<float>(size);
        volatile float quot = total_length / static_cast<float>(bucket_length);
        const int num_buckets = static_cast<int>(std::ceil(static_cast<double>(quot)));
        for (int i = 0; i < num_buckets; i++) {
            int start = i * bucket_length;
            int end = start + bucket_length + read_length;
            if (end > size) end = size;
            if (end - start <= read_length) continue;
            out.push_back(Bucket{static_cast<uint32_t>(r), static_cast<uint32_t>(i), static_cast<uint32_t>(start),
                                 static_cast<uint32_t>(end)});
        }
    }
    return out;
}

// The compile-time BM_BUCKET_NUM of the reference: sum over records of ceil(len/bucket_len) in awk
// double arithmetic (bucket_map/CMakeLists.txt:13-46).  Can exceed the number of kept buckets.
inline uint32_t awk_bucket_num(const Genome &g, uint32_t bucket_length) {
    uint64_t n = 0;
    for (auto &s : g.seqs)
        if (!s.empty()) {
            double v = static_cast<double>(s.size()) / bucket_length;
            n += (v == std::floor(v)) ? static_cast<uint64_t>(v) : static_cast<uint64_t>(v) + 1;
        }
    return static_cast<uint32_t>(n);
}

}  // namespace bm
