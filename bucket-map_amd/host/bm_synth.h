// bm_synth.h -- seeded synthetic genomes and simulated reads (bench / test inputs).
//
// The read model follows the reference's simulator (bucket_map/tools/short_read_simulator.h:157-189,
// :100-116, :191-240) -- bucket uniform, start uniform in [0, size-L-1), Poisson numbers of deletions,
// insertions, substitutions applied in that order at uniform positions, strand flip with p = 1/2,
// qualities all 'E' -- but every draw comes from a counter-based splitmix64 stream keyed by
// (seed, read index), so the data are reproducible and can be generated in parallel
// (the reference seeds from time()/random_device).  SURVEY.md 8(d) fixes the seeds.
#pragma once

#include "bm_genome.h"

#include <thread>

namespace bm {

// i.i.d. uniform ACGT; base i of record r is a pure function of (seed, r, i).
inline Genome synth_genome(uint64_t seed, const std::vector<uint64_t> &record_lengths, unsigned n_threads = 0) {
    Genome g;
    for (size_t r = 0; r < record_lengths.size(); r++) {
        g.ids.push_back("synth" + std::to_string(r + 1) + " seed=" + std::to_string(seed) +
                        " len=" + std::to_string(record_lengths[r]));
        g.seqs.emplace_back(record_lengths[r], 'A');
    }
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    for (size_t r = 0; r < record_lengths.size(); r++) {
        std::string &s = g.seqs[r];
        const uint64_t key = splitmix64(seed ^ (0xA5A5A5A5ull + r * 0x100000001B3ull));
        const uint64_t words = (s.size() + 31) / 32;
        auto work = [&](uint64_t w0, uint64_t w1) {
            for (uint64_t w = w0; w < w1; w++) {
                uint64_t bits = splitmix64(key + w);
                const uint64_t lim = std::min<uint64_t>(s.size(), (w + 1) * 32);
                for (uint64_t i = w * 32; i < lim; i++, bits >>= 2) s[i] = dna4_char(static_cast<uint8_t>(bits & 3));
            }
        };
        std::vector<std::thread> pool;
        const unsigned nt = static_cast<unsigned>(std::min<uint64_t>(n_threads, std::max<uint64_t>(1, words / 65536)));
        for (unsigned t = 0; t < nt; t++) pool.emplace_back(work, words * t / nt, words * (t + 1) / nt);
        for (auto &t : pool) t.join();
    }
    return g;
}

struct SimReads {
    std::vector<uint8_t> bases, quals;       // concatenated, ASCII / phred+33
    std::vector<uint64_t> offsets;           // n + 1
    std::vector<uint32_t> truth_bucket;      // global bucket id the read was drawn from
    std::vector<uint32_t> truth_offset;      // start inside the bucket
    std::vector<uint8_t> truth_rc;           // 1 = emitted as reverse complement
    size_t size() const { return truth_bucket.size(); }
};

// Knuth's product method; lambda is small (<= a few hundred) in every use.
inline int poisson(Rng &rng, double lambda) {
    if (lambda <= 0) return 0;
    if (lambda > 30) {   // split to keep exp(-lambda) away from underflow
        int n = 0;
        double rest = lambda;
        while (rest > 30) { n += poisson(rng, 30.0); rest -= 30.0; }
        return n + poisson(rng, rest);
    }
    const double limit = std::exp(-lambda);
    double p = 1.0;
    int k = 0;
    do { k++; p *= rng.unit(); } while (p > limit);
    return k - 1;
}

enum QualityMode { QUAL_ALL_E = 0, QUAL_NOISY = 1 };

inline SimReads simulate_reads(const Genome &g, const std::vector<Bucket> &buckets, uint32_t read_length,
                               uint64_t n_reads, double sub_rate, double ins_rate, double del_rate,
                               uint64_t seed, QualityMode qmode = QUAL_ALL_E, unsigned n_threads = 0) {
    SimReads out;
    if (buckets.empty()) throw std::runtime_error("no buckets to simulate reads from");
    std::vector<std::string> seqs(n_reads), quals(n_reads);
    out.truth_bucket.resize(n_reads);
    out.truth_offset.resize(n_reads);
    out.truth_rc.resize(n_reads);
    auto work = [&](uint64_t r0, uint64_t r1) {
        for (uint64_t r = r0; r < r1; r++) {
            Rng rng(splitmix64(seed) + r * 0x10000ull);
            // short_read_simulator.h:164-176
            const uint32_t b = rng.below(static_cast<uint32_t>(buckets.size()));
            const Bucket &bk = buckets[b];
            const uint32_t size = bk.end - bk.start;
            uint32_t start = 0;
            if (size > read_length + 1) start = rng.below(size - read_length - 1);
            const uint32_t end = std::min(start + read_length, size);
            std::string s(g.seqs[bk.record].data() + bk.start + start, end - start);
            // :123-135 numbers of errors, :114-116 order deletions -> insertions -> substitutions
            const int subs = poisson(rng, sub_rate * read_length);
            const int dels = poisson(rng, del_rate * read_length);
            const int inss = poisson(rng, ins_rate * read_length);
            for (int i = 0; i < dels && s.size() > 1; i++) s.erase(rng.below(static_cast<uint32_t>(s.size())), 1);
            for (int i = 0; i < inss; i++)
                s.insert(s.begin() + rng.below(static_cast<uint32_t>(s.size())), dna4_char(static_cast<uint8_t>(rng.below(4))));
            for (int i = 0; i < subs; i++) {
                const uint32_t at = rng.below(static_cast<uint32_t>(s.size()));
                char nt = dna4_char(static_cast<uint8_t>(rng.below(4)));
                while (nt == s[at]) nt = dna4_char(static_cast<uint8_t>(rng.below(4)));
                s[at] = nt;
            }
            // :66-79 reverse complement with probability 1/2
            const bool rc = rng.below(2) == 1;
            if (rc) {
                std::string t(s.rbegin(), s.rend());
                for (char &c : t) c = dna4_char(static_cast<uint8_t>(3 - dna4_rank(static_cast<uint8_t>(c))));
                s.swap(t);
            }
            std::string qv(s.size(), 'E');   // :225
            if (qmode == QUAL_NOISY) {
                for (char &c : qv) {
                    // rank ~ clipped N(34, 6^2) via the sum of 4 uniforms (variance 1/3 -> scale)
                    double z = (rng.unit() + rng.unit() + rng.unit() + rng.unit() - 2.0) * std::sqrt(3.0);
                    int rank = static_cast<int>(std::lround(34.0 + 6.0 * z));
                    rank = std::max(0, std::min(41, rank));
                    c = static_cast<char>(33 + rank);
                }
            }
            seqs[r].swap(s);
            quals[r].swap(qv);
            out.truth_bucket[r] = b;
            out.truth_offset[r] = start;
            out.truth_rc[r] = rc ? 1 : 0;
        }
    };
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = static_cast<unsigned>(std::min<uint64_t>(n_threads, std::max<uint64_t>(1, n_reads / 1024)));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++) pool.emplace_back(work, n_reads * t / nt, n_reads * (t + 1) / nt);
    for (auto &t : pool) t.join();
    out.offsets.resize(n_reads + 1);
    uint64_t total = 0;
    for (uint64_t r = 0; r < n_reads; r++) { out.offsets[r] = total; total += seqs[r].size(); }
    out.offsets[n_reads] = total;
    out.bases.resize(total);
    out.quals.resize(total);
    for (uint64_t r = 0; r < n_reads; r++) {
        std::memcpy(out.bases.data() + out.offsets[r], seqs[r].data(), seqs[r].size());
        std::memcpy(out.quals.data() + out.offsets[r], quals[r].data(), quals[r].size());
    }
    return out;
}

// generate_fastq_file (short_read_simulator.h:191-240): ids "@<i>", plus the two ground-truth files
// (CIGAR column written as '*': the reference's CIGAR bookkeeping is not on the hot path).
inline void write_fastq(const SimReads &rd, const Genome &, const std::vector<Bucket> &buckets, int bucket_length,
                        const std::string &prefix) {
    std::ofstream fq(prefix + ".fastq", std::ios::binary);
    std::ofstream bgt(prefix + ".bucket_ground_truth", std::ios::binary);
    std::ofstream pgt(prefix + ".position_ground_truth", std::ios::binary);
    if (!fq || !bgt || !pgt) throw std::runtime_error("cannot write " + prefix + ".fastq");
    for (size_t r = 0; r < rd.size(); r++) {
        const size_t o = rd.offsets[r], n = rd.offsets[r + 1] - o;
        fq << '@' << r << '\n';
        fq.write(reinterpret_cast<const char *>(rd.bases.data() + o), static_cast<std::streamsize>(n));
        fq << "\n+\n";
        fq.write(reinterpret_cast<const char *>(rd.quals.data() + o), static_cast<std::streamsize>(n));
        fq << '\n';
        const Bucket &bk = buckets[rd.truth_bucket[r]];
        bgt << rd.truth_bucket[r] << ' ' << rd.truth_offset[r] << ' ' << int(rd.truth_rc[r]) << " *\n";
        pgt << bk.record << ' ' << static_cast<uint64_t>(bk.index) * bucket_length + rd.truth_offset[r] + 1 << ' '
            << int(rd.truth_rc[r]) << " *\n";
    }
}

}  // namespace bm
