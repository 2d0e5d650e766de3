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

#include <algorithm>
#include <array>
#include <atomic>
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

// ---------------------------------------------------------------------------------------------------------
// A genome-LIKE genome.  i.i.d. uniform bases give every q-gram row of the index the same density, which makes
// the filter's data-dependent branches trivial (every row passes the distinguishability threshold, no read sits
// in a repeat, the pruning model meets exactly the density it assumed).  Real genomes are skewed
// (bucket_map/benchmark/short_read/log/bucketmap_3_map.log:8,12-14: 95.8 % of the rows pass the threshold, 94.9 %
// of the reads get a candidate), and no real genome is available offline, so this generator composes the
// features that produce that skew, every draw keyed by (seed, record, chunk) so that the result does not depend
// on the number of threads:
//   * base layer: an order-6 Markov chain whose transition weights are log-normally spread around the
//     isochore's base composition, with CpG depletion and a homopolymer bias; eight GC levels (35 .. 52 %) that
//     change smoothly along a record; every unique segment is written on a random strand, so q-gram and
//     reverse-complement q-gram are equally frequent (Chargaff's second rule);
//   * interspersed repeats: a short SINE-like family with an A tail, a long LINE-like family whose copies are
//     3'-anchored truncations, and a Zipf-weighted tail of other families; each copy carries its own 1 .. 20 %
//     divergence (substitutions and short indels) and strand;
//   * tandem repeats / low complexity: motifs of 1 .. 6 bases (and a few minisatellites) repeated with 2 % noise;
//   * satellite arrays: whole 512-Kbp and 4-Mbp regions filled with 1.5 %-diverged copies of one monomer (171, 68, 42
//     or 5 bases), the centromere-like stretches whose reads meet more than 30 buckets;
//   * segmental duplications: a 64-Kbp chunk that is a 0.5 .. 3 % diverged copy of another chunk of its record;
//   * assembly gaps: runs of N (dna5 -> dna4 folds them to A, utils.h:70,91: in memory they ARE runs of A;
//     Genome::gaps remembers them so that write_fasta can print N).
struct SkewProfile {
    unsigned order = 6;              // Markov order of the base layer
    double sigma = 0.34;             // spread (natural log units) of the transition weights
    double cpg = 0.22;               // weight factor of G after C
    double homopolymer = 1.35;       // weight factor of a third equal base in a row
    double gc_lo = 0.35, gc_hi = 0.52;
    unsigned levels = 8;
    unsigned families = 48;          // interspersed repeat families
    double p_repeat = 0.46, p_tandem = 0.10, p_gap = 0.0004;   // feature after a unique segment
    double mean_unique = 1400.0;     // mean length of a unique segment
    double p_young = 0.12;           // share of a family's copies that are young (1 .. 3 % divergence)
    double p_satellite = 0.012;      // a 512-Kbp region (and, independently, a 4-Mbp region) is one satellite array
    double p_segdup = 0.03;          // a 64-Kbp chunk is a 0.5 .. 3 % diverged copy of another chunk of its record
};

namespace skew_detail {

constexpr uint32_t kChunk = 65536;

struct Family {
    std::string consensus;
    double weight, divergence;       // share of the copies, base divergence of a copy
    bool anchored3;                  // copies are 3'-anchored truncations (LINE-like)
    uint32_t a_tail;                 // copies end in a poly-A tail of about this length (SINE-like)
};

struct Model {
    SkewProfile pf;
    uint32_t ctx_mask;
    std::vector<uint16_t> thr;       // [level][context][3] cumulative thresholds of a 16-bit draw
    std::vector<Family> fam;
    std::vector<double> fam_cum;
};

inline double approx_normal(uint64_t h) {   // sum of four 16-bit uniforms, variance 1
    const double u = ((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) / 65536.0;
    return (u - 2.0) * std::sqrt(3.0);
}

// one Markov step: 16 bits of randomness against the context's cumulative thresholds
struct Chain {
    const Model &m;
    const uint16_t *tab;
    uint32_t ctx;
    uint64_t bits = 0;
    int have = 0;
    Chain(const Model &model, unsigned level, Rng &rng)
        : m(model), tab(model.thr.data() + static_cast<size_t>(level) * (model.ctx_mask + 1) * 3),
          ctx(static_cast<uint32_t>(rng.next()) & model.ctx_mask) {}
    uint8_t step(Rng &rng) {
        if (have == 0) { bits = rng.next(); have = 4; }
        const uint32_t r = static_cast<uint32_t>(bits & 0xFFFF);
        bits >>= 16;
        have--;
        const uint16_t *t = tab + static_cast<size_t>(ctx) * 3;
        const uint8_t b = static_cast<uint8_t>((r >= t[0]) + (r >= t[1]) + (r >= t[2]));
        ctx = ((ctx << 2) | b) & m.ctx_mask;
        return b;
    }
};

inline Model make_model(uint64_t seed, const SkewProfile &pf) {
    Model m;
    m.pf = pf;
    m.ctx_mask = (1u << (2 * pf.order)) - 1u;
    const uint32_t n_ctx = m.ctx_mask + 1;
    m.thr.resize(static_cast<size_t>(pf.levels) * n_ctx * 3);
    const uint64_t tkey = splitmix64(seed ^ 0x7461626C65ull);
    for (unsigned l = 0; l < pf.levels; l++) {
        const double gc = pf.gc_lo + (pf.gc_hi - pf.gc_lo) * (pf.levels > 1 ? double(l) / (pf.levels - 1) : 0.5);
        const double pi[4] = {(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2};
        for (uint32_t c = 0; c < n_ctx; c++) {
            double w[4], sum = 0;
            for (unsigned b = 0; b < 4; b++) {
                w[b] = pi[b] * std::exp(pf.sigma * approx_normal(splitmix64(tkey + c * 4ull + b)));
                if ((c & 3) == 1 && b == 2) w[b] *= pf.cpg;
                if ((c & 3) == b && ((c >> 2) & 3) == b) w[b] *= pf.homopolymer;
                sum += w[b];
            }
            double acc = 0;
            for (unsigned b = 0; b < 3; b++) {
                acc += w[b] / sum;
                m.thr[(static_cast<size_t>(l) * n_ctx + c) * 3 + b] =
                    static_cast<uint16_t>(std::min(65535.0, std::max(1.0, std::floor(acc * 65536.0))));
            }
        }
    }
    // repeat families: consensus sequences drawn from the same chain (mid GC level)
    m.fam.resize(pf.families);
    double wsum = 0;
    for (unsigned f = 0; f < pf.families; f++) {
        Rng rng(splitmix64(seed ^ (0x66616D696C79ull + f * 0x9E3779B97F4A7C15ull)));
        Family &fa = m.fam[f];
        uint32_t len;
        if (f == 0) { len = 300; fa.weight = 0.38; fa.divergence = 0.10; fa.anchored3 = false; fa.a_tail = 22; }
        else if (f == 1) { len = 6000; fa.weight = 0.30; fa.divergence = 0.09; fa.anchored3 = true; fa.a_tail = 0; }
        else {
            len = static_cast<uint32_t>(150.0 * std::exp(rng.unit() * std::log(3000.0 / 150.0)));
            fa.weight = 0.32 / (f - 1.0) / 4.4;          // Zipf tail over the other families
            fa.divergence = 0.01 + 0.19 * rng.unit();
            fa.anchored3 = rng.below(3) == 0;
            fa.a_tail = 0;
        }
        Chain ch(m, pf.levels / 2, rng);
        fa.consensus.resize(len);
        for (uint32_t i = 0; i < len; i++) fa.consensus[i] = dna4_char(ch.step(rng));
        wsum += fa.weight;
    }
    double acc = 0;
    for (auto &fa : m.fam) { acc += fa.weight / wsum; m.fam_cum.push_back(acc); }
    return m;
}

inline void revcomp_inplace(std::string &s) {
    std::reverse(s.begin(), s.end());
    for (char &c : s) c = dna4_char(static_cast<uint8_t>(3 - dna4_rank(static_cast<uint8_t>(c))));
}

inline uint32_t geometric(Rng &rng, double p) {       // failures before the first success
    if (p >= 1.0) return 0;
    if (p <= 0.0) return 0xFFFFFFFFu;
    const double u = std::max(rng.unit(), 1e-300);
    const double g = std::floor(std::log(u) / std::log1p(-p));
    return g > 4e9 ? 0xFFFFFFFFu : static_cast<uint32_t>(g);
}

// One copy of a repeat family: a stretch of the consensus with the copy's own divergence.
inline void repeat_copy(const Model &m, Rng &rng, std::string &out) {
    const double u = rng.unit();
    size_t f = 0;
    while (f + 1 < m.fam_cum.size() && u >= m.fam_cum[f]) f++;
    const Family &fa = m.fam[f];
    const uint32_t L = static_cast<uint32_t>(fa.consensus.size());
    uint32_t a = 0, b = L;
    if (fa.anchored3) {            // most LINE-like copies are short 3' ends
        const uint32_t keep = std::min<uint32_t>(L, 100 + geometric(rng, 1.0 / 900.0));
        a = L - keep;
    } else if (rng.below(4) == 0 && L > 120) {
        a = rng.below(L - 100);
        b = a + 100 + rng.below(L - a - 100 + 1);
    }
    const double d = rng.unit() < m.pf.p_young ? 0.01 + 0.02 * rng.unit() : std::min(0.3, fa.divergence * (0.5 + rng.unit()));
    out.clear();
    uint32_t next = a + geometric(rng, d);
    for (uint32_t i = a; i < b; i++) {
        if (i == next) {
            const uint32_t kind = rng.below(10);
            if (kind < 8) {        // substitution
                char nt = dna4_char(static_cast<uint8_t>(rng.below(4)));
                while (nt == fa.consensus[i]) nt = dna4_char(static_cast<uint8_t>(rng.below(4)));
                out.push_back(nt);
            } else if (kind == 8) {   // insertion before the base
                out.push_back(dna4_char(static_cast<uint8_t>(rng.below(4))));
                out.push_back(fa.consensus[i]);
            }                      // kind 9: deletion
            next = i + 1 + geometric(rng, d);
        } else {
            out.push_back(fa.consensus[i]);
        }
    }
    if (fa.a_tail) {
        const uint32_t n = fa.a_tail / 2 + rng.below(fa.a_tail);
        for (uint32_t i = 0; i < n; i++) out.push_back(rng.below(25) == 0 ? dna4_char(static_cast<uint8_t>(rng.below(4))) : 'A');
    }
    if (rng.below(2)) revcomp_inplace(out);
}

inline void tandem_repeat(Rng &rng, std::string &out) {
    static const uint32_t unit_len[16] = {1, 1, 1, 2, 2, 2, 2, 3, 3, 4, 4, 5, 6, 17, 31, 52};
    const uint32_t ul = unit_len[rng.below(16)];
    char unit[64];
    for (uint32_t i = 0; i < ul; i++) unit[i] = dna4_char(static_cast<uint8_t>(rng.below(4)));
    const uint32_t total = (ul <= 6 ? 18u : 3 * ul) + geometric(rng, 1.0 / (ul <= 6 ? 60.0 : 400.0));
    out.clear();
    for (uint32_t i = 0; i < total; i++)
        out.push_back(rng.below(50) == 0 ? dna4_char(static_cast<uint8_t>(rng.below(4))) : unit[i % ul]);
}

inline unsigned isochore(uint64_t key, uint64_t region, unsigned levels) {
    auto u = [&](uint64_t r, uint64_t salt) { return double(splitmix64(key ^ (salt + r * 0x100000001B3ull)) >> 11) / 9007199254740992.0; };
    const double v = 0.5 * u(region >> 4, 0x11) + 0.3 * u(region >> 1, 0x22) + 0.2 * u(region, 0x33);
    return std::min(levels - 1, static_cast<unsigned>(v * levels));
}

// `src` with substitutions (and one indel in ten events) at rate d, appended to out
inline void mutate_into(const char *src, size_t n, double d, Rng &rng, std::string &out) {
    size_t next = geometric(rng, d);
    for (size_t i = 0; i < n; i++) {
        if (i != next) { out.push_back(src[i]); continue; }
        const uint32_t kind = rng.below(10);
        if (kind < 8) {
            char nt = dna4_char(static_cast<uint8_t>(rng.below(4)));
            while (nt == src[i]) nt = dna4_char(static_cast<uint8_t>(rng.below(4)));
            out.push_back(nt);
        } else if (kind == 8) {
            out.push_back(dna4_char(static_cast<uint8_t>(rng.below(4))));
            out.push_back(src[i]);
        }
        next = i + 1 + geometric(rng, d);
    }
}

inline double unit_hash(uint64_t key, uint64_t a, uint64_t salt) {
    return double(splitmix64(key ^ (salt + a * 0x100000001B3ull)) >> 11) / 9007199254740992.0;
}

// -1, or the key of the satellite array that covers the chunk starting at c0
inline int64_t satellite_array(const Model &m, uint64_t rkey, uint64_t c0) {
    const uint64_t coarse = c0 / (64ull * kChunk), fine = c0 / (8ull * kChunk);
    if (unit_hash(rkey, coarse, 0x5A7C) < m.pf.p_satellite) return static_cast<int64_t>(coarse * 2 + 1);
    if (unit_hash(rkey, fine, 0x5A7F) < m.pf.p_satellite) return static_cast<int64_t>(fine * 2);
    return -1;
}

inline void fill_satellite(const Model &m, uint64_t rkey, int64_t array, uint64_t chunk, char *s, uint64_t c0, uint64_t c1) {
    static const uint32_t monomer_len[4] = {171, 68, 42, 5};
    // the array's own variant of its family's monomer
    Rng arng(splitmix64(rkey ^ (0x5A7E11173ull + static_cast<uint64_t>(array) * 0x9E3779B97F4A7C15ull)));
    const uint32_t fam = arng.below(4);
    Rng frng(splitmix64(m.thr.size() * 0x1234567ull + fam + (static_cast<uint64_t>(m.thr[fam * 3 + 7]) << 20)));
    Chain ch(m, m.pf.levels / 2, frng);
    std::string consensus(monomer_len[fam], 'A'), monomer;
    for (char &c : consensus) c = dna4_char(ch.step(frng));
    mutate_into(consensus.data(), consensus.size(), monomer_len[fam] > 10 ? 0.06 : 0.0, arng, monomer);
    if (monomer.empty()) monomer = consensus;
    Rng rng(splitmix64(rkey + chunk * 0x2545F4914F6CDD1Dull + 0x5A7));
    std::string tmp;
    uint64_t pos = c0;
    while (pos < c1) {
        tmp.clear();
        mutate_into(monomer.data(), monomer.size(), 0.015, rng, tmp);
        const uint64_t n = std::min<uint64_t>(tmp.size(), c1 - pos);
        std::memcpy(s + pos, tmp.data(), n);
        pos += n;
    }
}

// The chunk's own content: unique segments interleaved with repeat copies, tandem repeats and gaps.
inline void fill_primary(const Model &m, uint64_t rkey, uint64_t chunk, char *s, uint64_t c0, uint64_t c1,
                         std::vector<std::pair<uint64_t, uint64_t>> &gaps) {
    Rng rng(splitmix64(rkey + chunk * 0x2545F4914F6CDD1Dull));
    const unsigned level = isochore(rkey, c0 / (2 * kChunk), m.pf.levels);
    std::string tmp;
    uint64_t pos = c0;
    auto put = [&](const std::string &t) {
        const uint64_t n = std::min<uint64_t>(t.size(), c1 - pos);
        std::memcpy(s + pos, t.data(), n);
        pos += n;
    };
    while (pos < c1) {
        // a unique segment on a random strand
        const uint64_t want = 50 + geometric(rng, 1.0 / m.pf.mean_unique);
        const uint64_t n = std::min<uint64_t>(want, c1 - pos);
        Chain ch(m, level, rng);
        tmp.resize(n);
        for (uint64_t i = 0; i < n; i++) tmp[i] = dna4_char(ch.step(rng));
        if (rng.below(2)) revcomp_inplace(tmp);
        put(tmp);
        if (pos >= c1) break;
        const double u = rng.unit();
        if (u < m.pf.p_repeat) {
            repeat_copy(m, rng, tmp);
            put(tmp);
        } else if (u < m.pf.p_repeat + m.pf.p_tandem) {
            tandem_repeat(rng, tmp);
            put(tmp);
        } else if (u < m.pf.p_repeat + m.pf.p_tandem + m.pf.p_gap) {
            const uint64_t len = std::min<uint64_t>(c1 - pos, static_cast<uint64_t>(100.0 * std::exp(rng.unit() * std::log(300.0))));
            std::memset(s + pos, 'A', len);
            gaps.emplace_back(pos, len);
            pos += len;
        }
    }
}

// Fills s[c0, c1) of one record (record_len bases long); gaps of this chunk are appended to `gaps` as
// (start, length) in the record.  A pure function of (model, rkey, chunk).
inline void fill_chunk(const Model &m, uint64_t rkey, uint64_t chunk, uint64_t record_len, char *s, uint64_t c0, uint64_t c1,
                       std::vector<std::pair<uint64_t, uint64_t>> &gaps) {
    const int64_t array = satellite_array(m, rkey, c0);
    if (array >= 0) {
        fill_satellite(m, rkey, array, chunk, s, c0, c1);
        return;
    }
    const uint64_t n_chunks = (record_len + kChunk - 1) / kChunk;
    if (n_chunks > 4 && unit_hash(rkey, chunk, 0x5D0B) < m.pf.p_segdup) {
        // a diverged copy of another chunk's OWN content (whatever that chunk itself turned out to be)
        Rng rng(splitmix64(rkey + chunk * 0x2545F4914F6CDD1Dull + 0x5D));
        uint64_t src = rng.below(static_cast<uint32_t>(n_chunks - 1));
        if (src >= chunk) src++;
        const uint64_t s0 = src * kChunk, s1 = std::min<uint64_t>(record_len, s0 + kChunk);
        std::string source(s1 - s0, 'A'), copy;
        std::vector<std::pair<uint64_t, uint64_t>> ignored;
        fill_primary(m, rkey, src, source.data() - s0, s0, s1, ignored);
        copy.reserve(source.size() + 64);
        mutate_into(source.data(), source.size(), 0.005 + 0.025 * rng.unit(), rng, copy);
        if (rng.below(2)) revcomp_inplace(copy);
        const uint64_t n = std::min<uint64_t>(copy.size(), c1 - c0);
        std::memcpy(s + c0, copy.data(), n);
        if (c0 + n < c1) {   // the copy came out shorter than the chunk: the rest is the chunk's own content
            std::string rest(c1 - c0, 'A');
            fill_primary(m, rkey, chunk, rest.data() - c0, c0, c1, ignored);
            std::memcpy(s + c0 + n, rest.data() + n, c1 - c0 - n);
        }
        return;
    }
    fill_primary(m, rkey, chunk, s, c0, c1, gaps);
}

}  // namespace skew_detail

inline Genome synth_genome_skewed(uint64_t seed, const std::vector<uint64_t> &record_lengths, unsigned n_threads = 0,
                                  const SkewProfile &pf = SkewProfile()) {
    using namespace skew_detail;
    Genome g;
    const Model m = make_model(seed, pf);
    struct Job { uint32_t r; uint64_t chunk; };
    std::vector<Job> jobs;
    for (size_t r = 0; r < record_lengths.size(); r++) {
        g.ids.push_back("synth" + std::to_string(r + 1) + " genome-like seed=" + std::to_string(seed) +
                        " len=" + std::to_string(record_lengths[r]));
        g.seqs.emplace_back(record_lengths[r], 'A');
        for (uint64_t c = 0; c * kChunk < record_lengths[r]; c++) jobs.push_back(Job{static_cast<uint32_t>(r), c});
    }
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = static_cast<unsigned>(std::min<size_t>(n_threads, std::max<size_t>(1, jobs.size() / 8)));
    std::vector<std::vector<std::array<uint64_t, 3>>> found(nt);
    std::atomic<size_t> next{0};
    auto work = [&](unsigned t) {
        std::vector<std::pair<uint64_t, uint64_t>> gaps;
        for (size_t j = next.fetch_add(64); j < jobs.size(); j = next.fetch_add(64))
            for (size_t i = j; i < std::min(jobs.size(), j + 64); i++) {
                const Job &jb = jobs[i];
                const uint64_t rkey = splitmix64(seed ^ (0x5EC0DE5ull + jb.r * 0x100000001B3ull));
                const uint64_t c0 = jb.chunk * kChunk, c1 = std::min<uint64_t>(record_lengths[jb.r], c0 + kChunk);
                gaps.clear();
                fill_chunk(m, rkey, jb.chunk, record_lengths[jb.r], g.seqs[jb.r].data(), c0, c1, gaps);
                for (auto &gp : gaps) found[t].push_back({jb.r, gp.first, gp.second});
            }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &t : pool) t.join();
    for (auto &v : found) g.gaps.insert(g.gaps.end(), v.begin(), v.end());
    std::sort(g.gaps.begin(), g.gaps.end());
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
