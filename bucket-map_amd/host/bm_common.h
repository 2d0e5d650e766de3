// bm_common.h -- small shared pieces of the host side (C++17, no third-party dependencies).
//
// The reference builds on SeqAn3 for alphabets, k-mer hashing and file I/O; SeqAn3 is not available
// here, so the few semantics the hot path relies on are restated from SURVEY.md Appendix C.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace bm {

// SeqAn3 dna4 assign_char: case-insensitive, U->T, IUPAC codes fold to a fixed
// base, anything else (N included) -> A.
inline uint8_t dna4_rank(uint8_t c) {
    switch (c) {
    case 'C': case 'c': case 'Y': case 'y': case 'S': case 's': case 'B': case 'b': return 1;
    case 'G': case 'g': case 'K': case 'k': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 0;
    }
}
inline char dna4_char(uint8_t r) { return "ACGT"[r & 3]; }

// utils.h:291-302
inline uint32_t hash_reverse_complement(uint32_t hash, uint32_t k) {
    uint32_t rc = 0;
    for (uint32_t i = 0; i < k; i++) {
        rc = (rc << 2) | ((~hash) & 3u);
        hash >>= 2;
    }
    return rc;
}

// Sampler::sample_deterministically (utils.h:160-178).  Stateless: the reference's cache variable
// `last_upper_bound` is never updated, so the only effect of the cache is upper_bound == 0, where the
// reference keeps stale positions (an out-of-bounds read downstream); that case yields zeros here.
inline std::vector<uint32_t> sample_deterministically(uint32_t n, uint32_t upper_bound) {
    std::vector<uint32_t> s;
    if (n == 0) return s;
    double delta = 0.0;
    if (n != 1) delta = static_cast<double>(upper_bound + 1u) / (n - 1u);
    for (uint32_t i = 0; i + 1 < n; i++) s.push_back(static_cast<uint32_t>(std::floor(i * delta)));
    s.push_back(upper_bound);
    return s;
}

// float32 parameter derivations; volatile pins the product to float32.
inline uint32_t ceil_mul_f32(float a, uint32_t b) {
    volatile float prod = a * static_cast<float>(b);
    return static_cast<uint32_t>(std::ceil(static_cast<double>(prod)));
}
inline uint32_t trunc_mul_f32(float a, uint32_t b) {
    volatile float prod = a * static_cast<float>(b);
    return static_cast<uint32_t>(prod);
}

// utils.h:309-311
using segment_info_t = std::pair<unsigned int, int>;
using segments_t = std::vector<std::vector<segment_info_t>>;

// splitmix64: the one PRNG of the synthetic-data tools (counter-based use: hash of seed+index).
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() { return splitmix64(s++); }
    // uniform in [0, n): multiply-shift (n < 2^32 in every use)
    uint32_t below(uint32_t n) { return static_cast<uint32_t>(((next() >> 32) * static_cast<uint64_t>(n)) >> 32); }
    double unit() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
};

}  // namespace bm
