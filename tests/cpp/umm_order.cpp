// umm_order.cpp -- TEST INFRASTRUCTURE.  The locator oracle (oracle/bm_locator_oracle.c) and the GPU scan
// replay the occurrences of a k-mer in DESCENDING bucket offset, because that is the order in which
// libstdc++'s std::unordered_multimap::equal_range yields equal keys for the reference's usage
// (bucket_map/locator/bucket_locator.h:168-176: reserve(BM_BUCKET_LEN), then emplace in ascending
// offset; :246-249 iterates equal_range).  This program checks the claim against the real container of
// this toolchain, with the reference's sizes and with a table that has to rehash.
// Prints "OK" and exits 0 when every equal_range comes back in strictly descending offset.
#include <cstdint>
#include <cstdio>
#include <unordered_map>
#include <vector>

static uint64_t splitmix64(uint64_t &s) {
    uint64_t x = (s += 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static bool check(unsigned reserve, unsigned n_entries, unsigned key_space, uint64_t seed) {
    std::unordered_multimap<unsigned int, int> index;
    index.reserve(reserve);
    std::vector<unsigned> keys(n_entries);
    for (unsigned i = 0; i < n_entries; i++) keys[i] = static_cast<unsigned>(splitmix64(seed) % key_space);
    int offset = 0;
    for (unsigned k : keys) {
        index.emplace(k, offset);
        offset++;
    }
    for (unsigned k = 0; k < key_space; k++) {
        auto range = index.equal_range(k);
        int last = 1 << 30;
        unsigned seen = 0;
        for (auto it = range.first; it != range.second; ++it) {
            if (it->second >= last) return false;
            last = it->second;
            seen++;
        }
        (void)seen;
    }
    return true;
}

int main() {
    // the reference's geometry: 65 536 reserved, 65 825 k-mers of a 65 836-base bucket, many repeats
    bool ok = check(65536, 65825, 4096, 1) && check(65536, 65825, 300, 2) && check(65536, 65825, 16777216, 3);
    // a bucket far larger than the reservation (forces rehashes while inserting)
    ok = ok && check(1024, 200000, 5000, 4) && check(16, 70000, 7, 5);
    std::puts(ok ? "OK" : "FAIL: equal_range is not in descending insertion order");
    return ok ? 0 : 1;
}
