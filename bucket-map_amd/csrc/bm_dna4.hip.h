// bm_dna4.hip.h -- SeqAn3's dna4 assign_char on the device, without a table, shared by the filter's
// sample kernel, the locator scan and the index build.  Letters fold by their low five bits (either case):
// C Y S B -> 1, G K -> 2, T U -> 3, every other byte -> 0 (A).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmdna {

__device__ __forceinline__ uint32_t dna4_code(uint32_t c) {
    constexpr uint64_t kRank = (1ull << (2 * 3)) | (1ull << (2 * 25)) | (1ull << (2 * 19)) | (1ull << (2 * 2)) |
                               (2ull << (2 * 7)) | (2ull << (2 * 11)) | (3ull << (2 * 20)) | (3ull << (2 * 21));
    const uint32_t letter = (c & 0xDFu) - 0x41u;
    return letter < 26u ? (uint32_t)(kRank >> (2u * (c & 31u))) & 3u : 0u;
}

// Four ASCII bytes -> their four ranks as one byte, the first (lowest-address) base in the top two bits.
// ((c >> 1) & 3) ^ (that >> 1) is right for A C G T in either case; v_perm_b32 rebuilds the letters those ranks stand
// for, and only a word that is NOT its own rebuild (N, IUPAC, anything else) takes the byte-wise exact folding.
__device__ __forceinline__ uint32_t dna4_pack4(uint32_t w) {
    const uint32_t t = (w >> 1) & 0x03030303u;
    uint32_t code = t ^ ((t >> 1) & 0x01010101u);
    if (__builtin_amdgcn_perm(0u, 0x54474341u, code) != (w & 0xDFDFDFDFu))
        code = dna4_code(w & 0xFFu) | (dna4_code((w >> 8) & 0xFFu) << 8) | (dna4_code((w >> 16) & 0xFFu) << 16) |
               (dna4_code(w >> 24) << 24);
    const uint32_t r = __builtin_amdgcn_perm(0u, code, 0x00010203u);    // byte-reversed: first base in byte 3
    const uint32_t x = r | (r >> 6);
    return (x | (x >> 12)) & 0xFFu;
}

// Sixteen ASCII bytes -> one word of the 2-bit stream, the first base in the top two bits.
__device__ __forceinline__ uint32_t dna4_pack16(const uint4 &v) {
    return (dna4_pack4(v.x) << 24) | (dna4_pack4(v.y) << 16) | (dna4_pack4(v.z) << 8) | dna4_pack4(v.w);
}

}  // namespace bmdna
