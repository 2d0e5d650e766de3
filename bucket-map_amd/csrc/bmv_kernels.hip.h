// bmv_kernels.hip.h -- gfx950 kernels of the alignment verifier (include/bmv.h).
//
// What the BM_ALIGN build of the reference asks of SeqAn3 per located candidate
// (bucket_map/locator/bucket_locator.h:520-528,560-589): the edit distance of the whole query against
// the best substring of the text (global alignment with free end gaps in sequence1), where that substring
// begins, and the alignment itself as a CIGAR.
//
// Myers' bit-vector recurrence (Hyyro's block form): one 64-bit word carries 64 query rows of one text
// column as +1/-1 vertical deltas (Pv, Mv); a column step is ~20 word operations.  A query of m bases is
// ceil(m/64) words; they are laid over the lanes of a GROUP (CW consecutive words per lane, 1..8) and the lanes
// are skewed along the text: at step t lane l works on column t-l and takes the horizontal delta that
// leaves lane l-1's last row (computed one step earlier) through a DPP move.  A group is exactly as many lanes
// as the longest query of the batch has words over CW (any size, not a power of two: lanes are addressed explicitly),
// so a wave verifies 64/G candidates at once: 32 for 1-kbp reads (2 lanes x 8 words), 4 for 5-kbp reads (16 x 5),
// 2 for 10-kbp reads (32 x 5); queries of up to 512 bases go one per LANE (bmv_align_lane_kernel, below).  The host picks
// CW (bmv_api.hip, pick_shape): what a step pays once -- the neighbour's delta, the text base, the loop -- is worth about
// two words' recurrences, so more words per lane and more alignments per wave win until the registers (10 per word) cost
// waves per SIMD.
// A query of more than 512 words (32 768 bases) is processed in STRIPS of 64 * CW words, one
// after the other over the whole text: the horizontal deltas entering a strip's first word are the stored
// ones that left the last word of the strip above.
//
// Traceback.  Per cell two bits decide the walk: "the diagonal predecessor is valid" (match with diagonal
// delta 0, or mismatch with diagonal delta 1: ~(Eq ^ D0)) and "the upper predecessor is valid" (vertical delta
// +1: Pv); when neither holds the left one must be.  Writing them for every cell is 16 B per word and column
// -- 25 KB per 300x307 alignment, 28 MB per 10-kbp alignment -- and the stores, not the arithmetic, set the
// kernel's time.  So the forward pass keeps only CHECKPOINTS: every kBlock = 16 STEPS the vertical state
// (Pv, Mv) of each word, and for every step the horizontal delta that leaves each word (a +1 bit and a -1 bit, 16 steps
// per 32-bit word): 1.25 B per word and column.  The traceback walks from the LAST minimum of the bottom row,
// trying diagonal, up, left in that order (include/bmv.h, tie rules 1-2); when it enters a (word, block) CELL
// it recomputes that block's 16 steps from the checkpoint -- the horizontal deltas entering
// the word are the stored ones of the word above, so one word is recomputed on its own -- and 16 / SLOTS lanes of
// the group keep the cell's 16 pairs of trace words in registers, SLOTS each; the group's other lanes do the same for the
// cells the walk will reach next if it keeps to its diagonal, so one round of recomputation serves GROUP * SLOTS / 16 cells.
// Blocks are cut in TIME, not by column: lane l is at column
// t - l, so a block of a word on lane l is the 16 columns 16 b + 1 - l ..; every lane of the wave ends its block at the
// same step and the wave stores once in 16 steps (cut by column, a few lanes of every group reached a block end at
// every step, and the wave ran the store path at every step).  The deltas of the word above for a block's columns are
// the same block's if that word sits on the same lane, shifted by one step if it sits on the lane before.
// Run-length CIGAR entries are left in reverse; bmv_gather_kernel (bmv_api.hip) reverses and packs them.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmv {

constexpr int kWave = 64;

struct Job {
    const uint8_t *genome;          // ASCII
    const uint8_t *reads;           // ASCII
    const uint8_t *lut;             // dna4 rank of a char (256 entries)
    const uint64_t *text_start;     // per alignment of the batch
    const uint32_t *text_len;
    const uint8_t *text_rc;
    const uint64_t *query_start;
    const uint32_t *query_len;
    const uint32_t *order;          // this launch handles alignments order[0 .. count): the host's length classes
    uint32_t count;
    uint64_t *trace;                // one region of trace_stride words per wave (64/GROUP alignments):
    uint64_t trace_stride;          //   checkpoints (Pv, Mv) [block][group][lane][c], then horizontal deltas
    uint32_t trace_words;           // words reserved per alignment and block in this batch (>= every W)
    uint32_t trace_blocks;          // column blocks reserved per alignment (>= every ceil(n / 16) + 1)
    uint32_t group;                 // lanes per alignment (1..64); 64 / group alignments per wave
    uint32_t *ops_rev;              // count x ops_stride reversed CIGAR entries
    uint32_t ops_stride;
    uint32_t text_lds_stride;       // bytes of LDS per group for the text window (2 bits per base, a multiple of 4)
    int32_t *out_score;             // per alignment of the batch
    uint32_t *out_begin;
    uint32_t *out_nops;             // per slot
    uint32_t stop_after;            // experiments (BMV_STOP_AFTER): 1 = return after the set-up, 2 = after the forward pass
};

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src_lane) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src_lane, kWave);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src_lane, kWave);
    return ((uint64_t)hi << 32) | lo;
}

constexpr uint32_t kBlock = 16;   // columns between checkpoints = horizontal deltas per 32-bit word

// One column step of one 64-row word (Myers 1999 in Hyyro's block form).  eq0: rows that match the text base.  The
// horizontal delta enters and leaves as two WORDS whose bit 31 counts (hpw: +1, hmw: -1): they are simply the high halves of
// the Ph and Mh of the word above, and the 64-bit shifts take them in through v_alignbit -- a packed 2-bit code cost a shift
// and two masks per word and step to put together and take apart.  Updates (pv, mv) to this column; ph / mh are the
// horizontal deltas of the rows BEFORE the shift (bit r = row r of the word), d0 the rows whose diagonal delta is 0.
__device__ __forceinline__ uint64_t shl1_carry(uint64_t v, uint32_t carry_word) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 31) << 32) | __builtin_amdgcn_alignbit(lo, carry_word, 31);
}

__device__ __forceinline__ void myers_step_carry(uint64_t eq0, uint32_t &hpw, uint32_t &hmw, uint64_t &pv, uint64_t &mv, uint64_t &ph,
                                                 uint64_t &mh, uint64_t &d0) {
    const uint64_t xv = eq0 | mv;
    const uint64_t eq = eq0 | (hmw >> 31);
    const uint64_t xh = (((eq & pv) + pv) ^ pv) | eq;
    ph = mv | ~(xh | pv);
    mh = pv & xh;
    d0 = xh | mv;
    const uint64_t phs = shl1_carry(ph, hpw), mhs = shl1_carry(mh, hmw);
    hpw = (uint32_t)(ph >> 32);
    hmw = (uint32_t)(mh >> 32);
    pv = mhs | ~(xv | phs);
    mv = phs & xv;
}

// The rows of a word that match a text base: the query is kept as the two bit planes of its ranks (q0: low bit, q1: high
// bit -- 4 registers per word where four match masks are 8), the base as two all-or-nothing words.  Rows past the query's
// end compare like any other: nothing below the last row flows back up (carries and shifts go towards higher rows).
__device__ __forceinline__ uint64_t match_rows(uint64_t q0, uint64_t q1, uint64_t not_t0, uint64_t not_t1) {
    return (q0 ^ not_t0) & (q1 ^ not_t1);                       // (xnor with the base's bits, the negation paid once per step)
}
// the all-or-nothing word of one bit of a text base, negated (for match_rows)
__device__ __forceinline__ uint64_t not_plane(bool bit) { return bit ? 0ull : ~0ull; }

// SLOTS: trace-word pairs a lane keeps during the traceback (16 / SLOTS lanes of a group hold a cell's 16 columns, and the
// group holds GROUP * SLOTS / 16 cells at a time; needs group >= 16 / SLOTS).  CW: 64-row words per lane.  STRIPS: queries of
// more than 64 * CW words are allowed (the group is then the whole wave).
template <int SLOTS, int CW, bool STRIPS>
__global__ __launch_bounds__(kWave) void bmv_align_kernel(Job J) {
    extern __shared__ uint8_t lds_text[];
    const uint32_t GROUP = J.group, GPW = kWave / GROUP;        // lanes per alignment, alignments per wave
    const uint32_t lane = threadIdx.x, grp = lane / GROUP, gl = lane - grp * GROUP, lane0 = grp * GROUP;
    const uint32_t slot = blockIdx.x * GPW + grp;
    const bool have = grp < GPW && slot < J.count;              // (lanes past the last whole group are spare)
    const uint32_t a = J.order[have ? slot : 0u];
    const uint32_t n = have ? J.text_len[a] : 0u, m = have ? J.query_len[a] : 0u;
    const uint32_t W = (m + 63u) >> 6;                         // words of the query
    const uint32_t strip_words = GROUP * CW;                    // words one pass over the text carries
    const uint32_t n_strips = (STRIPS && W) ? (W + strip_words - 1u) / strip_words : 1u;
    // LDS: the dna4 folding table, then per group the text window as a 2-bit stream (16 columns per 32-bit word, column
    // 16 i + x in bits 2x.. of word i).  Nothing else: the LDS a wave holds decides how many waves a CU runs, and this
    // kernel is a chain of dependent steps that needs them.
    uint8_t *lut = lds_text;
    uint32_t *text = reinterpret_cast<uint32_t *>(lds_text + 256 + (size_t)grp * J.text_lds_stride);
    reinterpret_cast<uint32_t *>(lut)[lane] = reinterpret_cast<const uint32_t *>(J.lut)[lane];
    __syncthreads();
    if (have) {
        // text window, reverse-complemented if asked (bucket_locator.h:562-567)
        const uint8_t *src = J.genome + J.text_start[a];
        const bool rc = J.text_rc[a] != 0;
        for (uint32_t i = gl; i * kBlock < n; i += GROUP) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t x = 0; x < kBlock; x++) {
                const uint32_t j = i * kBlock + x;
                if (j < n) {
                    const uint32_t r = lut[rc ? src[n - 1u - j] : src[j]];
                    v |= (rc ? 3u - r : r) << (2u * x);
                }
            }
            text[i] = v;
        }
    }
    __syncthreads();
    // checkpoint entry of (block b, this group, word w of the query): (b * GPW + grp) * TW + w
    const uint32_t TW = J.trace_words;
    uint64_t *ckpt = J.trace + (size_t)blockIdx.x * J.trace_stride;
    uint32_t *hbuf = reinterpret_cast<uint32_t *>(ckpt + (size_t)J.trace_blocks * GPW * TW * 2u);
    // (32-bit: bmv_create's limits keep a wave's entries below 2^23)
    auto entry = [&](uint32_t b, uint32_t w) { return (b * GPW + grp) * TW + w; };
    // behind the deltas: the bit planes of every query word, [group][word], for the traceback
    uint64_t *planes = reinterpret_cast<uint64_t *>(hbuf + (((size_t)J.trace_blocks * GPW * TW + 1u) & ~(size_t)1));

    uint64_t q0[CW], q1[CW];                                    // the bit planes of this lane's query words (match_rows)
    int32_t score = (int32_t)m, best = (int32_t)m;              // tracked by the lane that holds row m
    uint32_t best_j = 0;
    const uint32_t last_word = W ? W - 1u : 0u, last_bit = (m - 1u) & 63u;
    const uint32_t last_lane = (last_word % strip_words) / CW;
    // which of a lane's CW words can be the query's last: a wave that carries one alignment has the same answer in every
    // lane, and a branch skips the bottom-row bookkeeping in the other words' steps
    const uint32_t last_c = CW == 1 ? 0u : (GPW == 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(last_word % CW)) : last_word % CW);
    // the wave's groups run the same number of strips and steps (shuffles and barriers need the whole wave)
    uint32_t wave_strips = n_strips;
    if (STRIPS) {
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)wave_strips, o, kWave);
            wave_strips = other > wave_strips ? other : wave_strips;
        }
    }
    for (uint32_t strip = 0; strip < wave_strips; strip++) {
        const uint32_t w0 = strip * strip_words;                // first word of the strip
        const bool live = have && strip < n_strips;
        const uint32_t Ws = live ? (W - w0 < strip_words ? W - w0 : strip_words) : 0u;   // words in this strip
        const uint32_t L = (Ws + CW - 1u) / CW;                 // lanes of the group that hold rows of it
        // the two bit planes of this lane's words' ranks, straight from the read (rows past the query's end: rank 0 -- they
        // compare like any other row, nothing below the last row flows back up)
#pragma unroll
        for (int c = 0; c < CW; c++) {
            uint64_t p0 = 0, p1 = 0;
            const uint32_t w = gl * CW + c;
            if (live && w < Ws) {
                const uint32_t row0 = (w0 + w) * 64u, rows = m - row0 < 64u ? m - row0 : 64u;
                const uint8_t *q = J.reads + J.query_start[a] + row0;
#pragma unroll 8
                for (uint32_t k = 0; k < 64u; k++) {
                    const uint64_t r = k < rows ? lut[q[k]] : 0u;
                    p0 |= (r & 1u) << k;
                    p1 |= (r >> 1) << k;
                }
            }
            q0[c] = p0;
            q1[c] = p1;
        }
        uint64_t pv[CW], mv[CW];
        uint32_t hacc_p[CW], hacc_m[CW];                        // deltas leaving each word in this block of steps, the latest in bit 0
#pragma unroll
        for (int c = 0; c < CW; c++) {
            pv[c] = ~0ull;                                      // column 0: H[i][0] = i
            mv[c] = 0;
            hacc_p[c] = hacc_m[c] = 0;
        }
        uint32_t steps = (live && Ws) ? n + L - 1u : 0u;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)steps, o, kWave);
            steps = other > steps ? other : steps;
        }
        uint32_t hp_prev = 0, hm_prev = 0;                      // the deltas that left this lane's last word a step ago, in bit 31
        uint32_t habove = 0u;                                   // deltas leaving the strip above, 16 steps at a time
        const uint32_t above_lane = GROUP - 1u;                 // the lane that held the strip above's last word (a full strip)
        const bool holds = live && gl < L;                      // this lane carries words of the strip
        for (uint32_t t = 1; t <= steps; t++) {
            // what left lane l-1's last row a step ago: DPP moves down the whole wave by one lane (wave_shr:1), VALU
            // operations -- __shfl_up is a round trip through the LDS crossbar, waited for at the head of every step
            uint32_t hpw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hp_prev, 0x138, 0xF, 0xF, false);
            uint32_t hmw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm_prev, 0x138, 0xF, 0xF, false);
            const uint32_t j = t - gl;                          // 1-based text column of this lane
            const uint32_t xt = (t - 1u) % kBlock;              // blocks are cut in TIME: the same for every lane of the wave
            if (holds && j >= 1u && j <= n && t >= gl + 1u) {
                const uint32_t ch = (text[(j - 1u) / kBlock] >> (2u * ((j - 1u) % kBlock))) & 3u;
                const uint64_t nt0 = not_plane((ch & 1u) != 0), nt1 = not_plane((ch & 2u) != 0);
                if (gl == 0) {
                    // row 0 is all zeros (free leading text gaps); below the first strip the deltas come from
                    // the last word of the strip above, which passed column j at its own step j + above_lane
                    if (STRIPS && strip) {
                        const uint32_t ta = j + above_lane - 1u;
                        if (ta % kBlock == 0u || j == 1u) habove = hbuf[entry(ta / kBlock, w0 - 1u)];
                        hpw = habove << (kBlock + ta % kBlock); // step x of a block: +1 in bit 15 - x, -1 in bit 31 - x
                        hmw = habove << (ta % kBlock);
                    } else {
                        hpw = hmw = 0;
                    }
                }
                // Every word of the lane, also those past the query's end in its last lane: rows nobody looks at.  The test
                // is never true (trace_words >= GROUP * CW): a scalar branch between the words keeps the scheduler from
                // interleaving them, which costs registers and so waves (CW = 8: 161 against 148) for nothing.
#pragma unroll
                for (int c = 0; c < CW; c++) {
                    if ((uint32_t)c >= TW) break;
                    const uint64_t eq0 = match_rows(q0[c], q1[c], nt0, nt1);
                    uint64_t ph, mh, d0;
                    myers_step_carry(eq0, hpw, hmw, pv[c], mv[c], ph, mh, d0);
                    if ((uint32_t)c == last_c && w0 + gl * CW + (uint32_t)c == last_word) {
                        if (CW > 1) asm volatile("" ::: "memory");   // keeps this a branch: not to be if-converted
                        score += (int32_t)((ph >> last_bit) & 1ull) - (int32_t)((mh >> last_bit) & 1ull);
                        if (score <= best) {                    // the LAST minimum of the bottom row
                            best = score;
                            best_j = j;
                        }
                    }
                    hacc_p[c] = __builtin_amdgcn_alignbit(hacc_p[c], hpw, 31);     // (hacc << 1) | bit 31 of the delta word
                    hacc_m[c] = __builtin_amdgcn_alignbit(hacc_m[c], hmw, 31);
                }
                hp_prev = hpw;
                hm_prev = hmw;
            } else if (holds) {
                // a step this lane sits out (the skew's ramps): its place in the block's record stays empty
#pragma unroll
                for (int c = 0; c < CW; c++) {
                    hacc_p[c] <<= 1;
                    hacc_m[c] <<= 1;
                }
            }
            // End of a time block -- for the whole wave at once, a branch taken one step in 16 (cut by COLUMN, the lanes of
            // a group, one column apart, reached their block ends one after the other and the wave stored something at every
            // step): the horizontal deltas of the block and the vertical state the next one starts from.
            if (xt == kBlock - 1u || t == steps) {
                const uint32_t bt = (t - 1u) / kBlock;
#pragma unroll
                for (int c = 0; c < CW; c++) {
                    const uint32_t w = gl * CW + c;
                    if (holds && w < Ws) {
                        const uint32_t e = entry(bt, w0 + w);
                        // step x of the block to bit 15 - x (+1) and 31 - x (-1), also in a last block of fewer than 16 steps
                        hbuf[e] = ((hacc_p[c] << (kBlock - 1u - xt)) & 0xFFFFu) | (hacc_m[c] << (2u * kBlock - 1u - xt));
                        uint64_t *ck = ckpt + (size_t)(e + GPW * TW) * 2u;
                        ck[0] = pv[c];
                        ck[1] = mv[c];
                    }
                    hacc_p[c] = hacc_m[c] = 0;
                }
            }
        }
        // the planes of this strip's words, for the traceback (which recomputes any word of any strip, on any lane)
#pragma unroll
        for (int c = 0; c < CW; c++) {
            const uint32_t w = gl * CW + c;
            if (holds && w < Ws) {
                planes[(size_t)(grp * TW + w0 + w) * 2u] = q0[c];
                planes[(size_t)(grp * TW + w0 + w) * 2u + 1u] = q1[c];
            }
        }
        // the next strip (and the traceback) read what other lanes of this wave wrote: same wave, same L1,
        // lines never read before, so completing the stores (workgroup scope) is all the ordering needed --
        // a device-wide fence here cost 3.7 ms per million alignments
        __threadfence_block();
    }
    __syncthreads();
    if (!have) return;

    best = __shfl(best, (int)(lane0 + last_lane), kWave);
    best_j = (uint32_t)__shfl((int)best_j, (int)(lane0 + last_lane), kWave);
    if (m == 0) {                                               // H[0][j] = 0 everywhere: last column
        best = 0;
        best_j = n;
    }

    // Traceback, the whole group in step; lane 0 of the group writes.  A ROUND recomputes not one (word, block) but as many
    // as the group has lanes to hold: n_cells = GROUP / kHold cells, kHold = 16 / SLOTS lanes each keeping SLOTS columns
    // of their cell (lane h of a cell's lanes keeps column x in slot x / kHold if x % kHold == h) -- the cells the walk will
    // cross if it keeps to the diagonal it stands on.  A refill is the same 16 column steps for every lane, whatever cell it
    // works on, so a round costs what one cell cost; the walk then goes from cell to cell for as long as the cell it enters is
    // the one the next lanes hold (an indel near a cell's corner can send it elsewhere: a new round starts there).  A 10-kbp
    // read crosses ~850 cells: that many rounds of ~900 instructions were a quarter of the kernel.
    constexpr int kSlots = SLOTS;
    constexpr uint32_t kHold = kBlock / SLOTS;
    const uint32_t n_cells = GROUP / kHold ? GROUP / kHold : 1u;
    const uint32_t cell = gl / kHold, hold = gl % kHold;        // (lanes past the last whole cell: cell >= n_cells, never asked)
    uint64_t db[kSlots], ub[kSlots];
#pragma unroll
    for (int q = 0; q < kSlots; q++) db[q] = ub[q] = 0;
    // Blocks are cut in time: word w sits on lane lane_of(w) and passes column j at step j + lane_of(w); block b of a word
    // is its steps 16 b + 1 .. 16 b + 16.
    auto lane_of = [&](uint32_t w) { return (STRIPS ? w % strip_words : w) / CW; };
    // checkpoint of (word w, block b): vertical state the block starts from, and the deltas entering the word at the block's
    // 16 steps = the deltas that left the word above at the same COLUMNS (row 0 for the first word: all zero).  The word
    // above passed those columns at the same steps if it sits on the same lane, one step earlier on the lane before, and
    // GROUP - 1 steps later if it is the last word of the strip above.
    struct Checkpoint {
        uint64_t pv, mv;
        uint32_t hw;
    };
    auto load_checkpoint = [&](uint32_t w, uint32_t b) {
        Checkpoint k{~0ull, 0ull, 0u};                          // block 0 starts from column 0: H[i][0] = i; row 0: deltas 0
        if (b) {
            const uint64_t *ck = ckpt + (size_t)entry(b, w) * 2u;
            k.pv = ck[0];
            k.mv = ck[1];
        }
        if (w) {
            const uint32_t lw = lane_of(w), la = lane_of(w - 1u);
            // (a record holds step x of its block at bit 15 - x, +1, and 31 - x, -1)
            if (la == lw) {
                k.hw = hbuf[entry(b, w - 1u)];
            } else if (la + 1u == lw) {                         // entry x of this block = entry x - 1 of the word above's
                const uint32_t before = b ? hbuf[entry(b - 1u, w - 1u)] : 0u;
                k.hw = ((hbuf[entry(b, w - 1u)] >> 1) & 0x7FFF7FFFu) | ((before & 0x00010001u) << 15);
            } else {                                            // first word of a strip: entry x = entry x + (GROUP - 1) above
                const uint32_t sh = (GROUP - 1u) % kBlock, q = (GROUP - 1u) / kBlock;
                const uint32_t lo = hbuf[entry(b + q, w - 1u)], hi = hbuf[entry(b + q + 1u, w - 1u)];
                const uint32_t plus = (((lo & 0xFFFFu) << sh) | ((hi & 0xFFFFu) >> (kBlock - sh))) & 0xFFFFu;
                const uint32_t minus = (((lo >> 16) << sh) | ((hi >> 16) >> (kBlock - sh))) & 0xFFFFu;
                k.hw = plus | (minus << 16);
            }
        }
        return k;
    };
    auto refill = [&](uint32_t w, uint32_t b, Checkpoint k) {
        // the word's bit planes, parked beside the checkpoints by the forward pass
        const uint64_t pm0 = planes[(size_t)(grp * TW + w) * 2u], pm1 = planes[(size_t)(grp * TW + w) * 2u + 1u];
        // the block's 16 text bases: columns 16 b + 1 - lane_of(w) + x, out of two words of the 2-bit stream
        const int32_t c0 = (int32_t)(b * kBlock) - (int32_t)lane_of(w);          // 0-based column of x = 0 (may be negative)
        const int32_t tq = c0 >> 4;                             // floor
        const uint32_t tsh = (uint32_t)(c0 & 15);
        const uint32_t tlo = tq >= 0 ? text[tq] : 0u, thi = tq + 1 >= 0 ? text[tq + 1] : 0u;
        const uint32_t tw = tsh ? (tlo >> (2u * tsh)) | (thi << (32u - 2u * tsh)) : tlo;
#pragma unroll
        for (int x = 0; x < (int)kBlock; x++) {
            const int32_t col = c0 + 1 + x;                     // 1-based
            if (col >= 1 && col <= (int32_t)n) {
                const uint64_t eq0 = match_rows(pm0, pm1, not_plane(((tw >> (2 * x)) & 1u) != 0), not_plane(((tw >> (2 * x + 1)) & 1u) != 0));
                uint32_t hpw = k.hw << (x + (int)kBlock), hmw = k.hw << x;     // the step's bits to bit 31
                uint64_t ph, mh, d0;
                myers_step_carry(eq0, hpw, hmw, k.pv, k.mv, ph, mh, d0);
                if (hold == (uint32_t)x % kHold) {
                    db[x / (int)kHold] = ~(eq0 ^ d0);           // diagonal predecessor valid
                    ub[x / (int)kHold] = k.pv;                  // upper predecessor valid
                }
            }
        }
    };
    uint32_t *ops = J.ops_rev + (size_t)slot * J.ops_stride;
    uint32_t i = m, j = best_j, n_rev = 0, cur_op = 3, cur_len = 0;
    auto emit = [&](uint32_t op, uint32_t len) {                // run-length CIGAR, in reverse
        if (op == cur_op) {
            cur_len += len;
        } else {
            if (cur_len && gl == 0) ops[n_rev] = (cur_len << 4) | cur_op;
            n_rev += cur_len ? 1u : 0u;
            cur_op = op;
            cur_len = len;
        }
    };
    constexpr uint32_t kNoCell = 0xFFFFFFFFu;
    while (__ballot(i > 0) != 0) {
        // the cells of this round: the one the walk stands in, then the ones its diagonal crosses
        uint32_t my_w = kNoCell, my_b = 0;
        {
            uint32_t ci = i, cj = j;
            for (uint32_t k = 0; k < n_cells && ci > 0 && cj > 0; k++) {
                const uint32_t w = (ci - 1u) >> 6, tcol = cj + lane_of(w) - 1u;
                if (cell == k) {
                    my_w = w;
                    my_b = tcol / kBlock;
                }
                // diagonal steps until the walk leaves the word, the block of STEPS (which reaches lane_of(w) columns before
                // the text's first) or the text
                const uint32_t rows = ((ci - 1u) & 63u) + 1u, cols = tcol % kBlock + 1u;
                uint32_t s = rows < cols ? rows : cols;
                s = s < cj ? s : cj;
                ci -= s;
                cj -= s;
            }
        }
        if (my_w != kNoCell) refill(my_w, my_b, load_checkpoint(my_w, my_b));
        // walk, from cell to cell while the prediction holds
        uint32_t at_cell = 0;
        uint32_t cell_w = (uint32_t)__shfl((int)my_w, (int)lane0, kWave), cell_b = (uint32_t)__shfl((int)my_b, (int)lane0, kWave);
        while (i > 0) {
            if (j == 0) {                                       // column 0: only upper predecessors, all the way
                emit(1u, i);
                i = 0;
                break;
            }
            const uint32_t wc = (i - 1u) >> 6, lc = lane_of(wc), tcol = j + lc - 1u, bc = tcol / kBlock;
            if (wc != cell_w || bc != cell_b) {                 // left the cell: into the next one held, or the round is over
                if (++at_cell >= n_cells) break;
                cell_w = (uint32_t)__shfl((int)my_w, (int)(lane0 + at_cell * kHold), kWave);
                cell_b = (uint32_t)__shfl((int)my_b, (int)(lane0 + at_cell * kHold), kWave);
                if (wc != cell_w || bc != cell_b) break;
            }
            const uint32_t first = lane0 + at_cell * kHold;     // the lanes that hold this cell's columns
            const uint32_t x = tcol % kBlock, held = x / kHold, bit = (i - 1u) & 63u;
            // A read mostly runs down the diagonal: how far from this cell?  Every lane looks at the columns it holds, at the
            // rows the diagonal through (i, j) meets them (same word only: the bit index must stay >= 0); one ballot per slot
            // gathers the answers into a mask over the block's columns, and the run is the stretch of set bits from column x
            // downwards.  The whole run is then ONE step of the walk instead of up to sixteen.
            uint32_t colmask = 0;
#pragma unroll
            for (int q = 0; q < kSlots; q++) {
                const uint32_t xq = (uint32_t)q * kHold + hold;
                const int32_t at = (int32_t)bit - (int32_t)(x - xq);
                // (columns before the text's first hold stale words: the refill only writes columns 1 .. n)
                const bool ok = xq <= x && x - xq < j && at >= 0 && ((db[q] >> (at & 63)) & 1ull) != 0;
                const uint64_t votes = __ballot(ok);
                colmask |= (uint32_t)((votes >> first) & ((1ull << kHold) - 1ull)) << ((uint32_t)q * kHold);
            }
            const uint32_t gaps = ~colmask & ((2u << x) - 1u);
            const uint32_t run = gaps ? x - (31u - (uint32_t)__builtin_clz(gaps)) : x + 1u;
            if (run) {
                emit(0u, run);
                i -= run;
                j -= run;
                continue;
            }
            // the diagonal predecessor is not valid: up, else left
            uint64_t mu = ub[0];
#pragma unroll
            for (int q = 1; q < kSlots; q++) mu = held == (uint32_t)q ? ub[q] : mu;
            const uint64_t u = shfl64(mu, (int)(first + x % kHold));
            if ((u >> bit) & 1ull) {
                emit(1u, 1u);
                i--;
            } else {
                emit(2u, 1u);
                j--;
            }
        }
    }
    if (cur_len) {
        if (gl == 0) ops[n_rev] = (cur_len << 4) | cur_op;
        n_rev++;
    }
    if (gl == 0) {
        J.out_score[a] = -best;
        J.out_begin[a] = j;
        J.out_nops[slot] = n_rev;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One alignment per LANE: queries of up to 64 * CW bases (short reads).  Sixty-four alignments share every instruction of
// the column step, against 64 / GROUP above (32 for a 300-bp read as 2 lanes x 3 words, and those 6 words for 5): nothing is
// skewed, no delta crosses lanes, the text's column is the same for the whole wave.  What the lanes do together is the
// LOADING: alignment g's query and text are read 64 bases at a time by the whole wave (one cache line per instruction
// instead of 64), and two ballots turn 64 bases into the two bit planes of their ranks: of one query word (left in lane g's
// registers) or of 64 text columns (left in LDS, 32 columns per 32-bit word, low plane then high).
// Checkpoints as above, laid out [block][word][lane]; behind them the query's planes, [plane][word][lane].  The traceback is each lane's own: the 16 columns of the current
// (word, block) are kept ROTATED -- column x's trace words turned right by x bits -- so that the cells of one diagonal sit
// at the same bit of all 16: a diagonal run is read off with one bit extraction per column, whatever lane asks for whatever
// diagonal, and no register is indexed by a value the compiler cannot see.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)src);
    return ((uint64_t)hi << 32) | lo;
}

// `old` with lane `dst`'s value replaced by v (v and dst the same in every lane)
// (v_writelane_b32 has no builtin in this compiler.  The s_nop: v is a ballot, an SGPR pair a VALU compare has just written,
// and gfx950 wants two wait states before another VALU instruction reads it -- the compiler's hazard pass does not look
// inside an asm block; nor does it see that M0, written by the SALU, is read as a lane select -- one wait state, covered.)
__device__ __forceinline__ uint64_t writelane64(uint64_t old, uint64_t v, uint32_t dst) {
    uint32_t lo = (uint32_t)old, hi = (uint32_t)(old >> 32);
    asm("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"   // (one SGPR operand each: the lane in M0)
        : "+v"(lo), "+v"(hi)
        : "s"((uint32_t)v), "s"((uint32_t)(v >> 32)), "s"(dst));   // (M0 cannot be named as clobbered; nothing else here uses it)
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t rotr64(uint64_t v, uint32_t by) { return by ? (v >> by) | (v << (64u - by)) : v; }

// bit `at` (0..63, any lane its own) of each of 16 words, gathered into bits 0..15
__device__ __forceinline__ uint32_t gather_bit(const uint64_t (&words)[kBlock], uint32_t at) {
    const bool upper = at >= 32u;
    const uint32_t sh = at & 31u;
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < (int)kBlock; q++) {
        const uint32_t half = upper ? (uint32_t)(words[q] >> 32) : (uint32_t)words[q];
        out |= ((half >> sh) & 1u) << q;
    }
    return out;
}

template <int CW>
__global__ __launch_bounds__(kWave) void bmv_align_lane_kernel(Job J) {
    extern __shared__ uint8_t lds_text[];
    const uint32_t lane = threadIdx.x, slot = blockIdx.x * kWave + lane;
    const bool have = slot < J.count;
    const uint32_t a = J.order[have ? slot : 0u];
    const uint32_t n = have ? J.text_len[a] : 0u, m = have ? J.query_len[a] : 0u;
    const uint32_t W = (m + 63u) >> 6;                          // <= CW (the host's promise)
    uint8_t *lut = lds_text;
    reinterpret_cast<uint32_t *>(lut)[lane] = reinterpret_cast<const uint32_t *>(J.lut)[lane];
    __syncthreads();
    // the text of alignment g: pairs (low plane, high plane) of 32 columns each
    auto text_of = [&](uint32_t g) { return reinterpret_cast<uint32_t *>(lds_text + 256 + (size_t)g * J.text_lds_stride); };

    uint64_t q0[CW], q1[CW];
#pragma unroll
    for (int c = 0; c < CW; c++) q0[c] = q1[c] = 0;
    {
        // Alignment g + 1's bases are on their way while alignment g's are turned into planes: CW query chunks and the
        // first CW + 1 text chunks of 64 (a text is a little longer than its query; a longer one fetches the rest in place).
        constexpr int TK = CW + 1;
        struct Raw {
            uint32_t q[CW], t[TK];
        };
        const uint64_t q_at = have ? J.query_start[a] : 0u, t_at = have ? J.text_start[a] : 0u;
        const uint32_t rc = have ? J.text_rc[a] : 0u;
        // (Loads without a condition around them -- a lane past the end reads the last base again, or the byte an empty query
        // or window begins at: the buffers have slack for that -- so that the compiler knows how many loads are in flight and
        // waiting for alignment g's bases does not wait for alignment g + 1's too.  A base in SGPRs plus a 32-bit lane offset.)
        auto fetch = [&](uint32_t g, Raw &raw) {                // (everything indexed by g is the same in every lane)
            const uint32_t mg = (uint32_t)__builtin_amdgcn_readlane((int)m, (int)g);
            const uint32_t ng = (uint32_t)__builtin_amdgcn_readlane((int)n, (int)g);
            const uint8_t *q = J.reads + readlane64(q_at, g);
            const uint8_t *src = J.genome + readlane64(t_at, g);
            const bool rcg = __builtin_amdgcn_readlane((int)rc, (int)g) != 0;
            const uint32_t q_last = mg ? mg - 1u : 0u, t_last = ng ? ng - 1u : 0u;
            const uint32_t flip = rcg ? ~0u : 0u, from = rcg ? t_last + 1u : 0u;   // (col ^ flip) + from: col, or t_last - col
#pragma unroll
            for (int c = 0; c < CW; c++) {
                const uint32_t row = (uint32_t)c * 64u + lane;
                raw.q[c] = q[row < q_last ? row : q_last];
            }
#pragma unroll
            for (int k = 0; k < TK; k++) {
                const uint32_t col = (uint32_t)k * 64u + lane;
                raw.t[k] = src[((col < t_last ? col : t_last) ^ flip) + from];
            }
        };
        auto deposit = [&](uint32_t g, const Raw &raw) {
            const uint32_t mg = (uint32_t)__builtin_amdgcn_readlane((int)m, (int)g);
            const uint32_t ng = (uint32_t)__builtin_amdgcn_readlane((int)n, (int)g);
            const bool rcg = __builtin_amdgcn_readlane((int)rc, (int)g) != 0;
            // ranks first, all of them at once (rows past a query's end and columns past a text's get whatever the table holds
            // for the stand-in byte: never looked at)
            uint32_t rq[CW], rt[TK];
#pragma unroll
            for (int c = 0; c < CW; c++) rq[c] = lut[raw.q[c]];
#pragma unroll
            for (int k = 0; k < TK; k++) rt[k] = lut[raw.t[k]] ^ (rcg ? 3u : 0u);   // (3 - r: the complement)
#pragma unroll
            for (int c = 0; c < CW; c++) {
                if ((uint32_t)c * 64u < mg) {
                    const uint64_t b0 = __ballot((rq[c] & 1u) != 0), b1 = __ballot((rq[c] & 2u) != 0);
                    q0[c] = writelane64(q0[c], b0, g);          // into lane g's registers
                    q1[c] = writelane64(q1[c], b1, g);
                }
            }
            // text window, reverse-complemented if asked (bucket_locator.h:562-567)
            uint32_t *tg = text_of(g);
            auto planes = [&](uint32_t k, uint32_t r) {
                const uint64_t lo = __ballot((r & 1u) != 0), hi = __ballot((r & 2u) != 0);
                if (lane == 0) {
                    tg[4u * k + 0u] = (uint32_t)lo;
                    tg[4u * k + 1u] = (uint32_t)hi;
                    tg[4u * k + 2u] = (uint32_t)(lo >> 32);
                    tg[4u * k + 3u] = (uint32_t)(hi >> 32);
                }
            };
#pragma unroll
            for (int k = 0; k < TK; k++) {
                if ((uint32_t)k * 64u < ng) planes((uint32_t)k, rt[k]);
            }
            if ((uint32_t)TK * 64u < ng) {
                const uint8_t *src = J.genome + readlane64(t_at, g);
                for (uint32_t k = TK; k * 64u < ng; k++) {
                    const uint32_t col = k * 64u + lane;
                    const uint32_t r = col < ng ? lut[rcg ? src[ng - 1u - col] : src[col]] : 0u;
                    planes(k, rcg ? 3u - r : r);
                }
            }
        };
        Raw even, odd;
        fetch(0u, even);
        for (uint32_t g = 0; g < kWave; g += 2u) {
            fetch(g + 1u, odd);
            deposit(g, even);
            fetch((g + 2u) % kWave, even);                      // (the last round fetches alignment 0 again, for nothing: no branch)
            deposit(g + 1u, odd);
        }
    }
    __syncthreads();
    const uint32_t *text = text_of(lane);

    // checkpoint entry of (block b, word w) of this lane's alignment
    const uint32_t TW = J.trace_words;
    uint64_t *ckpt = J.trace + (size_t)blockIdx.x * J.trace_stride;
    uint32_t *hbuf = reinterpret_cast<uint32_t *>(ckpt + (size_t)J.trace_blocks * kWave * TW * 2u);
    auto entry = [&](uint32_t b, uint32_t w) { return (b * TW + w) * kWave + lane; };
    if (J.stop_after == 1u) {                                   // (phase timing, tools/bench_verify.py: no results)
        if (have) J.out_nops[slot] = 0;
        return;
    }

    uint64_t pv[CW], mv[CW];
    uint32_t hacc_p[CW], hacc_m[CW];                            // deltas leaving each word in this block, the latest in bit 0
#pragma unroll
    for (int c = 0; c < CW; c++) {
        pv[c] = ~0ull;                                          // column 0: H[i][0] = i
        mv[c] = 0;
        hacc_p[c] = hacc_m[c] = 0;
    }
    int32_t score = (int32_t)m, best = (int32_t)m;
    uint32_t best_j = 0;
    const uint32_t last_word = W ? W - 1u : 0u, last_bit = (m - 1u) & 63u;
    uint32_t steps = W ? n : 0u, wave_words = W;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)steps, o, kWave), other_w = (uint32_t)__shfl_xor((int)wave_words, o, kWave);
        steps = other > steps ? other : steps;
        wave_words = other_w > wave_words ? other_w : wave_words;
    }
    // A lane runs as many words as the wave's longest query has: the instructions are issued for the wave anyway, rows past
    // a query's end harm nothing, and a test the whole wave agrees on is a scalar branch (a lane's own "c < W" is a saved and
    // restored exec mask per word and step, and copies of everything carried from word to word).
    wave_words = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_words);
    uint32_t tlo = 0, thi = 0;                                  // the planes of the current 32 columns
    for (uint32_t t = 1; t <= steps; t++) {                     // column t, the same for every lane
        const uint32_t xt = (t - 1u) % kBlock, x32 = (t - 1u) & 31u;
        if (W && t <= n) {
            if (x32 == 0u) {
                tlo = text[2u * ((t - 1u) >> 5)];
                thi = text[2u * ((t - 1u) >> 5) + 1u];
            }
            const uint64_t nt0 = not_plane(((tlo >> x32) & 1u) != 0), nt1 = not_plane(((thi >> x32) & 1u) != 0);
            uint32_t hpw = 0, hmw = 0;                          // row 0 is all zeros: free leading text gaps
#pragma unroll
            for (int c = 0; c < CW; c++) {
                if ((uint32_t)c >= wave_words) break;
                const uint64_t eq0 = match_rows(q0[c], q1[c], nt0, nt1);
                uint64_t ph, mh, d0;
                myers_step_carry(eq0, hpw, hmw, pv[c], mv[c], ph, mh, d0);
                if ((uint32_t)c == last_word) {
                    if (CW > 1) asm volatile("" ::: "memory");   // keeps this a branch: not to be if-converted
                    score += (int32_t)((ph >> last_bit) & 1ull) - (int32_t)((mh >> last_bit) & 1ull);
                    if (score <= best) {                        // the LAST minimum of the bottom row
                        best = score;
                        best_j = t;
                    }
                }
                hacc_p[c] = __builtin_amdgcn_alignbit(hacc_p[c], hpw, 31);     // (hacc << 1) | bit 31 of the delta word
                hacc_m[c] = __builtin_amdgcn_alignbit(hacc_m[c], hmw, 31);
            }
        }
        if (xt == kBlock - 1u || t == steps) {
            const uint32_t bt = (t - 1u) / kBlock;
#pragma unroll
            for (int c = 0; c < CW; c++) {
                if ((uint32_t)c < W && bt * kBlock < n) {
                    // step x of the block to bit 15 - x (+1) and 31 - x (-1), however many steps this lane's text had here
                    const uint32_t done = n - bt * kBlock < kBlock ? n - bt * kBlock : kBlock;
                    hbuf[entry(bt, (uint32_t)c)] = ((hacc_p[c] << (kBlock - done)) & 0xFFFFu) | (hacc_m[c] << (2u * kBlock - done));
                    uint64_t *ck = ckpt + (size_t)entry(bt + 1u, (uint32_t)c) * 2u;
                    ck[0] = pv[c];
                    ck[1] = mv[c];
                }
                hacc_p[c] = hacc_m[c] = 0;
            }
        }
    }
    // The query's planes leave the registers here: the traceback needs those of ONE word per round, and registers kept for
    // that would cost the kernel a wave per SIMD (its 16 + 16 trace words are the budget).
    uint64_t *planes_at = reinterpret_cast<uint64_t *>(hbuf + (((size_t)J.trace_blocks * kWave * TW + 1u) & ~(size_t)1)) + lane;
#pragma unroll
    for (int c = 0; c < CW; c++) {
        if ((uint32_t)c < W) {
            planes_at[(uint32_t)c * kWave] = q0[c];
            planes_at[(TW + (uint32_t)c) * kWave] = q1[c];
        }
    }
    __threadfence_block();                                      // (the traceback reads back this lane's own stores)
    if (J.stop_after == 2u) {
        if (have) J.out_nops[slot] = 0;
        return;
    }
    if (!have) return;
    if (m == 0) {                                               // H[0][j] = 0 everywhere: last column
        best = 0;
        best_j = n;
    }

    // traceback: this lane's walk, in rounds of one (word, block) -- every lane recomputes the 16 columns it stands in,
    // then walks until it leaves them
    uint64_t D[kBlock], U[kBlock];                              // rotated: bit (r - x) & 63 of [x] is row r of column x
#pragma unroll
    for (int x = 0; x < (int)kBlock; x++) D[x] = U[x] = 0;
    uint32_t *ops = J.ops_rev + (size_t)slot * J.ops_stride;
    uint32_t i = m, j = best_j, n_rev = 0, cur_op = 3, cur_len = 0;
    auto emit = [&](uint32_t op, uint32_t len) {
        if (op == cur_op) {
            cur_len += len;
        } else {
            if (cur_len) ops[n_rev++] = (cur_len << 4) | cur_op;
            cur_op = op;
            cur_len = len;
        }
    };
    while (__ballot(i > 0) != 0) {
        uint32_t wc = 0, bc = 0;
        if (i > 0 && j > 0) {
            wc = (i - 1u) >> 6;
            bc = (j - 1u) / kBlock;
            // checkpoint of (word, block): the vertical state the block starts from (block 0: column 0, H[i][0] = i) and the
            // deltas that left the word above at the same columns (row 0 above the first word: all zero)
            uint64_t kpv = ~0ull, kmv = 0ull;
            uint32_t hw = 0;
            if (bc) {
                const uint64_t *ck = ckpt + (size_t)entry(bc, wc) * 2u;
                kpv = ck[0];
                kmv = ck[1];
            }
            if (wc) hw = hbuf[entry(bc, wc - 1u)];
            const uint64_t w0 = planes_at[wc * kWave], w1 = planes_at[(TW + wc) * kWave];
            const uint32_t sh = 16u * (bc & 1u);
            const uint32_t lo16 = text[2u * (bc >> 1)] >> sh, hi16 = text[2u * (bc >> 1) + 1u] >> sh;
#pragma unroll
            for (int x = 0; x < (int)kBlock; x++) {
                if (bc * kBlock + 1u + (uint32_t)x <= n) {
                    const uint64_t eq0 = match_rows(w0, w1, not_plane(((lo16 >> x) & 1u) != 0), not_plane(((hi16 >> x) & 1u) != 0));
                    uint32_t hpw = hw << (x + (int)kBlock), hmw = hw << x;    // the step's bits to bit 31
                    uint64_t ph, mh, d0;
                    myers_step_carry(eq0, hpw, hmw, kpv, kmv, ph, mh, d0);
                    D[x] = rotr64(~(eq0 ^ d0), (uint32_t)x);    // diagonal predecessor valid
                    U[x] = rotr64(kpv, (uint32_t)x);            // upper predecessor valid
                }
            }
        }
        while (i > 0 && (j == 0 || (((i - 1u) >> 6) == wc && (j - 1u) / kBlock == bc))) {
            if (j == 0) {                                       // column 0: only upper predecessors, all the way
                emit(1u, i);
                i = 0;
                break;
            }
            // diagonal first, as far as it goes: the cells (i - s, j - s) sit at bit `at` of columns x - s
            const uint32_t x = (j - 1u) % kBlock, bit = (i - 1u) & 63u, at = (bit - x) & 63u;
            const uint32_t room = (x < bit ? x : bit) + 1u;     // cells of this diagonal inside the word and the block
            const uint32_t gaps = ~gather_bit(D, at) & ((2u << x) - 1u);
            uint32_t run = gaps ? x - (31u - (uint32_t)__builtin_clz(gaps)) : x + 1u;
            run = run < room ? run : room;
            if (run) {
                emit(0u, run);
                i -= run;
                j -= run;
            }
            if (run < room) {                                   // stopped by a cell whose diagonal predecessor is not valid
                if ((gather_bit(U, at) >> (x - run)) & 1u) {
                    emit(1u, 1u);
                    i--;
                } else {
                    emit(2u, 1u);
                    j--;
                }
            }
        }
    }
    if (cur_len) ops[n_rev++] = (cur_len << 4) | cur_op;
    J.out_score[a] = -best;
    J.out_begin[a] = j;
    J.out_nops[slot] = n_rev;
}

}  // namespace bmv
