/*
 * bm_align_oracle.h -- CPU restatement of bucketmap_align's alignment verification (TEST INFRASTRUCTURE).
 *
 * Oracle for the GPU verifier (include/bmv.h).  Restates, in plain C, what the BM_ALIGN branch of
 * bucket_locator::locate asks of SeqAn3 (bucket_map/locator/bucket_locator.h:520-528, 560-589):
 *
 *   align_pairwise(text, query) with method_global{free_end_gaps_sequence1_leading, ..._trailing} and
 *   edit_scheme, i.e. the query is aligned end to end against the best substring of the text, unit costs;
 *   outputs: score (= -edit distance), sequence1_begin_position, the alignment as a CIGAR.
 *
 * PARITY UNPINNED, and more so than the other oracles: the arithmetic lives in SeqAn3 (fetched at the
 * floating tag `main`, bucket_map/CMakeLists.txt:69-80, absent from this machine) and the reference has no
 * test or fixture for it.  The SCORE is unique by definition and is pinned by the dual formulation below
 * (full dynamic-programming matrix here, Myers bit-vectors on the GPU) and by hand-worked cases in
 * tests/test_align.py.  Among equally good alignments SeqAn3's choice is ASSUMED to be:
 *   (1) end column: the LAST column of the bottom row that attains the minimum;
 *   (2) traceback: a cell's predecessors are tried diagonal first, then up (a query base against a gap,
 *       CIGAR I), then left (a text base against a gap, CIGAR D);
 *   (3) CIGAR alphabet M/I/D (cigar_from_alignment without extended_cigar).
 * Those three are stated in one place (here) so that a maintainer with SeqAn3 at hand can correct them.
 */
#ifndef BM_ALIGN_ORACLE_H
#define BM_ALIGN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CIGAR operation codes of the packed form (len << 4 | op), as in BAM. */
enum { BMAO_OP_M = 0, BMAO_OP_I = 1, BMAO_OP_D = 2 };

/* One alignment.  text / query: ASCII bases (dna4 folding as everywhere: N and friends -> A ...); when
 * text_rc != 0 the text is reverse-complemented first (bucket_locator.h:563-567).  Writes score
 * (<= 0), begin (0-based index in the -- possibly reverse-complemented -- text of the first aligned text
 * base) and up to cigar_cap packed CIGAR entries; returns the number of CIGAR entries (which can exceed
 * cigar_cap: nothing past the cap is written), or -1 if out of memory. */
int bmao_align(const uint8_t *text, uint32_t n, int text_rc, const uint8_t *query, uint32_t m, int32_t *out_score,
               uint32_t *out_begin, uint32_t *out_cigar, uint32_t cigar_cap);

/* Batch with the buffer layout of bmv_align (include/bmv.h): alignment a aligns
 * reads[query_start[a] .. +query_len[a]) against genome[text_start[a] .. +text_len[a]).
 * out_cigar_offset has n + 1 entries; out_cigar holds cigar_cap entries.  Returns 0, or 1 if cigar_cap was
 * too small (out_cigar_offset is still complete, so the caller can size the buffer and call again). */
int bmao_align_batch(const uint8_t *genome, const uint8_t *reads, const uint64_t *text_start, const uint32_t *text_len,
                     const uint8_t *text_rc, const uint64_t *query_start, const uint32_t *query_len, uint32_t n,
                     int32_t *out_score, uint32_t *out_begin, uint64_t *out_cigar_offset, uint32_t *out_cigar,
                     uint64_t cigar_cap);

/* O(n)-memory CHECK of a reported alignment (thousands of 10-kbp alignments without the 440 MB matrix):
 *   (a) the optimal score by a two-row dynamic programme over the same recurrence (no traceback);
 *   (b) the reported CIGAR walked from `begin` over text and query: it must consume the whole query, stay inside the
 *       text, and cost exactly -score edits (mismatches under M, one per I and D base).
 * Returns 0 when score == optimum and the CIGAR is a valid alignment of that cost; otherwise a bit set:
 *   1 score differs from the optimum, 2 the CIGAR does not consume the query exactly / leaves the text,
 *   4 the CIGAR's cost differs from -score, 8 out of memory.  *out_optimum receives the optimum (<= 0). */
int bmao_check(const uint8_t *text, uint32_t n, int text_rc, const uint8_t *query, uint32_t m, int32_t score, uint32_t begin,
               const uint32_t *cigar, uint64_t n_cigar, int32_t *out_optimum);

/* bmao_check for alignments [first, last) of a batch laid out as bmao_align_batch's; out_bad[a] = its result. */
void bmao_check_batch(const uint8_t *genome, const uint8_t *reads, const uint64_t *text_start, const uint32_t *text_len,
                      const uint8_t *text_rc, const uint64_t *query_start, const uint32_t *query_len, uint32_t first,
                      uint32_t last, const int32_t *score, const uint32_t *begin, const uint64_t *cigar_offset,
                      const uint32_t *cigar, uint8_t *out_bad);

#ifdef __cplusplus
}
#endif
#endif
