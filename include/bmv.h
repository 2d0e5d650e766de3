/*
 * bmv.h -- C ABI of the MI355X alignment verifier ("bucket-map verify", part of libbmf.so).
 *
 * the `bucketmap_align` build of the reference (BM_ALIGN, CMakeLists.txt:138) sends
 * every located candidate through a SeqAn3 pairwise alignment before it is written to the SAM file:
 *
 *   reference (bucket_map/locator/bucket_locator.h)               this ABI
 *   ------------------------------------------------------------  -------------------------------------
 *   align_config: method_global, sequence1 end gaps free,         the only configuration the kernel has
 *     edit_scheme, output score/begin/alignment       :520-528
 *   text window = bucket_seq[bucket][offset, +width)  :549-550    (text_start, text_len) views, chosen by
 *   reverse complement of the text for strand 16      :562-567      the caller; text_rc
 *   align_pairwise(text, query)                       :569        bmv_align, all candidates in one batch
 *   alignment.score(), sequence1_begin_position(),    :570-576    out_score, out_begin, CIGAR entries
 *     cigar_from_alignment
 *
 * What stays on the host (bucket-map_amd/host/bucket_locator.h): the window arithmetic (:550), the
 * MAPQ = 60 + score rule and its threshold (:570-573), SAM output.
 *
 * Semi-global edit distance (the whole query against the best substring of the text, unit costs) by
 * Myers' bit-vector recurrence in words of 64 query rows: a query of up to 512 bases is one lane's work (64 alignments per
 * wave), a longer one is spread over a group of lanes skewed along the text (beyond 32 768 bases in strips of that many
 * rows); the traceback runs on the device too.  The score is unique; between equally good alignments SeqAn3's own choice is not
 * pinned by anything in the reference (no test, no fixture, SeqAn3 itself absent), so the rules are stated
 * here and a maintainer with SeqAn3 at hand can correct them in one place:
 *   (1) the alignment ends at the LAST text column whose bottom-row score is the minimum;
 *   (2) the traceback prefers the diagonal predecessor, then the upper one (a query base against a gap,
 *       CIGAR I), then the left one (a text base against a gap, CIGAR D);
 *   (3) CIGAR alphabet M / I / D (cigar_from_alignment without extended_cigar).
 *
 * Conventions as in bmf.h: plain C types, int status, message in bmv_last_error(), no CPU fallback.
 */
#ifndef BMV_H
#define BMV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { BMV_OK = 0, BMV_ERR_ARG = 1, BMV_ERR_HIP = 2, BMV_ERR_STATE = 3, BMV_ERR_UNSUPPORTED = 5 };

/* CIGAR entries are packed as in BAM: length << 4 | op. */
enum { BMV_OP_M = 0, BMV_OP_I = 1, BMV_OP_D = 2 };

typedef struct bmv_params {
    uint32_t max_query_len;   /* longest read handed to bmv_align (<= 65536)                        */
    uint32_t max_text_len;    /* longest text window: max_query_len + 1 + indel allowance (<= 81920) */
    int32_t  device;
} bmv_params;

typedef struct bmv_ctx bmv_ctx;

const char *bmv_last_error(void);
int  bmv_create(const bmv_params *params, bmv_ctx **out);
void bmv_destroy(bmv_ctx *ctx);

/* The reference genome as one byte string (ASCII; the same string bml_load_genome takes).  Uploaded once;
 * text windows are views into it. */
int  bmv_load_genome(bmv_ctx *ctx, const uint8_t *bases, uint64_t n_bases);
/* ... or from records that are buffers of their own on the host (concatenated on the device, in order; bml_load_genome_records). */
int  bmv_load_genome_records(bmv_ctx *ctx, const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records);

/* One batch of alignments: alignment a aligns the query reads[query_start[a], +query_len[a]) against the
 * text genome[text_start[a], +text_len[a]), reverse-complemented first when text_rc[a] != 0.  `reads` is
 * n_read_bytes of ASCII bases (dna4 folding as in bmf.h).  Results stay on the device until
 * bmv_results; *total_cigar receives the number of CIGAR entries of the whole batch. */
int  bmv_align(bmv_ctx *ctx, const uint8_t *reads, uint64_t n_read_bytes, const uint64_t *text_start,
               const uint32_t *text_len, const uint8_t *text_rc, const uint64_t *query_start,
               const uint32_t *query_len, uint32_t n, uint64_t *total_cigar);

/* Results of the last bmv_align:
 *   out_score[a]        alignment.score() = -(edit distance)                       (bucket_locator.h:570)
 *   out_begin[a]        alignment.sequence1_begin_position(), 0-based in the text  (:576)
 *   out_cigar_offset    n + 1 entries; alignment a owns out_cigar[offset[a] .. offset[a+1])
 *   out_cigar           total_cigar packed entries, in alignment order, 5' to 3' of the query */
int  bmv_results(bmv_ctx *ctx, int32_t *out_score, uint32_t *out_begin, uint64_t *out_cigar_offset,
                 uint32_t *out_cigar);

/* Kernel time of the last bmv_align in ms (edit-distance columns + traceback, all chunks) and the number
 * of dynamic-programming cells it stands for (sum of query_len * text_len). */
int  bmv_last_stats(bmv_ctx *ctx, float *ms_kernels, uint64_t *n_cells);

#ifdef __cplusplus
}
#endif
#endif
