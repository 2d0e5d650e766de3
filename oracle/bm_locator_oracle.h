/*
 * bm_locator_oracle.h -- CPU restatement of bucket_locator's candidate scan (TEST INFRASTRUCTURE).
 *
 * Oracle for the GPU locator scan (include/bml.h).  PARITY UNPINNED, like bm_oracle.h: the reference
 * cannot be built here and has no fixtures for this path.  Restates, in plain C:
 *   _create_kmer_index   bucket_map/locator/bucket_locator.h:162-177
 *   _find_offset         bucket_map/locator/bucket_locator.h:209-290
 * One assumption is imported from the C++ library rather than from the reference's text: libstdc++'s
 * std::unordered_multimap::equal_range visits equal keys in REVERSE insertion order, i.e. in
 * DESCENDING bucket offset for the reference's ascending inserts.
 * tests/cpp/umm_order.cpp checks that claim against the real container of this toolchain.
 */
#ifndef BM_LOCATOR_ORACLE_H
#define BM_LOCATOR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bmlo_params {
    uint32_t k;                /* seed length (-l)                                             */
    uint32_t num_samples;      /* p (-p)                                                       */
    int32_t  allowed_mismatch; /* ceil(e*p) in float32 (bucket_locator.h:419)                  */
    int32_t  allowed_indel;    /* ceil(n*read_len) in float32 (bucket_locator.h:420)           */
} bmlo_params;

typedef struct bmlo_bucket_index bmlo_bucket_index;

/* _create_kmer_index: every k-mer of the bucket (ASCII bases, dna4 folding), offset = start position */
bmlo_bucket_index *bmlo_index_bucket(const uint8_t *bases_ascii, uint32_t len, uint32_t k);
void bmlo_index_free(bmlo_bucket_index *ix);

/* _find_offset for one (window, bucket, strand).  kmers/indices: the window's p sampled k-mer hashes
 * and their start positions in the window (query_sequences_storage).  Writes the winning start offset
 * (can be <= 0; the CALLER keeps only > 0, bucket_locator.h:674,686) and its votes, or -1 / 0. */
void bmlo_find_offset(const bmlo_params *p, const bmlo_bucket_index *ix, const uint32_t *kmers,
                      const uint16_t *indices, uint32_t segment_length, int reverse_complement,
                      int32_t *out_offset, uint32_t *out_votes);

/* Batch form with the buffer layout of bml_locate (include/bml.h).  Pairs must be grouped by bucket. */
int bmlo_locate(const bmlo_params *p, const uint8_t *genome, const uint64_t *bucket_start,
                const uint32_t *bucket_len, uint32_t n_buckets, const uint32_t *sample_hash,
                const uint16_t *sample_pos, const uint32_t *seg_len, const uint32_t *pair_bucket,
                const uint32_t *pair_window, const uint8_t *pair_rc, uint32_t n_pairs, int32_t *out_offset,
                uint32_t *out_votes);

/* _prepare_read_query (bucket_map/locator/bucket_locator.h:292-347) for a batch of windows, buffer layout
 * of bml_sample_windows (include/bml.h): window w is the view [win_start[w], +win_len[w]) of `bases` (ASCII)
 * and `quals` (phred+33).  Per window: the k-mers whose quality sum over their k bases is >=
 * min_base_quality (:325-327; all k-mers if none is, :330-332), Sampler(p) over them (:333-335, utils.h:
 * 160-178 in fp64), and for each sampled k-mer its hash and its start position in the window.
 * out_has[w] = 0 for a window shorter than k (its p entries are zeros). */
void bmlo_sample_windows(uint32_t k, uint32_t p, uint32_t min_base_quality, const uint8_t *bases, const uint8_t *quals,
                         const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows, uint32_t *out_hash,
                         uint16_t *out_pos, uint8_t *out_has);

#ifdef __cplusplus
}
#endif
#endif
