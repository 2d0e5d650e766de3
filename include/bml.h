/*
 * bml.h -- C ABI of the MI355X locator candidate scan ("bucket-map locate", part of libbmf.so).
 *
 * Second hot path of BucketMap.  For every candidate
 * (window, bucket, strand) produced by the filter it replaces
 *
 *   reference (bucket_map/locator/bucket_locator.h)            this ABI
 *   --------------------------------------------------------  ---------------------------------
 *   bucket_locator ctor: allowed_mismatch/indel   :419-420     bml_create
 *   _initialize_kmer_index (bucket sequences)     :151-160     bml_load_genome
 *   _prepare_read_query (k-mer sampling)          :292-347     bml_sample_windows
 *   _create_kmer_index + _find_offset, per bucket :162-177,    bml_locate  (all candidates of one
 *     and per candidate, inside _locate's loop    :209-290,      _locate call in one batch)
 *                                                 :651-695
 *
 * What stays on the host (bucket-map_amd/host/bucket_locator.h): FASTQ parsing and windowing, the order in
 * which results are appended per read (:651-693), _filter_best_locations (:350-405) and SAM output (:455-611).
 *
 * Instead of one hash multimap per bucket (65 k node allocations per bucket in the reference) the
 * device scans each candidate bucket against the few hundred k-mers actually asked of it, writes every candidate's
 * occurrences to a segment of their own (grouped by sample) and replays the order-dependent vote of _find_offset exactly -- per sample,
 * occurrences in the order libstdc++'s unordered_multimap::equal_range yields them, i.e. DESCENDING bucket
 * offset: one thread per candidate orders and replays a handful of occurrences; a candidate with many (a k-mer
 * of a tandem repeat occurs thousands of times in a bucket) gets a workgroup and dense bitmaps of the start
 * positions, an order-free statement of the same vote (bml_kernels.hip.h).
 *
 * Conventions as in bmf.h: plain C types, int status, message in bml_last_error(), no CPU fallback.
 */
#ifndef BML_H
#define BML_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { BML_OK = 0, BML_ERR_ARG = 1, BML_ERR_HIP = 2, BML_ERR_STATE = 3, BML_ERR_UNSUPPORTED = 5 };

typedef struct bml_params {
    uint32_t k;                /* seed length (-l), <= 16                                         */
    uint32_t num_samples;      /* p (-p): sampled k-mers per window, <= 64                        */
    int32_t  allowed_mismatch; /* ceil(e*p), float32 (bucket_locator.h:419)                       */
    int32_t  allowed_indel;    /* ceil(n*read_len), float32 (bucket_locator.h:420)                */
    uint32_t max_bucket_bases; /* longest bucket incl. overlap: bucket_len + read_len             */
    int32_t  device;
} bml_params;

typedef struct bml_ctx bml_ctx;

const char *bml_last_error(void);
int  bml_create(const bml_params *params, bml_ctx **out);
void bml_destroy(bml_ctx *ctx);

/* The reference genome as one byte string (ASCII, any record concatenation) and the kept buckets as
 * (start, length) views into it (iterate_through_buckets, utils.h:72-97).  Uploaded once. */
int  bml_load_genome(bml_ctx *ctx, const uint8_t *bases, uint64_t n_bases, const uint64_t *bucket_start,
                     const uint32_t *bucket_len, uint32_t n_buckets);

/* The same for a genome whose records are buffers of their own on the host: they go to the device back to back, in
 * order, through page-locked pieces -- the caller needs no flattened copy; bucket_start counts in that concatenation. */
int  bml_load_genome_records(bml_ctx *ctx, const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records,
                             const uint64_t *bucket_start, const uint32_t *bucket_len, uint32_t n_buckets);

/* _prepare_read_query (:292-347) for a batch of windows: window w is the view [win_start[w], +win_len[w]) of
 * `bases` (ASCII, dna4 folding) and `quals` (phred+33), as in bmf_map_windows.  Per window: the k-mers whose
 * quality sum over their k bases is >= min_base_quality (:325-327; all k-mers if none is, :330-332),
 * Sampler(p) over them (utils.h:160-178, tabulated in fp64 on the host), and for each sampled k-mer
 *   out_hash[w*p + s]  its hash,   out_pos[w*p + s]  its start in the window (u16, as the reference stores it);
 *   out_has[w] = 0 for a window shorter than k (its entries are zeros and it must not be located).
 * The arrays are exactly bml_locate's sample_hash / sample_pos. */
int  bml_sample_windows(bml_ctx *ctx, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                        const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows,
                        uint32_t min_base_quality, uint32_t *out_hash, uint16_t *out_pos, uint8_t *out_has);

/* The same for windows whose bases and qualities lie apart in one buffer (a FASTQ text, as bmf_map_text_windows_compact
 * takes them): window w = text[seq_start[w] .. +win_len[w]) with qualities text[qual_start[w] .. +win_len[w]).  The
 * library gathers the windows into page-locked buffers piece by piece (a few host threads), so the caller copies nothing. */
int  bml_sample_text_windows(bml_ctx *ctx, const uint8_t *text, uint64_t n_bytes, const uint64_t *seq_start,
                             const uint64_t *qual_start, const uint32_t *win_len, uint32_t n_windows,
                             uint32_t min_base_quality, uint32_t *out_hash, uint16_t *out_pos, uint8_t *out_has);

/* One batch of candidates.
 *   windows : sample_hash[w*p + s], sample_pos[w*p + s] (start of the k-mer in the window, u16 as
 *             query_sequences_storage keeps it), seg_len[w] = length of the window
 *   pairs   : pair_bucket[i], pair_window[i], pair_rc[i] (1 = reverse-complement candidate).  MUST be
 *             grouped by bucket (all pairs of one bucket adjacent); order inside a group is free.
 *   outputs : out_offset[i] = winning start offset in the bucket (as _find_offset returns it: can be 0;
 *             the caller keeps only > 0) or -1; out_votes[i] = its votes or 0. */
int  bml_locate(bml_ctx *ctx, const uint32_t *sample_hash, const uint16_t *sample_pos, const uint32_t *seg_len,
                uint32_t n_windows, const uint32_t *pair_bucket, const uint32_t *pair_window, const uint8_t *pair_rc,
                uint32_t n_pairs, int32_t *out_offset, uint32_t *out_votes);

/* Times of the last bml_locate in ms: ms_scan = its scan kernels, ms_replay = its vote kernels (HIP events on the
 * context's stream), ms_host = what the call spent outside those kernels from the first scan on (stream syncs, the
 * count downloads, growth of the occurrence buffer, placing the groups' segments) -- and the k-mer occurrences it handled. */
int  bml_last_stats(bml_ctx *ctx, float *ms_scan, float *ms_host, float *ms_replay, uint64_t *n_occurrences);

/* Candidates of the last bml_locate that had more occurrences than one thread replays (repeats): they went through
 * the workgroup-per-candidate kernel. */
int  bml_last_heavy_candidates(bml_ctx *ctx, uint32_t *n_heavy);

/* Statistics of the last bml_locate (n_pairs = its candidate count): candidates[b] = candidates whose occurrence count lies in
 * [2^(b-1), 2^b) (b = 0: none), occurrences[b] = the occurrences they hold.  33 entries each. */
int  bml_last_count_histogram(bml_ctx *ctx, uint32_t n_pairs, uint64_t *candidates, uint64_t *occurrences);

#ifdef __cplusplus
}
#endif
#endif
