/*
 * bmf.h -- C ABI of the MI355X candidate-bucket filter ("bucket-map filter", libbmf.so).
 *
 * This is the drop-in boundary for BucketMap's per-read candidate-bucket filter.  It replaces what
 * the reference's abstract `mapper` interface (bucket_map/mapper/mapper.h:4-34) hides behind
 * `q_gram_mapper<NB>` (bucket_map/mapper/q_gram_mapper.h:204-646):
 *
 *   reference                                               this ABI
 *   ------------------------------------------------------  -------------------------------------
 *   q_gram_mapper<NB>::q_gram_mapper(...)   :281-308         bmf_create
 *   mapper::load(index_dir, indicator)      :318-372         bmf_load_index_files / bmf_load_index
 *   q_gram_mapper::query_sequence(seq,qual) :414-480         bmf_map_windows (batched, both strands)
 *   mapper::reset()                         :638-645         bmf_reset
 *   ~q_gram_mapper()                        :310-315         bmf_destroy
 *
 * `mapper::map(fastq)` (:483-557) itself -- FASTQ parsing, long-read windowing and the scatter of
 * (read, window) pairs into per-bucket lists -- stays on the host in C++
 * (bucket-map_amd/host/gpu_q_gram_mapper.h) and calls bmf_map_windows for the arithmetic.
 *
 * Conventions: plain C types only; every function returns 0 (BMF_OK) or a BMF_ERR_* code and never
 * throws; one context per device, calls on one context are serialised by the caller; all buffers
 * are caller-owned unless stated.  There is NO CPU fallback: every entry point that computes needs
 * a gfx950 device and fails with BMF_ERR_HIP if none is usable.
 */
#ifndef BMF_H
#define BMF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMF_ABI_VERSION 1

enum {
    BMF_OK = 0,
    BMF_ERR_ARG = 1,         /* bad argument (null pointer, k<q, window longer than read_len ...) */
    BMF_ERR_HIP = 2,         /* a HIP runtime call failed / no device                             */
    BMF_ERR_STATE = 3,       /* e.g. map before load, load while an index is loaded               */
    BMF_ERR_IO = 4,          /* index files missing or short                                      */
    BMF_ERR_UNSUPPORTED = 5  /* parameter outside what the kernels were built for                 */
};

/* Run-time form of q_gram_mapper's constructor arguments (q_gram_mapper.h:281-286) plus NB, which
 * is a compile-time template parameter in the reference (BM_BUCKET_NUM).  Derived values
 * (num_fault, threshold, min_base_quality) are passed already derived, exactly as main.cpp:202-209
 * and q_gram_mapper.h:163,303 compute them (float32!); helpers below do the derivation. */
typedef struct bmf_params {
    uint32_t num_buckets;         /* NB                                                          */
    uint32_t q;                   /* index seed length (-k)                                      */
    uint32_t k;                   /* query seed length (-l), q <= k <= 16                        */
    uint32_t num_samples;         /* S (-s), 1..64                                               */
    uint32_t num_fault;           /* F = ceil(S*e) in float32, 1..31                             */
    uint32_t threshold;           /* (unsigned)(d*NB) in float32                                 */
    uint32_t min_base_quality;    /* b*k                                                         */
    uint32_t max_candidates;      /* 30 in the reference; 1..64                                  */
    uint32_t read_len;            /* -r: longest window handed to bmf_map_windows                */
    uint32_t num_segment_samples; /* 5 in the reference (used by the host wrapper only)          */
    int32_t  device;              /* HIP device ordinal                                          */
    uint32_t flags;               /* BMF_FLAG_* (0 = behave byte for byte like round 1's default)  */
} bmf_params;

/* Exact pruning: skip index bytes that cannot change the outputs.  A bucket with >= F misses is in no level
 * of the reference's filter whatever the remaining samples are, so a wave stops when every bucket is dead
 * and stops loading the 128-bucket chunks that hold no live bucket; where the index is sparse enough a first
 * pass over a few rows per sample finds the live chunks and only those are counted exactly.  The outputs
 * are identical; only the number of row bytes actually read (not the algorithmic row count) drops. */
#define BMF_FLAG_EARLY_EXIT 1u

typedef struct bmf_ctx bmf_ctx;
typedef struct bmf_batch bmf_batch;

/* float32 parameter derivations (main.cpp:207, q_gram_mapper.h:163, bucket_locator.h:419-420) */
uint32_t bmf_fault_from_rate(uint32_t samples, float max_error_rate);
uint32_t bmf_threshold(float distinguishability, uint32_t num_buckets);
uint32_t bmf_ceil_mul_f32(float rate, uint32_t n);

int  bmf_abi_version(void);
/* Message of the last failure on this thread (create failures included). Never NULL. */
const char *bmf_last_error(void);

/* q_gram_mapper ctor (q_gram_mapper.h:281-308). Selects the device, creates the stream, uploads
 * the sampler position table (utils.h:160-178 tabulated in fp64 on the host). */
int  bmf_create(const bmf_params *params, bmf_ctx **out);
void bmf_destroy(bmf_ctx *ctx);

/* mapper::load (q_gram_mapper.h:318-372) from memory: `rows` = n_rows x ceil(NB/8) bytes in the
 * .qgram layout (bucket j <-> byte j>>3, bit j&7), `kmer_to_index` = the .kmers_index table
 * (n_kmers = 4^q entries, -1 = q-gram not indexed).  Uploads to HBM (rows padded to 128-byte
 * pitch), computes per-row zero counts and the "highly distinguishable q-gram" bitmap on device
 * (q_gram_mapper.h:171-196).  BMF_ERR_STATE if an index is already loaded (:325-328). */
int  bmf_load_index(bmf_ctx *ctx, const uint8_t *rows, uint64_t n_rows, const int32_t *kmer_to_index,
                    uint64_t n_kmers);
/* GPU form of the host indexer (bucket_indexer::index / _insert_into_bucket,
 * bucket_map/indexer/bucket_indexer.h:49-61,170-216): builds the index rows straight in HBM from the
 * genome (one byte string, ASCII) and the kept buckets given as (start, length) views
 * (iterate_through_buckets, utils.h:72-97), for the q-grams kmer_to_index keeps (numbered 0,1,2,... in
 * ascending hash, bucket_indexer.h:147-157).  Leaves the context loaded, exactly as bmf_load_index would
 * with the rows the host indexer writes.  Optional: indexing stays a host job in the reference, this is
 * the fast path for benchmarks and for `bucketmap -x`.  3 <= q <= 10. */
int  bmf_build_index(bmf_ctx *ctx, const uint8_t *genome, uint64_t n_bases, const uint64_t *bucket_start,
                     const uint32_t *bucket_len, uint32_t n_buckets, const int32_t *kmer_to_index,
                     uint64_t n_kmers);
/* Copies the loaded index back in the .qgram layout (n_rows x ceil(NB/8) bytes); rows_out may be NULL to
 * query n_rows only. */
int  bmf_index_download(bmf_ctx *ctx, uint8_t *rows_out, uint64_t *n_rows_out);
/* Same, reading <index_dir>/<indicator>.kmers_index and .qgram (q_gram_mapper.h:331-358). */
int  bmf_load_index_files(bmf_ctx *ctx, const char *index_dir, const char *indicator);
/* mapper::reset (q_gram_mapper.h:638-645): frees the index in HBM; the context stays usable. */
int  bmf_reset(bmf_ctx *ctx);
/* Per-row zero counts as distinguishability_filter::read computes them (:171-187); n_rows u32. */
int  bmf_index_zeros(bmf_ctx *ctx, uint32_t *out_zeros);

/* q_gram_mapper::map's windowing rule (q_gram_mapper.h:510-516): a record longer than 2*read_len is cut
 * into n_seg windows starting at Sampler(n_seg) positions over [0, len-read_len-1]; otherwise one
 * window at 0.  Writes the window starts, returns their number (1 or n_seg).  Pure host helper. */
uint32_t bmf_window_starts(uint32_t record_len, uint32_t read_len, uint32_t n_seg, uint32_t *out);

/* q_gram_mapper::query_sequence (q_gram_mapper.h:414-480) for a batch of windows, host buffers.
 * `bases` (ASCII, dna4 folding as SeqAn3) and `quals` (phred+33) hold n_bytes bytes of reads; window w
 * is the view [win_start[w], win_start[w] + win_len[w]) of both.  Windows may overlap (the 5 windows
 * of a long read) or be shorter than their read (truncation to read_len, q_gram_mapper.h:521); every
 * window must be <= read_len long and lie inside the buffers.
 *   out_counts[2w]   = number of candidate buckets, read as-is         (<= max_candidates)
 *   out_counts[2w+1] = number of candidate buckets, reverse complement
 *   out_buckets[(2w+o)*max_candidates + i] = i-th bucket id, ascending (entries >= count untouched)
 * Synchronous: returns when the outputs are in host memory. */
int  bmf_map_windows(bmf_ctx *ctx, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                     const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows,
                     uint32_t *out_counts, uint32_t *out_buckets);

/* Same computation, packed output -- what the host mapper uses: the lists hold < 1 id on average, so
 * max_candidates slots per list is mostly air.  out_counts as above; out_ids receives the lists back to back in
 * window order (read as-is, then reverse complement, per window), *n_ids their total.  ids_capacity is the room
 * in out_ids; 2 * n_windows * max_candidates always suffices (BMF_ERR_ARG when it was too small). */
int  bmf_map_windows_compact(bmf_ctx *ctx, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                             const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows,
                             uint32_t *out_counts, uint32_t *out_ids, uint64_t ids_capacity, uint64_t *n_ids);

/* The same call for reads that do NOT lie as bmf_map_windows wants them: a FASTQ text, where a read's bases and its
 * qualities are apart.  Window w = text[seq_start[w] .. +win_len[w]) with qualities text[qual_start[w] .. +win_len[w]).
 * The library gathers the windows of a piece back to back into page-locked buffers of its own (a few host threads,
 * BMF_GATHER_THREADS) while the piece before is on the device, so the caller copies nothing: what `bucketmap` calls with
 * the memory-mapped FASTQ file.  Outputs as bmf_map_windows_compact. */
int  bmf_map_text_windows_compact(bmf_ctx *ctx, const uint8_t *text, uint64_t n_bytes, const uint64_t *seq_start,
                                  const uint64_t *qual_start, const uint32_t *win_len, uint32_t n_windows,
                                  uint32_t *out_counts, uint32_t *out_ids, uint64_t ids_capacity, uint64_t *n_ids);

/* Optional: sizes the staging buffers of bmf_map_windows[_compact] (text_windows = 0) or bmf_map_text_windows_compact
 * (text_windows = 1) for calls of up to max_windows_per_call windows of read_len bases NOW -- a mapper calls it when it
 * loads its index, so that its first batch does not pay for page-locking and device allocations.  Buffers still grow on
 * demand. */
int  bmf_map_reserve(bmf_ctx *ctx, uint32_t max_windows_per_call, int text_windows);

/* Device-resident form (benchmarks, pipelines): upload once, run many times, download when needed. */
int  bmf_batch_create(bmf_ctx *ctx, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                      const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows, bmf_batch **out);
int  bmf_batch_run(bmf_ctx *ctx, bmf_batch *batch);            /* async on the context's stream */
int  bmf_batch_download(bmf_ctx *ctx, bmf_batch *batch, uint32_t *out_counts, uint32_t *out_buckets);
/* Number of index rows ANDed by the last run (reference row reads, q_gram_mapper.h:405, both
 * orientations, all windows) -- the unit of the algorithmic-bytes figure. Synchronises. */
int  bmf_batch_rows_anded(bmf_ctx *ctx, bmf_batch *batch, uint64_t *out);
void bmf_batch_destroy(bmf_ctx *ctx, bmf_batch *batch);
int  bmf_sync(bmf_ctx *ctx);

/* HIP-event profiling of bmf_batch_run on the context's own stream: after bmf_profile_begin every
 * run records events around the sample kernel and the vote kernel; bmf_profile_end synchronises
 * and returns per-run kernel durations in milliseconds (arrays of max_runs floats). */
int  bmf_profile_begin(bmf_ctx *ctx, uint32_t max_runs);
int  bmf_profile_end(bmf_ctx *ctx, uint32_t *n_runs, float *ms_sample, float *ms_vote);

/* Page-locked host memory for read staging buffers (hipHostMalloc): copies from it to the device run at
 * link speed.  Optional: every entry point accepts ordinary memory too. */
int  bmf_pinned_alloc(size_t bytes, void **out);
void bmf_pinned_free(void *p);

/* Free and total memory of a device (hipMemGetInfo): the tools print what is in use when they are done -- the
 * contexts' buffers only grow, so that is the run's peak but for buffers that were re-allocated larger. */
int  bmf_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* Introspection for DESIGN.md / bench: bytes per padded row in HBM, kernel variant chosen. */
int  bmf_info(bmf_ctx *ctx, uint32_t *row_pitch_bytes, uint32_t *chunks_per_lane, uint32_t *planes,
              uint32_t *rows_in_flight);
/* BMF_FLAG_EARLY_EXIT only: index rows per sample the first pass of the two-pass pruning kernel streams
 * for the loaded index; 0 = the single-pass pruning kernel (or no pruning) serves it. */
int  bmf_pass1_rows(bmf_ctx *ctx, uint32_t *out);
/* Same flag: the form of that first pass.  fold = 2 or 4: it streams a FOLDED copy of the index (one bit per group of
 * `fold` buckets, 1/fold of the bytes per row, built at load time) and reads `rows` of its rows per sample; fold = 1:
 * it streams `rows` rows of the index itself (0 when another kernel serves the index). */
int  bmf_pass1_fold(bmf_ctx *ctx, uint32_t *fold, uint32_t *rows);
/* Two-pass pruning, after a run of `batch`: how many (window, orientation) items still had a live bucket after
 * the first pass and went to the packed recount kernel, and how many took the slow full-width path.  Both 0
 * when another kernel served the run.  Synchronises. */
int  bmf_batch_pass2_counts(bmf_ctx *ctx, bmf_batch *batch, uint32_t *recounted, uint32_t *slow);
/* Same run: 16-byte column loads the recount kernel issued (each one its own 64-byte sector of a random row) --
 * the unit its HBM traffic is priced in.  Synchronises. */
int  bmf_batch_recount_loads(bmf_ctx *ctx, bmf_batch *batch, uint64_t *loads);
/* Same flag, statistics of the last two-pass run: stored[n] = items that left pass 1 with n stored chunks (n = 0..32;
 * [33] = items that went to the slow kernel), lowest[n] = items with n chunks at their lowest level.  34 entries each. */
int  bmf_batch_live_histogram(bmf_ctx *ctx, bmf_batch *batch, uint64_t *stored, uint64_t *lowest);

#ifdef __cplusplus
}
#endif
#endif
