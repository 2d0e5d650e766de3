/*
 * bm_oracle.h -- CPU restatement of BucketMap's candidate-bucket filter (TEST INFRASTRUCTURE).
 *
 * This is the parity ORACLE for the HIP path.  It is test infrastructure only: nothing
 * outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it,
 * and the product library (libbmf.so) never links or loads it.
 *
 * PARITY UNPINNED: the reference (GZHoffie/bucket-map) cannot be compiled in this image
 * (SeqAn3/Sharg are fetched by CMake FetchContent at tag `main`, bucket_map/CMakeLists.txt:69-80,
 * no copy on disk, no network) and it ships no golden vectors or tests for this path
 * (SURVEY.md section 4 / 8c).  The oracle is therefore pinned only by hand-derived known-answer
 * vectors, a dual-formulation property test (bit-plane filter vs integer miss counts, see
 * oracle/bm_oracle_np.py) and exhaustively enumerated tiny indexes -- not by reference outputs.
 *
 * Every function cites the reference file:line (relative to /root/reference/) it restates.
 * Written from the behavioural spec in SURVEY.md Appendix A; no reference source is copied.
 */
#ifndef BM_ORACLE_H
#define BM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Run-time form of the reference's constructor arguments
 * (bucket_map/mapper/q_gram_mapper.h:281-308, bucket_map/main.cpp:202-209). */
typedef struct bmo_params {
    uint32_t num_buckets;        /* NB: the reference's NUM_BUCKETS template argument            */
    uint32_t q;                  /* index seed length  (-k, default 9)                           */
    uint32_t k;                  /* query seed length  (-l, default 12)                          */
    uint32_t num_samples;        /* S  (-s, default 15)                                          */
    uint32_t num_fault;          /* F = ceil(S*e) computed in float32 (main.cpp:207)             */
    uint32_t threshold;          /* (unsigned)(d*NB) in float32 (q_gram_mapper.h:163)            */
    uint32_t min_base_quality;   /* b*k (q_gram_mapper.h:303)                                    */
    uint32_t max_candidates;     /* 30 (q_gram_mapper.h:285)                                     */
    uint32_t read_len;           /* -r (default 300)                                             */
    uint32_t num_segment_samples;/* 5 (q_gram_mapper.h:286)                                      */
} bmo_params;

typedef struct bmo_index bmo_index;

/* ---- parameter derivation, float32 exactly as the reference ---- */
uint32_t bmo_fault_from_rate(uint32_t samples, float max_error_rate);      /* main.cpp:207            */
uint32_t bmo_threshold(float distinguishability, uint32_t num_buckets);    /* q_gram_mapper.h:163     */
uint32_t bmo_ceil_mul_f32(float a, uint32_t b);                            /* bucket_locator.h:419-420*/

/* ---- small pure functions ---- */
/* utils.h:160-178.  Writes n positions.  Deviation (documented in DESIGN.md): the reference skips
 * re-sampling when upper_bound == 0 (its cache variable is never updated) and then indexes with
 * stale positions, which is an out-of-bounds read; here upper_bound == 0 yields n zeros. */
void     bmo_sample_positions(uint32_t n, uint32_t upper_bound, uint32_t *out);
uint32_t bmo_hash_reverse_complement(uint32_t hash, uint32_t k);           /* utils.h:291-302         */
uint8_t  bmo_dna4_rank(uint8_t c);                                         /* SeqAn3 dna4 (App. C.2)  */
/* SeqAn3 views::kmer_hash(ungapped{k}) on dna4 (App. C.1): out[j], j in [0, max(len+1,k)-k). */
uint32_t bmo_kmer_hashes(const uint8_t *bases_ascii, uint32_t len, uint32_t k, uint32_t *out);
/* quality_filter.h:531-534,611-631: sliding SUM of phred ranks (ASCII-33) over k bases. */
uint32_t bmo_kmer_qualities(const uint8_t *quals_ascii, uint32_t len, uint32_t k, uint32_t *out);
/* q_gram_mapper.h:510-516: window starts of one record. Returns count (1 or num_segment_samples). */
uint32_t bmo_window_starts(uint32_t record_len, uint32_t read_len, uint32_t n_seg, uint32_t *out);

/* ---- index: q_gram_mapper.h:318-372 (load), :171-187 (zeros) ---- */
/* rows: n_rows x ceil(NB/8) bytes, LSB-first bit j of row <-> bucket j (bucket_indexer.h:64-73). */
bmo_index *bmo_index_create(const bmo_params *p, const uint8_t *rows, uint64_t n_rows,
                            const int32_t *kmer_to_index, uint64_t n_kmers);
/* Reads <dir>/<indicator>.kmers_index and .qgram like mapper::load. NULL on failure. */
bmo_index *bmo_index_load(const bmo_params *p, const char *dir, const char *indicator);
void       bmo_index_destroy(bmo_index *ix);
uint64_t   bmo_index_rows(const bmo_index *ix);
const uint32_t *bmo_index_zeros(const bmo_index *ix);
/* q_gram_mapper.h:189-196 */
int        bmo_is_highly_distinguishable(const bmo_index *ix, uint32_t kmer_hash);

/* ---- the vote: q_gram_mapper.h:380-412 + fault_tolerate_filter :59-102 ----
 * out must hold NB entries; returns the number of bucket ids written (ascending). */
uint32_t bmo_query(const bmo_index *ix, const uint32_t *kmer_hashes, uint32_t n, uint32_t *out);

/* q_gram_mapper.h:414-480.  out_fwd / out_rc hold max_candidates entries each.
 * Optionally returns the sampled hashes (S entries) and #good k-mers for white-box tests. */
void bmo_query_sequence(const bmo_index *ix, const uint8_t *bases_ascii, const uint8_t *quals_ascii,
                        uint32_t len, uint32_t *out_fwd, uint32_t *n_fwd, uint32_t *out_rc,
                        uint32_t *n_rc, uint32_t *dbg_samples, uint32_t *dbg_n_good);

/* Batch form with the same buffer layout as bmf_map_windows (include/bmf.h): window w is the view
 * [win_start[w], win_start[w]+win_len[w]) of bases/quals.  out_counts[2w]=fwd count, [2w+1]=rc count;
 * out_buckets[(2w+o)*max_candidates + i].  Also returns the number of index rows ANDed
 * (the reference's row reads, both orientations) for the algorithmic-bytes figure. */
uint64_t bmo_map_windows(const bmo_index *ix, const uint8_t *bases, const uint8_t *quals,
                         const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows,
                         uint32_t *out_counts, uint32_t *out_buckets);

#ifdef __cplusplus
}
#endif
#endif
