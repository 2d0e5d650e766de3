/*
 * bm_locator_oracle.c -- CPU restatement of bucket_locator's candidate scan.  TEST INFRASTRUCTURE ONLY
 * (see bm_locator_oracle.h; "PARITY UNPINNED").  Plain C, deliberately simple data structures: the
 * bucket index is an array sorted by (hash ascending, offset DESCENDING) -- the order in which
 * libstdc++'s unordered_multimap::equal_range yields equal keys for the reference's ascending inserts --
 * and the std::map<int,unsigned> vote counter is a sorted array.
 */
#include "bm_locator_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static uint8_t dna4_rank(uint8_t c) {
    switch (c) {
    case 'C': case 'c': case 'Y': case 'y': case 'S': case 's': case 'B': case 'b': return 1;
    case 'G': case 'g': case 'K': case 'k': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 0;
    }
}

/* utils.h:291-302 */
static uint32_t revcomp_hash(uint32_t h, uint32_t k) {
    uint32_t rc = 0;
    for (uint32_t i = 0; i < k; i++) {
        rc = (rc << 2) | ((~h) & 3u);
        h >>= 2;
    }
    return rc;
}

typedef struct {
    uint32_t hash;
    int32_t offset;
} entry_t;

struct bmlo_bucket_index {
    entry_t *e;
    uint32_t n;
};

static int cmp_entry(const void *a, const void *b) {
    const entry_t *x = (const entry_t *)a, *y = (const entry_t *)b;
    if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
    return x->offset > y->offset ? -1 : (x->offset < y->offset ? 1 : 0);   /* descending offset */
}

/* bucket_locator.h:162-177 */
bmlo_bucket_index *bmlo_index_bucket(const uint8_t *bases, uint32_t len, uint32_t k) {
    bmlo_bucket_index *ix = (bmlo_bucket_index *)calloc(1, sizeof *ix);
    if (!ix) return NULL;
    if (k == 0 || len < k) return ix;
    ix->n = len - k + 1;
    ix->e = (entry_t *)malloc((size_t)ix->n * sizeof(entry_t));
    for (uint32_t j = 0; j < ix->n; j++) {
        uint32_t h = 0;
        for (uint32_t t = 0; t < k; t++) h = (h << 2) | dna4_rank(bases[j + t]);
        ix->e[j].hash = h;
        ix->e[j].offset = (int32_t)j;
    }
    qsort(ix->e, ix->n, sizeof(entry_t), cmp_entry);
    return ix;
}

void bmlo_index_free(bmlo_bucket_index *ix) {
    if (!ix) return;
    free(ix->e);
    free(ix);
}

/* first entry with hash >= h */
static uint32_t lower_bound_hash(const bmlo_bucket_index *ix, uint32_t h) {
    uint32_t lo = 0, hi = ix->n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (ix->e[mid].hash < h) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* std::map<int, unsigned int> as a sorted array */
typedef struct {
    int32_t *key;
    uint32_t *val;
    uint32_t n, cap;
} votemap_t;

static uint32_t vm_lower(const votemap_t *m, int32_t k) {   /* first index with key >= k */
    uint32_t lo = 0, hi = m->n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (m->key[mid] < k) lo = mid + 1; else hi = mid;
    }
    return lo;
}
static uint32_t vm_upper(const votemap_t *m, int32_t k) {   /* first index with key > k */
    uint32_t lo = 0, hi = m->n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (m->key[mid] <= k) lo = mid + 1; else hi = mid;
    }
    return lo;
}
static void vm_increment(votemap_t *m, int32_t k) {          /* vote_counter[k]++ */
    uint32_t i = vm_lower(m, k);
    if (i < m->n && m->key[i] == k) {
        m->val[i]++;
        return;
    }
    if (m->n == m->cap) {
        m->cap = m->cap ? 2 * m->cap : 64;
        m->key = (int32_t *)realloc(m->key, m->cap * sizeof(int32_t));
        m->val = (uint32_t *)realloc(m->val, m->cap * sizeof(uint32_t));
    }
    memmove(m->key + i + 1, m->key + i, (m->n - i) * sizeof(int32_t));
    memmove(m->val + i + 1, m->val + i, (m->n - i) * sizeof(uint32_t));
    m->key[i] = k;
    m->val[i] = 1;
    m->n++;
}

/* bucket_locator.h:209-290 */
void bmlo_find_offset(const bmlo_params *p, const bmlo_bucket_index *ix, const uint32_t *kmers,
                      const uint16_t *indices, uint32_t length, int reverse_complement, int32_t *out_offset,
                      uint32_t *out_votes) {
    votemap_t vc;
    memset(&vc, 0, sizeof vc);
    const int32_t num_samples = (int32_t)p->num_samples;
    for (int32_t i = 0; i < num_samples; i++) {
        /* :235-243 first sample first for the read as-is, last sample first for its reverse complement */
        int32_t sample_index = reverse_complement ? num_samples - 1 - i : i;
        uint32_t current_kmer = kmers[sample_index], current_index = indices[sample_index];
        if (reverse_complement) {
            current_kmer = revcomp_hash(current_kmer, p->k);
            current_index = length - p->k - current_index;
        }
        uint32_t first = lower_bound_hash(ix, current_kmer), last = first;
        while (last < ix->n && ix->e[last].hash == current_kmer) last++;
        if (vc.n == 0) {
            /* :247-252 no proposal yet: every occurrence proposes */
            for (uint32_t o = first; o < last; o++) vm_increment(&vc, (int32_t)((uint32_t)ix->e[o].offset - current_index));
        } else {
            for (uint32_t o = first; o < last; o++) {
                /* :254-271 vote for EVERY existing proposal within +-allowed_indel, else propose */
                int voted = 0;
                const int32_t position = (int32_t)((uint32_t)ix->e[o].offset - current_index);
                uint32_t lb = vm_lower(&vc, position - p->allowed_indel);
                uint32_t ub = vm_upper(&vc, position + p->allowed_indel);
                for (uint32_t v = lb; v < ub; v++) {
                    vc.val[v]++;
                    voted = 1;
                }
                if (!voted) vm_increment(&vc, position);
            }
        }
    }
    *out_offset = -1;
    *out_votes = 0;
    if (vc.n) {
        /* :281-283 most votes, ties -> smallest offset */
        uint32_t best = 0;
        for (uint32_t v = 1; v < vc.n; v++)
            if (vc.val[v] > vc.val[best]) best = v;
        /* :284 unsigned >= int compares as unsigned */
        if (vc.val[best] >= (uint32_t)(num_samples - p->allowed_mismatch) && vc.key[best] >= 0) {
            *out_offset = vc.key[best];
            *out_votes = vc.val[best];
        }
    }
    free(vc.key);
    free(vc.val);
}

int bmlo_locate(const bmlo_params *p, const uint8_t *genome, const uint64_t *bucket_start, const uint32_t *bucket_len,
                uint32_t n_buckets, const uint32_t *sample_hash, const uint16_t *sample_pos, const uint32_t *seg_len,
                const uint32_t *pair_bucket, const uint32_t *pair_window, const uint8_t *pair_rc, uint32_t n_pairs,
                int32_t *out_offset, uint32_t *out_votes) {
    bmlo_bucket_index *ix = NULL;
    uint32_t cur = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n_pairs; i++) {
        const uint32_t b = pair_bucket[i], w = pair_window[i];
        if (b >= n_buckets) { bmlo_index_free(ix); return 1; }
        if (b != cur) {
            bmlo_index_free(ix);
            ix = bmlo_index_bucket(genome + bucket_start[b], bucket_len[b], p->k);
            cur = b;
        }
        bmlo_find_offset(p, ix, sample_hash + (size_t)w * p->num_samples, sample_pos + (size_t)w * p->num_samples,
                         seg_len[w], pair_rc[i], &out_offset[i], &out_votes[i]);
    }
    bmlo_index_free(ix);
    return 0;
}

/* bucket_locator.h:292-347 */
void bmlo_sample_windows(uint32_t k, uint32_t p, uint32_t min_base_quality, const uint8_t *bases, const uint8_t *quals,
                         const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows, uint32_t *out_hash,
                         uint16_t *out_pos, uint8_t *out_has) {
    for (uint32_t w = 0; w < n_windows; w++) {
        const uint8_t *b = bases + win_start[w], *q = quals + win_start[w];
        const uint32_t len = win_len[w];
        const uint32_t nk = len >= k ? len - k + 1 : 0;
        uint32_t *hash = out_hash + (size_t)w * p;
        uint16_t *pos = out_pos + (size_t)w * p;
        out_has[w] = nk ? 1 : 0;
        if (!nk) {
            for (uint32_t s = 0; s < p; s++) {
                hash[s] = 0;
                pos[s] = 0;
            }
            continue;
        }
        uint16_t *good = (uint16_t *)malloc((size_t)nk * sizeof(uint16_t));
        uint32_t n_good = 0;
        for (uint32_t j = 0; j < nk; j++) {                       /* :325-327 */
            uint32_t qs = 0;
            for (uint32_t t = 0; t < k; t++) qs += (uint32_t)q[j + t] - 33u;
            if (qs >= min_base_quality) good[n_good++] = (uint16_t)j;
        }
        if (n_good == 0)                                          /* :330-332 */
            for (uint32_t j = 0; j < nk; j++) good[n_good++] = (uint16_t)j;
        /* Sampler(p).sample_deterministically(n_good - 1), utils.h:160-178 */
        const uint32_t ub = n_good - 1;
        double delta = 0.0;
        if (p != 1) delta = (double)(ub + 1u) / (double)(p - 1u);
        for (uint32_t s = 0; s < p; s++) {
            const uint32_t at = s + 1 < p ? (uint32_t)floor((double)s * delta) : ub;
            const uint16_t j = good[at];
            uint32_t h = 0;
            for (uint32_t t = 0; t < k; t++) h = (h << 2) | dna4_rank(b[j + t]);
            hash[s] = h;
            pos[s] = j;
        }
        free(good);
    }
}
