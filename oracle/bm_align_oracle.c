/*
 * bm_align_oracle.c -- see bm_align_oracle.h.  TEST INFRASTRUCTURE: never linked into the product.
 *
 * Deliberately the textbook formulation (one full (m+1) x (n+1) matrix of edit distances, then a
 * traceback that re-derives which predecessors are valid), so that it shares nothing with the
 * bit-vector kernel it checks.
 */
#include "bm_align_oracle.h"

#include <stdlib.h>
#include <string.h>

/* seqan3::dna4 char -> rank folding, same table as bm_oracle.c */
static uint8_t rank_of(uint8_t c) {
    switch (c) {
    case 'C': case 'c': case 'Y': case 'y': case 'S': case 's': case 'B': case 'b': return 1;
    case 'G': case 'g': case 'K': case 'k': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 0;
    }
}

int bmao_align(const uint8_t *text, uint32_t n, int text_rc, const uint8_t *query, uint32_t m, int32_t *out_score,
               uint32_t *out_begin, uint32_t *out_cigar, uint32_t cigar_cap) {
    uint8_t *t = (uint8_t *)malloc(n ? n : 1), *q = (uint8_t *)malloc(m ? m : 1);
    uint32_t *H = (uint32_t *)malloc(((size_t)m + 1) * ((size_t)n + 1) * sizeof(uint32_t));
    uint32_t *rev = (uint32_t *)malloc(((size_t)m + n + 1) * sizeof(uint32_t));
    if (!t || !q || !H || !rev) {
        free(t); free(q); free(H); free(rev);
        return -1;
    }
    /* bucket_locator.h:563-567: text | reverse | complement (complement of rank r is 3 - r) */
    for (uint32_t j = 0; j < n; j++) t[j] = text_rc ? (uint8_t)(3 - rank_of(text[n - 1 - j])) : rank_of(text[j]);
    for (uint32_t i = 0; i < m; i++) q[i] = rank_of(query[i]);
    const size_t W = (size_t)n + 1;
    /* free_end_gaps_sequence1_leading: row 0 costs nothing; sequence2 (the query) pays for its gaps */
    for (uint32_t j = 0; j <= n; j++) H[j] = 0;
    for (uint32_t i = 1; i <= m; i++) {
        H[i * W] = i;
        for (uint32_t j = 1; j <= n; j++) {
            uint32_t best = H[(i - 1) * W + (j - 1)] + (q[i - 1] != t[j - 1] ? 1u : 0u);
            const uint32_t up = H[(i - 1) * W + j] + 1u, left = H[i * W + (j - 1)] + 1u;
            if (up < best) best = up;
            if (left < best) best = left;
            H[i * W + j] = best;
        }
    }
    /* free_end_gaps_sequence1_trailing: best cell of the bottom row; ASSUMPTION (1): the last one */
    uint32_t jend = 0, best = H[m * W];
    for (uint32_t j = 1; j <= n; j++)
        if (H[m * W + j] <= best) {
            best = H[m * W + j];
            jend = j;
        }
    /* traceback, ASSUMPTION (2): diagonal, then up, then left */
    uint32_t i = m, j = jend, n_rev = 0;
    while (i > 0) {
        uint32_t op;
        if (j > 0 && H[i * W + j] == H[(i - 1) * W + (j - 1)] + (q[i - 1] != t[j - 1] ? 1u : 0u)) {
            op = BMAO_OP_M;
            i--; j--;
        } else if (H[i * W + j] == H[(i - 1) * W + j] + 1u) {
            op = BMAO_OP_I;
            i--;
        } else {
            op = BMAO_OP_D;
            j--;
        }
        if (n_rev && (rev[n_rev - 1] & 15u) == op) rev[n_rev - 1] += 16u;
        else rev[n_rev++] = (1u << 4) | op;
    }
    *out_score = -(int32_t)best;
    *out_begin = j;
    for (uint32_t x = 0; x < n_rev && x < cigar_cap; x++) out_cigar[x] = rev[n_rev - 1 - x];
    free(t); free(q); free(H); free(rev);
    return (int)n_rev;
}

int bmao_align_batch(const uint8_t *genome, const uint8_t *reads, const uint64_t *text_start, const uint32_t *text_len,
                     const uint8_t *text_rc, const uint64_t *query_start, const uint32_t *query_len, uint32_t n,
                     int32_t *out_score, uint32_t *out_begin, uint64_t *out_cigar_offset, uint32_t *out_cigar,
                     uint64_t cigar_cap) {
    uint64_t at = 0;
    int short_buf = 0;
    for (uint32_t a = 0; a < n; a++) {
        out_cigar_offset[a] = at;
        const uint64_t room = at < cigar_cap ? cigar_cap - at : 0;
        const int c = bmao_align(genome + text_start[a], text_len[a], text_rc[a], reads + query_start[a], query_len[a],
                                 &out_score[a], &out_begin[a], out_cigar ? out_cigar + (at < cigar_cap ? at : 0) : 0,
                                 (uint32_t)(room > 0xFFFFFFFFull ? 0xFFFFFFFFull : room));
        if (c < 0) return -1;
        if ((uint64_t)c > room) short_buf = 1;
        at += (uint64_t)c;
    }
    out_cigar_offset[n] = at;
    return short_buf;
}

/* one row of the two-row programme; compiled a second time for AVX2 where the CPU has it (the first loop is 8 cells an instruction) */
__attribute__((target_clones("avx2", "default")))
static void two_row_step(const uint32_t *prev, uint32_t *cur, const uint8_t *t, uint8_t qi, uint32_t n) {
    for (uint32_t j = 1; j <= n; j++) {                 /* diagonal and up first: no dependence along the row */
        const uint32_t d = prev[j - 1] + (qi != t[j - 1] ? 1u : 0u), u = prev[j] + 1u;
        cur[j] = d < u ? d : u;
    }
    uint32_t run = cur[0];                              /* then the gap in the query, left to right (carried in a register) */
    for (uint32_t j = 1; j <= n; j++) {
        const uint32_t v = cur[j], l = run + 1u;
        run = l < v ? l : v;
        cur[j] = run;
    }
}

int bmao_check(const uint8_t *text, uint32_t n, int text_rc, const uint8_t *query, uint32_t m, int32_t score, uint32_t begin,
               const uint32_t *cigar, uint64_t n_cigar, int32_t *out_optimum) {
    uint8_t *t = (uint8_t *)malloc(n ? n : 1), *q = (uint8_t *)malloc(m ? m : 1);
    uint32_t *prev = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t)), *cur = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    if (!t || !q || !prev || !cur) {
        free(t); free(q); free(prev); free(cur);
        return 8;
    }
    for (uint32_t j = 0; j < n; j++) t[j] = text_rc ? (uint8_t)(3 - rank_of(text[n - 1 - j])) : rank_of(text[j]);
    for (uint32_t i = 0; i < m; i++) q[i] = rank_of(query[i]);
    /* (a) two rows of the same recurrence: row 0 is free, column 0 costs i */
    for (uint32_t j = 0; j <= n; j++) prev[j] = 0;
    for (uint32_t i = 1; i <= m; i++) {
        cur[0] = i;
        two_row_step(prev, cur, t, q[i - 1], n);
        uint32_t *x = prev; prev = cur; cur = x;
    }
    uint32_t best = prev[0];
    for (uint32_t j = 1; j <= n; j++)
        if (prev[j] < best) best = prev[j];
    *out_optimum = -(int32_t)best;
    int bad = score == -(int32_t)best ? 0 : 1;
    /* (b) the CIGAR as a path */
    uint64_t i = 0, j = begin, cost = 0;
    int shape_ok = begin <= n;
    for (uint64_t x = 0; x < n_cigar && shape_ok; x++) {
        const uint32_t len = cigar[x] >> 4, op = cigar[x] & 15u;
        if (len == 0 || op > BMAO_OP_D || (x && (cigar[x - 1] & 15u) == op)) shape_ok = 0;      /* run-length form */
        else if (op == BMAO_OP_M) {
            if (i + len > m || j + len > n) shape_ok = 0;
            else for (uint32_t y = 0; y < len; y++) cost += q[i + y] != t[j + y];
            i += len; j += len;
        } else if (op == BMAO_OP_I) {
            if (i + len > m) shape_ok = 0;
            i += len; cost += len;
        } else {
            if (j + len > n) shape_ok = 0;
            j += len; cost += len;
        }
    }
    if (!shape_ok || i != m) bad |= 2;
    else if ((int64_t)cost != -(int64_t)score) bad |= 4;
    free(t); free(q); free(prev); free(cur);
    return bad;
}

void bmao_check_batch(const uint8_t *genome, const uint8_t *reads, const uint64_t *text_start, const uint32_t *text_len,
                      const uint8_t *text_rc, const uint64_t *query_start, const uint32_t *query_len, uint32_t first,
                      uint32_t last, const int32_t *score, const uint32_t *begin, const uint64_t *cigar_offset,
                      const uint32_t *cigar, uint8_t *out_bad) {
    for (uint32_t a = first; a < last; a++) {
        int32_t opt = 0;
        out_bad[a] = (uint8_t)bmao_check(genome + text_start[a], text_len[a], text_rc[a], reads + query_start[a], query_len[a], score[a],
                                         begin[a], cigar + cigar_offset[a], cigar_offset[a + 1] - cigar_offset[a], &opt);
    }
}
