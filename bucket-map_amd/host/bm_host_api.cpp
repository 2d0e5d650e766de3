// bm_host_api.cpp -- C ABI over the host plumbing (libbmhost.so): synthetic genomes, FASTA I/O, bucket
// cutting, the host indexer and the read simulator.  Used by tests/ and bench.py through ctypes; the
// C++ command-line tool links the same headers directly.  No GPU code here.
#include "bm_genome.h"
#include "bm_indexer.h"
#include "bm_synth.h"

#include <cstdarg>
#include <cstdio>

namespace {
thread_local char g_err[512] = "";
void set_err(const char *what) { snprintf(g_err, sizeof g_err, "%s", what); }
}  // namespace

struct bmh_genome { bm::Genome g; };
struct bmh_index { bm::QgramIndex ix; };
struct bmh_reads { bm::SimReads rd; };

#define BMH_GUARD(ret_on_error, ...)                  \
    try { __VA_ARGS__ } catch (const std::exception &e) { \
        set_err(e.what());                            \
        return ret_on_error;                          \
    }

extern "C" {

const char *bmh_last_error(void) { return g_err; }

bmh_genome *bmh_genome_synth(uint64_t seed, const uint64_t *record_lengths, uint32_t n_records, uint32_t threads) {
    BMH_GUARD(nullptr, {
        std::vector<uint64_t> lens(record_lengths, record_lengths + n_records);
        auto *h = new bmh_genome();
        h->g = bm::synth_genome(seed, lens, threads);
        return h;
    })
}

// profile 1: the genome-like generator (bm_synth.h, synth_genome_skewed); sigma <= 0 keeps the profile's default
bmh_genome *bmh_genome_synth_skewed(uint64_t seed, const uint64_t *record_lengths, uint32_t n_records, uint32_t threads,
                                    double sigma, double p_repeat) {
    BMH_GUARD(nullptr, {
        std::vector<uint64_t> lens(record_lengths, record_lengths + n_records);
        bm::SkewProfile pf;
        if (sigma > 0) pf.sigma = sigma;
        if (p_repeat >= 0) pf.p_repeat = p_repeat;
        auto *h = new bmh_genome();
        h->g = bm::synth_genome_skewed(seed, lens, threads, pf);
        return h;
    })
}
uint64_t bmh_genome_gap_bases(const bmh_genome *g) {
    uint64_t n = 0;
    for (auto &gp : g->g.gaps) n += gp[2];
    return n;
}

bmh_genome *bmh_genome_read_fasta(const char *path) {
    BMH_GUARD(nullptr, {
        auto *h = new bmh_genome();
        h->g = bm::read_fasta(path);
        return h;
    })
}

int bmh_genome_write_fasta(const bmh_genome *g, const char *path) {
    BMH_GUARD(1, { bm::write_fasta(g->g, path); return 0; })
}

void bmh_genome_free(bmh_genome *g) { delete g; }
uint32_t bmh_genome_records(const bmh_genome *g) { return static_cast<uint32_t>(g->g.seqs.size()); }
uint64_t bmh_genome_record_len(const bmh_genome *g, uint32_t i) { return g->g.seqs[i].size(); }
const char *bmh_genome_record_id(const bmh_genome *g, uint32_t i) { return g->g.ids[i].c_str(); }
const char *bmh_genome_record_seq(const bmh_genome *g, uint32_t i) { return g->g.seqs[i].data(); }

// All records back to back in one caller buffer of bmh_genome_total(g) bytes; rec_off receives
// records+1 offsets.  (What the GPU index build and the GPU locator scan take as "the genome".)
uint64_t bmh_genome_total(const bmh_genome *g) { return g->g.total_length(); }
void bmh_genome_flatten(const bmh_genome *g, uint8_t *out, uint64_t *rec_off) {
    uint64_t o = 0;
    for (size_t r = 0; r < g->g.seqs.size(); r++) {
        rec_off[r] = o;
        std::memcpy(out + o, g->g.seqs[r].data(), g->g.seqs[r].size());
        o += g->g.seqs[r].size();
    }
    rec_off[g->g.seqs.size()] = o;
}

// FracMinHash row selection alone (bucket_indexer.h:147-157): fills 4^q entries, returns the row count.
uint64_t bmh_select_qgrams(uint32_t q, float kmer_frac, uint64_t hash_seed, int32_t *out_k2i) {
    bm::QgramIndex ix;
    bm::select_qgrams(ix, q, bm::FracMinHash::from_seed(hash_seed), kmer_frac);
    std::memcpy(out_k2i, ix.kmer_to_index.data(), ix.kmer_to_index.size() * sizeof(int32_t));
    return ix.num_rows;
}

// Same with the hash function given outright -- (x*i + y) % p % table, the closure hash_function_generator::generate
// returns (hash_function_generator.h:105-116) -- so that a selection made by the reference's own header can be reproduced.
uint64_t bmh_select_qgrams_xy(uint32_t q, float kmer_frac, uint64_t x, uint64_t y, uint64_t p, uint64_t table,
                              int32_t *out_k2i) {
    bm::QgramIndex ix;
    bm::FracMinHash h;
    h.x = x; h.y = y; h.p = p; h.table = table;
    bm::select_qgrams(ix, q, h, kmer_frac);
    std::memcpy(out_k2i, ix.kmer_to_index.data(), ix.kmer_to_index.size() * sizeof(int32_t));
    return ix.num_rows;
}

// Walks a FASTQ file with the tool's own reader: number of records, total bases and a checksum over
// ids, sequences and qualities (for tests of the block reader).
int bmh_fastq_stats(const char *path, uint64_t *n_records, uint64_t *n_bases, uint64_t *checksum) {
    BMH_GUARD(1, {
        uint64_t n = 0, b = 0, h = 1469598103934665603ull;
        auto mix = [&](std::string_view v) {
            for (unsigned char c : v) h = (h ^ c) * 1099511628211ull;
            h = (h ^ 0xFFu) * 1099511628211ull;
        };
        bm::for_each_fastq(path, [&](const bm::FastqRecord &r) {
            n++;
            b += r.seq.size();
            mix(r.id); mix(r.seq); mix(r.qual);
        });
        *n_records = n; *n_bases = b; *checksum = h;
        return 0;
    })
}

// BM_BUCKET_NUM as bucket_map/CMakeLists.txt:13-46 computes it
uint32_t bmh_awk_bucket_num(const bmh_genome *g, uint32_t bucket_len) { return bm::awk_bucket_num(g->g, bucket_len); }

// utils.h:72-97.  out (may be NULL) receives 4 u32 per kept bucket: record, index, start, end.
uint32_t bmh_cut_buckets(const bmh_genome *g, uint32_t bucket_len, uint32_t read_len, uint32_t *out) {
    auto b = bm::cut_buckets(g->g, static_cast<int>(bucket_len), static_cast<int>(read_len));
    if (out)
        for (size_t i = 0; i < b.size(); i++) {
            out[4 * i] = b[i].record; out[4 * i + 1] = b[i].index; out[4 * i + 2] = b[i].start; out[4 * i + 3] = b[i].end;
        }
    return static_cast<uint32_t>(b.size());
}

bmh_index *bmh_index_build(const bmh_genome *g, uint32_t num_buckets, uint32_t bucket_len, uint32_t read_len,
                           uint32_t q, float kmer_frac, uint64_t hash_seed, uint32_t threads) {
    BMH_GUARD(nullptr, {
        auto *h = new bmh_index();
        h->ix = bm::build_index(g->g, num_buckets, static_cast<int>(bucket_len), static_cast<int>(read_len), q,
                                bm::FracMinHash::from_seed(hash_seed), kmer_frac, threads);
        return h;
    })
}
void bmh_index_free(bmh_index *ix) { delete ix; }
uint64_t bmh_index_num_rows(const bmh_index *ix) { return ix->ix.num_rows; }
uint32_t bmh_index_row_bytes(const bmh_index *ix) { return ix->ix.row_bytes; }
const uint8_t *bmh_index_rows(const bmh_index *ix) { return ix->ix.rows.data(); }
const int32_t *bmh_index_kmer_to_index(const bmh_index *ix) { return ix->ix.kmer_to_index.data(); }
uint64_t bmh_index_num_kmers(const bmh_index *ix) { return ix->ix.kmer_to_index.size(); }
int bmh_index_write(const bmh_index *ix, const char *dir, const char *indicator) {
    BMH_GUARD(1, { bm::write_index(ix->ix, dir, indicator); return 0; })
}

// qmode: 0 = all 'E' (as the reference's simulator), 1 = noisy qualities
bmh_reads *bmh_reads_simulate(const bmh_genome *g, uint32_t bucket_len, uint32_t index_read_len, uint32_t read_len,
                              uint64_t n_reads, double sub_rate, double ins_rate, double del_rate, uint64_t seed,
                              uint32_t qmode, uint32_t threads) {
    BMH_GUARD(nullptr, {
        auto buckets = bm::cut_buckets(g->g, static_cast<int>(bucket_len), static_cast<int>(index_read_len));
        auto *h = new bmh_reads();
        h->rd = bm::simulate_reads(g->g, buckets, read_len, n_reads, sub_rate, ins_rate, del_rate, seed,
                                   qmode ? bm::QUAL_NOISY : bm::QUAL_ALL_E, threads);
        return h;
    })
}
void bmh_reads_free(bmh_reads *r) { delete r; }
uint64_t bmh_reads_count(const bmh_reads *r) { return r->rd.size(); }
const uint8_t *bmh_reads_bases(const bmh_reads *r) { return r->rd.bases.data(); }
const uint8_t *bmh_reads_quals(const bmh_reads *r) { return r->rd.quals.data(); }
const uint64_t *bmh_reads_offsets(const bmh_reads *r) { return r->rd.offsets.data(); }
const uint32_t *bmh_reads_truth_bucket(const bmh_reads *r) { return r->rd.truth_bucket.data(); }
const uint32_t *bmh_reads_truth_offset(const bmh_reads *r) { return r->rd.truth_offset.data(); }
const uint8_t *bmh_reads_truth_rc(const bmh_reads *r) { return r->rd.truth_rc.data(); }
int bmh_reads_write_fastq(const bmh_reads *r, const bmh_genome *g, uint32_t bucket_len, uint32_t index_read_len,
                          const char *prefix) {
    BMH_GUARD(1, {
        auto buckets = bm::cut_buckets(g->g, static_cast<int>(bucket_len), static_cast<int>(index_read_len));
        bm::write_fastq(r->rd, g->g, buckets, static_cast<int>(bucket_len), prefix);
        return 0;
    })
}

}  // extern "C"
