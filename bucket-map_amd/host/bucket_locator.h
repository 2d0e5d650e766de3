// bucket_locator.h -- exact position inside the candidate buckets + SAM output (host, CPU).
//
// Restates bucket_locator (bucket_map/locator/bucket_locator.h) without SeqAn3:
//   query_sequences_storage / _prepare_read_query   :19-103, :292-347
//   _create_kmer_index                               :162-177
//   _find_offset                                     :209-290
//   _filter_best_locations                           :350-405
//   locate (SAM)                                     :455-611  (non-BM_ALIGN branch)
//   _locate (bucket loop and its ordering contract)  :613-705
// This is the caller on the far side of the mapper boundary; it stays on the
// host in this round and talks to the filter only through bm::mapper, exactly as the reference does.
// Order-sensitive details are kept on purpose: occurrences of one k-mer are visited
// in the order libstdc++'s unordered_multimap::equal_range yields them, revcomp candidates of a bucket
// are scanned in reverse list order, `offset > 0` drops an exact hit at bucket offset 0.
#pragma once

#include "bm_genome.h"
#include "mapper.h"

#include <algorithm>
#include <chrono>
#include <iostream>
#include <map>
#include <tuple>
#include <unordered_map>

namespace bm {

class bucket_locator {
public:
    // (bucket id, offset in the bucket, window offset in the read, votes, true = read as-is)
    using locate_t = std::tuple<unsigned int, int, unsigned int, unsigned int, bool>;

private:
    mapper *_m;
    const Genome *genome_ = nullptr;
    std::vector<Bucket> buckets_;

    unsigned int bucket_length, read_length, min_base_quality;
    uint8_t k;
    int allowed_mismatch, allowed_indel;
    float allowed_indel_rate;
    int num_samples;
    unsigned int num_segment_samples;

    // query_sequences_storage (:19-103): sampled k-mers of every window, keyed by (read, window start)
    struct Record {
        std::vector<unsigned int> kmers;
        std::vector<uint16_t> indices;
        unsigned int segment_length = 0;
    };
    std::map<segment_info_t, unsigned int> segment_to_index;
    std::vector<Record> records;
    std::vector<unsigned int> read_lengths;

    static uint32_t kmer_hash_at(const char *s, uint32_t k_) {
        uint32_t h = 0;
        for (uint32_t t = 0; t < k_; t++) h = (h << 2) | dna4_rank(static_cast<uint8_t>(s[t]));
        return h;
    }

    // _prepare_read_query (:292-347)
    void prepare_read_query(const std::string &fastq) {
        unsigned int read_index = 0;
        for_each_fastq(fastq, [&](const FastqRecord &rec) {
            const uint32_t len = static_cast<uint32_t>(rec.seq.size());
            std::vector<uint32_t> starting_positions{0};
            if (len > 2 * read_length) starting_positions = sample_deterministically(num_segment_samples, len - read_length - 1);
            for (uint32_t i : starting_positions) {
                const uint32_t begin = i, end = std::min(i + read_length, len);
                const uint32_t seg_len = end - begin;
                const int num_kmers = seg_len >= k ? static_cast<int>(seg_len - k + 1) : 0;
                // quality filter only (:325-327): sum of phred ranks over the k bases >= b*k
                std::vector<uint16_t> good_indices;
                for (int j = 0; j < num_kmers; j++) {
                    unsigned int qs = 0;
                    for (uint32_t t = 0; t < k; t++) qs += static_cast<uint8_t>(rec.qual[begin + j + t]) - 33u;
                    if (qs >= min_base_quality) good_indices.push_back(static_cast<uint16_t>(j));
                }
                if (good_indices.empty())
                    for (int j = 0; j < num_kmers; j++) good_indices.push_back(static_cast<uint16_t>(j));
                Record r;
                r.segment_length = seg_len;
                if (!good_indices.empty()) {
                    // Sampler(p) over the good k-mers (:333-335)
                    for (uint32_t p : sample_deterministically(static_cast<uint32_t>(num_samples),
                                                               static_cast<uint32_t>(good_indices.size() - 1))) {
                        const uint16_t j = good_indices[p];
                        r.indices.push_back(j);
                        r.kmers.push_back(kmer_hash_at(rec.seq.data() + begin + j, k));
                    }
                }
                segment_to_index[segment_info_t{read_index, static_cast<int>(i)}] = static_cast<unsigned int>(records.size());
                records.push_back(std::move(r));
            }
            read_lengths.push_back(len);
            read_index++;
        });
    }

    // _create_kmer_index (:162-177): every k-mer of the bucket, inserted in ascending offset
    void create_kmer_index(std::unordered_multimap<unsigned int, int> &index, const Bucket &b) const {
        index.clear();
        index.reserve(bucket_length);
        const char *s = genome_->seqs[b.record].data() + b.start;
        const uint32_t len = b.end - b.start;
        if (len < k) return;
        const uint32_t mask = k >= 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
        uint32_t h = 0;
        for (uint32_t i = 0; i < len; i++) {
            h = ((h << 2) | dna4_rank(static_cast<uint8_t>(s[i]))) & mask;
            if (i + 1 >= k) index.emplace(h, static_cast<int>(i + 1 - k));
        }
    }

    // _find_offset (:209-290)
    std::pair<int, unsigned int> find_offset(const std::unordered_multimap<unsigned int, int> &bucket_kmer_index,
                                             const segment_info_t &segment, bool reverse_complement) {
        const Record &rec = records[segment_to_index[segment]];
        const unsigned int length = rec.segment_length;
        std::map<int, unsigned int> vote_counter;
        if (static_cast<int>(rec.kmers.size()) < num_samples) return std::make_pair(-1, 0u);
        for (int i = 0; i < num_samples; i++) {
            int sample_index = reverse_complement ? num_samples - 1 - i : i;
            unsigned int current_kmer = rec.kmers[sample_index], current_index = rec.indices[sample_index];
            if (reverse_complement) {
                current_kmer = hash_reverse_complement(current_kmer, k);
                current_index = length - k - current_index;
            }
            auto range = bucket_kmer_index.equal_range(current_kmer);
            if (vote_counter.empty()) {
                for (auto it = range.first; it != range.second; ++it) vote_counter[it->second - static_cast<int>(current_index)]++;
            } else {
                for (auto it = range.first; it != range.second; ++it) {
                    bool voted = false;
                    const int position = it->second - static_cast<int>(current_index);
                    auto lower = vote_counter.lower_bound(position - allowed_indel);
                    auto upper = vote_counter.upper_bound(position + allowed_indel);
                    for (auto v = lower; v != upper; ++v) {
                        v->second++;
                        voted = true;
                    }
                    if (!voted) vote_counter[position]++;
                }
            }
        }
        if (!vote_counter.empty()) {
            // most votes, ties -> smallest offset (:281-283)
            auto best = vote_counter.begin();
            for (auto it = vote_counter.begin(); it != vote_counter.end(); ++it)
                if (it->second > best->second) best = it;
            // unsigned >= int compares as unsigned in the reference (:284)
            if (best->second >= static_cast<unsigned int>(num_samples - allowed_mismatch) && best->first >= 0)
                return std::make_pair(best->first, best->second);
        }
        return std::make_pair(-1, 0u);
    }

    // _filter_best_locations (:350-405)
    std::vector<locate_t> filter_best_locations(const std::vector<locate_t> &mapped_locations, unsigned int read_len) const {
        std::map<std::tuple<unsigned int, int, bool>, unsigned int> loc_votes;
        for (auto &[bucket_id, bucket_offset, segment_offset, votes, is_orig] : mapped_locations) {
            (void)segment_offset;
            if (loc_votes.empty()) {
                loc_votes[{bucket_id, bucket_offset, is_orig}] = votes;
            } else {
                bool found_close_loc = false;
                // int = int -/+ float product, truncated (:365-366)
                const int lower_bound = static_cast<int>(bucket_offset - read_len * allowed_indel_rate);
                const int upper_bound = static_cast<int>(bucket_offset + read_len * allowed_indel_rate);
                for (auto it = loc_votes.begin(); it != loc_votes.end(); ++it) {
                    const int proposed = std::get<1>(it->first);
                    if (bucket_id == std::get<0>(it->first) && proposed <= upper_bound && proposed >= lower_bound &&
                        std::get<2>(it->first) == is_orig) {
                        it->second += votes;
                        found_close_loc = true;
                    }
                }
                if (!found_close_loc) loc_votes[{bucket_id, bucket_offset, is_orig}] = votes;
            }
        }
        std::vector<locate_t> res;
        unsigned int max_votes = 0;
        for (auto &kv : loc_votes) {
            if (kv.second > max_votes) {
                res.clear();
                max_votes = kv.second;
            }
            if (kv.second == max_votes)
                res.push_back(std::make_tuple(std::get<0>(kv.first), std::get<1>(kv.first), 0u, kv.second, std::get<2>(kv.first)));
        }
        return res;
    }

public:
    // bucket_locator ctor (:409-432); the indexer pointer of the reference is replaced by the genome
    // (locator::initialize -> indexer::index is done by the caller, see main.cpp).
    bucket_locator(mapper *map, unsigned int bucket_len, unsigned int read_len, uint8_t seed_len, float mismatch_rate,
                   float indel_rate, unsigned int sample_size, unsigned int quality_threshold,
                   unsigned int num_segment_samples_ = 5)
        : _m(map), bucket_length(bucket_len), read_length(read_len), k(seed_len) {
        allowed_mismatch = static_cast<int>(ceil_mul_f32(mismatch_rate, sample_size));   // :419
        allowed_indel = static_cast<int>(ceil_mul_f32(indel_rate, read_len));            // :420
        allowed_indel_rate = indel_rate;
        num_samples = static_cast<int>(sample_size);
        num_segment_samples = num_segment_samples_;
        min_base_quality = quality_threshold * k;                                         // :431
    }

    // initialize (:440-453): remember the genome, load the q-gram index into the mapper
    void initialize(const Genome &genome, std::filesystem::path const &index_directory, std::string const &indicator) {
        genome_ = &genome;
        _m->load(index_directory, indicator);
    }

    // _locate (:613-705)
    std::vector<std::vector<locate_t>> locate_reads(const std::string &sequence_file) {
        auto [sequence_ids_orig, sequence_ids_rev_comp] = _m->map(sequence_file);
        _m->reset();
        buckets_ = cut_buckets(*genome_, static_cast<int>(bucket_length), static_cast<int>(read_length));
        records.clear();
        segment_to_index.clear();
        read_lengths.clear();
        prepare_read_query(sequence_file);
        std::vector<std::vector<locate_t>> res(_m->num_records);
        std::unordered_multimap<unsigned int, int> bucket_kmer_index;
        float index_s = 0, query_s = 0;
        for (size_t i = 0; i < sequence_ids_orig.size(); i++) {
            auto &orig = sequence_ids_orig[i];
            auto &rev = sequence_ids_rev_comp[i];
            if (orig.empty() && rev.empty()) continue;
            if (i >= buckets_.size()) continue;   // padding bucket ids (NB > kept buckets) hold no sequence
            auto t0 = std::chrono::steady_clock::now();
            create_kmer_index(bucket_kmer_index, buckets_[i]);
            auto t1 = std::chrono::steady_clock::now();
            for (auto &id : orig) {
                auto [offset, vote] = find_offset(bucket_kmer_index, id, false);
                if (offset > 0)
                    res[id.first].push_back(std::make_tuple(static_cast<unsigned int>(i), offset - id.second,
                                                            static_cast<unsigned int>(id.second), vote, true));
            }
            for (auto it = rev.rbegin(); it != rev.rend(); ++it) {
                auto &id = *it;
                auto [offset, vote] = find_offset(bucket_kmer_index, id, true);
                if (offset > 0) {
                    // get_read_length / get_segment_length return uint16_t in the reference (:53-60)
                    const int segment_offset_ = static_cast<uint16_t>(read_lengths[id.first]) - id.second -
                                                static_cast<uint16_t>(records[segment_to_index[id]].segment_length);
                    res[id.first].push_back(std::make_tuple(static_cast<unsigned int>(i), offset - segment_offset_,
                                                            static_cast<unsigned int>(id.second), vote, false));
                }
            }
            auto t2 = std::chrono::steady_clock::now();
            index_s += std::chrono::duration<float>(t1 - t0).count();
            query_s += std::chrono::duration<float>(t2 - t1).count();
        }
        std::cerr << "[BENCHMARK]\tTotal time used for building k-mer index for each bucket: " << index_s << " s.\n";
        std::cerr << "[BENCHMARK]\tTotal time used for finding exact location of the sequences: " << query_s << " s ("
                  << query_s * 1000 * 1000 / _m->num_records << " μs/seq).\n";
        return res;
    }

    // locate (:455-611), non-BM_ALIGN branch: one SAM record per surviving location
    void locate(const std::string &sequence_file, std::filesystem::path const &index_file,
                std::filesystem::path const &sam_file, unsigned int quality_threshold = 30) {
        (void)quality_threshold;   // only used under BM_ALIGN in the reference
        auto locate_res = locate_reads(sequence_file);

        // .bucket_id -> @SQ lines and per-bucket offsets (:473-503)
        std::ifstream bucket_info(index_file);
        std::vector<std::string> bucket_name, ref_ids;
        std::vector<unsigned int> bucket_offsets;
        std::vector<size_t> ref_lengths;
        std::string name, last_bucket_name;
        unsigned int bucket_index = 0;
        for (size_t i = 0; i < buckets_.size(); i++) {
            std::getline(bucket_info, name);
            name = name.substr(0, name.find(' '));
            if (name != last_bucket_name) {
                if (bucket_index != 0) {
                    ref_ids.push_back(last_bucket_name);
                    ref_lengths.push_back(static_cast<size_t>(bucket_index) * bucket_length);
                }
                last_bucket_name = name;
                bucket_index = 0;
            }
            bucket_name.push_back(name);
            bucket_offsets.push_back(bucket_index * bucket_length);
            bucket_index++;
        }
        if (bucket_index != 0) {
            ref_ids.push_back(last_bucket_name);
            ref_lengths.push_back(static_cast<size_t>(bucket_index) * bucket_length);
        }

        // SAM as seqan3::sam_file_output writes it
        std::ofstream sam(sam_file, std::ios::binary);
        if (!sam) throw std::runtime_error("cannot write " + sam_file.string());
        sam << "@HD\tVN:1.6\n";
        for (size_t i = 0; i < ref_ids.size(); i++) sam << "@SQ\tSN:" << ref_ids[i] << "\tLN:" << ref_lengths[i] << "\n";
        unsigned int read_id = 0, mapped_locations = 0;
        auto t0 = std::chrono::steady_clock::now();
        for_each_fastq(sequence_file, [&](const FastqRecord &rec) {
            auto best = filter_best_locations(locate_res[read_id], static_cast<unsigned int>(rec.seq.size()));
            for (auto &[bucket_id, offset, segment_offset, votes, is_original] : best) {
                (void)segment_offset;
                const unsigned int map_qual = std::min(60u, 6 * votes);                      // :591
                const size_t ref_offset = static_cast<size_t>(bucket_offsets[bucket_id]) + offset;  // :592, 0-based
                sam << rec.id << '\t' << (is_original ? 0 : 16) << '\t' << bucket_name[bucket_id] << '\t'
                    << ref_offset + 1 << '\t' << map_qual << "\t*\t*\t0\t0\t" << rec.seq << '\t' << rec.qual << '\n';
                mapped_locations++;
            }
            read_id++;
        });
        const float s = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        std::cerr << "[BENCHMARK]\tTotal mapped locations: " << mapped_locations << " ("
                  << static_cast<float>(mapped_locations) / read_id << " per sequence).\n";
        std::cerr << "[BENCHMARK]\tTotal time used for alignment verification and output: " << s << " s ("
                  << s / mapped_locations * 1000 * 1000 << " μs per pairwise alignment).\n";
    }
};

}  // namespace bm
