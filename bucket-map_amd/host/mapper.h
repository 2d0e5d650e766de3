// mapper.h -- the drop-in boundary: the reference's abstract mapper interface, member for member
// (bucket_map/mapper/mapper.h:4-34).  bucket_locator talks to this and nothing else.
#pragma once

#include "bm_common.h"

#include <filesystem>
#include <string>
#include <utility>

namespace bm {

class mapper {
public:
    mapper() {}
    virtual ~mapper() = default;

    unsigned int num_records = 0;

    // Load the q-gram index files <index_dir>/<indicator>.{kmers_index,qgram}.
    virtual void load(std::filesystem::path const &index_dir, const std::string &indicator) = 0;

    // Map every read of a FASTQ file to its candidate buckets.  Element b of each vector lists the
    // (read index, window start) pairs mapped to bucket b: first = read as-is, second = reverse
    // complemented.
    virtual std::pair<segments_t, segments_t> map(std::filesystem::path const &sequence_file) = 0;

    // Release the index.
    virtual void reset() = 0;
};

}  // namespace bm
