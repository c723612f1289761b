// Device-side construction of the forest snapshot (see dpf_build.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "dpf_forest.hpp"

namespace dpf {

// Same layout and numbering as ForestSnapshot, in device memory (hipMalloc'ed; the caller
// takes the pointers over or calls release()).
struct DeviceForest {
    uint32_t *roots = nullptr;     // [L * P * SEG]
    uint32_t *dnodes = nullptr;    // [(num_dnodes + 1) * DNS], node 0 reserved
    uint32_t *leaf_off = nullptr;  // [num_leaves + 1]
    int32_t *ids = nullptr;        // [num_ids]
    int64_t num_dnodes = 0, num_leaves = 0, num_ids = 0, max_leaf = 0, misflag_events = 0;
    std::vector<int64_t> part_counts;   // [L * P]
    void release();
};

// d_hashes / d_parts: [n][L] in device memory, ids are the row numbers.  0 on success, -1 when an arena
// would exceed 32-bit indexing, -2 on a HIP error (err holds the message).
int build_forest_device(const ForestGeometry &g, const int32_t *d_hashes, const int32_t *d_parts, int64_t n, DeviceForest &out,
                        std::string &err);
int filter_forest_rows(DeviceForest &f, int64_t lo, int64_t hi, std::string &err);
int make_forest_tiles(const DeviceForest &f, int rows_per_tile, uint32_t **leaf_tile0, uint4 **tiles, int64_t *ntiles, std::string &err);

}  // namespace dpf
