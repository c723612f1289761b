// Host-side construction of the flattened forest snapshot that the probe
// kernels read.  The reference grows each table by pointer insertion under
// locks (RandomDrawTreeMap.java:1558-1790); here the final shape of every
// trie is derived from the sorted hashes in one pass.
#pragma once

#include <cstdint>
#include <vector>

namespace dpf {

struct ForestGeometry {
    int32_t num_tables;       // L
    int32_t partition_bits;   // pb, P = 1 << pb sub-indexes per table
    int32_t bucket_bits;      // BUCKET_LENGTH (RandomDrawTreeMap.java:435-438)
    int32_t dir_node_size;    // DIRECTORY_NODE_SIZE
    int32_t chain_length;     // TOTAL_HASH_LENGTH
    int32_t bucket_overflow;  // BUCKET_OVERFLOW
    // derived exactly as RandomDrawTreeMap.java:435-459
    int32_t seg;              // 1 << (32 - bucket_bits)
    int32_t nbits;            // log2(dir_node_size)
    int32_t max_level;        // (chain_length - (32 - bucket_bits)) / nbits - 1
    int32_t num_parts;        // 1 << partition_bits
    bool derive();            // false when the combination is unsupported
};

// Snapshot layout (all indices global across tables):
//   roots[(t * P + p) * SEG + seg]  -> d-node index, 0 = empty
//   dnodes[node * DNS + slot]       -> child word: 0 = empty,
//                                      (leaf  << 1) | 1  = bucket,
//                                      (dnode << 1)      = directory (dnode >= 1)
//   leaf_off[leaf] .. leaf_off[leaf+1] -> range in ids[] (ascending ids)
struct ForestSnapshot {
    std::vector<uint32_t> roots;
    std::vector<uint32_t> dnodes;     // node 0 is a reserved all-empty node
    std::vector<uint32_t> leaf_off;   // num_leaves + 1
    std::vector<int32_t> ids;
    std::vector<int64_t> table_dnodes, table_leaves;
    std::vector<int64_t> part_counts; // [L * P] ids per sub-index
    int64_t max_leaf = 0;
    int64_t misflag_events = 0;
};

// hashes / parts: [n][L] row-major, ids are the row numbers 0..n-1 (insertion
// order of DensevectorRDFInit.newFastFit, DensevectorRDFInit.scala:127-151).
// Returns 0, or a negative value when an arena would exceed 32-bit indexing.
int build_forest(const ForestGeometry &g, const int32_t *hashes, const int32_t *parts,
                 int64_t n, int threads, ForestSnapshot &out);

}  // namespace dpf
