// Parameter blocks and launchers shared by the kernels and the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace dpf {

static constexpr uint32_t NO_LEAF = 0xffffffffu;

// Everything k_hash needs besides the vectors (all pointers device memory).
struct HashParams {
    int32_t family;          // 0 angle, 1 pStable
    int32_t dim, F, L, m, pb;
    int32_t type_of_index;
    const double *fnsT;      // [dim][F]  transposed function pool
    const double *b;         // [F]       pStable offsets
    const int32_t *w;        // [F]       pStable widths
    const int32_t *chain_fn; // [L][m]
    const double *part_fns;  // [L][pb][32]
    const double *part_b;    // [L][pb]
    const int32_t *part_w;   // [L][pb]
    const double *fn_norm;   // [F]       Euclidean norms of the pool functions (matrix-core projection)
    int32_t sampling_index[32];
    int32_t angle_label_lut[29];
};

struct ProbeParams {
    // forest snapshot
    const uint32_t *roots;    // [L][P][SEG]
    const uint32_t *dnodes;   // [num_dnodes][DNS]
    const uint32_t *leaf_off; // [num_leaves + 1]
    const int32_t *ids;       // bucket contents
    int32_t L, pb, bucket_bits, seg, nbits, dns, max_level;
    int32_t steps, exclude_self;
    int32_t t_begin, t_end;   // table shard probed by this handle
    uint32_t sub_mask = 0xffffffffu;   // sub-indexes this handle probes (bit p = sub-index p)
    int32_t probe_own = 0;    // 1: walk the unflipped hash only (RDT:630-675) instead of its bit flips (RDT:742-797)
    // query batch
    const int32_t *qids;      // [nq] or nullptr
    const int32_t *hashes;    // [nq][L]
    const int32_t *parts;     // [nq][L]
    int64_t row_begin, row_end;
    int32_t chunk_words;      // LDS bitmap words per pass (mode 1)
    // mode 0 outputs
    unsigned long long *raw;  // [nq] ids visited (duplicates included)
    int64_t *bound;           // [nq] min(raw, rows in shard)
    unsigned long long *short_hash_events;
    uint32_t *lq_cnt;         // [num_leaves] optional: (query, bucket) incidence histogram
    uint2 *q_leaves;          // [nq][q_leaf_cap] optional: the distinct non-empty buckets each query reached,
                              // with the query's place in the bucket's list
    int32_t *q_leaf_cnt;      // [nq] their number (may exceed q_leaf_cap: list truncated)
    int32_t q_leaf_cap;
    int32_t *q_leaf_overflow; // set to 1 when any list was truncated
    // mode 1 inputs / outputs
    const int64_t *cand_off;  // [nq + 1] exclusive scan of bound
    int32_t *cand;            // candidate arena
    int32_t *cnt;             // [nq] set sizes
    // mode 2 (bucket -> queries lists + per-query seed candidates)
    const int64_t *lq_off;    // [num_leaves + 1] exclusive scan of lq_cnt
    const uint32_t *lq_off32 = nullptr;   // the same offsets in 32 bits (k_fill_lists: half the gathered table)
    uint32_t *lq_cursor;      // [num_leaves] zeroed
    int32_t *leaf_queries;    // [total pairs]
    int32_t *seed_cand;       // [nq][seed_cap] ids from table 0's buckets
    int32_t *seed_cnt;        // [nq]
    int32_t seed_cap;
    const int32_t *abort = nullptr;   // k_fill_lists: when set and non-zero the batch is being abandoned, write nothing
    // k_probe_lanes: the bit patterns with at most `steps` of the pb sub-index bits set (RDT:613-621: a query probes the
    // sub-indexes own ^ sub_delta[j]), filled by launch_probe_lanes
    int32_t nsub = 0;
    uint8_t sub_delta[32] = {0};
    // k_probe_lanes, binned form (bin_cursor != nullptr): the (bucket, query) incidences are not counted into lq_cnt but
    // written as pairs into the region of the bucket's BIN of pairs[].  A bin is a range of consecutive buckets made of
    // whole granules (1 << gran_shift buckets; bin_map[bucket >> gran_shift] is the bin), cut so that the bins of the
    // batch that seeded them held about equally many pairs.  Region b starts at bin_off[b], holds at most bin_cap[b]
    // pairs, bin_cursor[b] (zeroed) counts what was asked for; *bin_overflow is raised when a region is too small (the
    // batch is then run again the other way)
    uint32_t *bin_cursor = nullptr;
    const uint32_t *bin_cap = nullptr, *bin_off = nullptr;
    const uint16_t *bin_map = nullptr;
    uint2 *pairs = nullptr;
    int32_t gran_shift = 0, nbins = 0;
    int32_t *bin_overflow = nullptr;
};

constexpr int BIN_MAX = 1024;              // bins at most
constexpr int BIN_MAX_LEAVES = 8192;       // buckets of one bin at most
constexpr int BIN_MAX_GRANULES = 8192;     // granules of a forest at most
constexpr int BIN_LEAF_BITS = 22;          // a forest with bins has fewer than 1 << 22 buckets

// The bucket lists of a batch from the binned pairs (see ProbeParams): one workgroup per bin builds its buckets' incidence
// counts, list offsets, unit counts and the lists themselves in LDS.
struct BinListParams {
    const uint32_t *bin_cursor, *bin_cap, *bin_off;
    const uint32_t *bin_lo;           // [nbins + 1] first bucket of each bin
    const uint2 *pairs;
    int32_t nbins;
    int64_t nl;                       // buckets
    int64_t nq;                       // queries of the batch
    int32_t stage;                    // 0: the list entries are stored straight to leaf_queries[] (measurement), 1: staged in LDS
    const uint32_t *leaf_tile0;       // [nl + 1] first tile of each bucket
    int64_t cap_pairs;                // room in leaf_queries[]
    int32_t *flags;                   // flags[3]: the batch is abandoned
    uint32_t *lq_cnt;                 // [nl] out
    int64_t *lq_off;                  // [nl + 1] out
    int64_t *unit_off;                // [nl] out: units before the bucket WITHIN its bin
    int64_t *bin_units;               // [nbins] out: units of each bin
    int32_t *leaf_queries;            // out
};
void launch_bin_lists(const BinListParams &p, hipStream_t stream);
// bin_unit_base[b] = units of the bins before b; totals[0] = pairs, totals[1] = units; flags[3] raised when the units
// exceed cap_units
void launch_bin_prefix(const uint32_t *bin_cursor, const int64_t *bin_units, int32_t nbins, int64_t cap_units, int64_t *bin_unit_base,
                       int64_t *totals, int32_t *flags, hipStream_t stream);
// The bins for the batches to come from the incidence counts (lq_cnt) of a batch whose lists were built the other way:
// granule totals (gran_cnt, scratch of ngran words), then the cut into bins of about equally many pairs and at most
// BIN_MAX_LEAVES buckets each: bin_map[ngran], bin_lo[bins + 1], the regions' capacities (2 x count + 1024) and offsets;
// info[0] = the capacities' sum, info[1] = the number of bins (0: more than BIN_MAX, no bins)
void launch_bin_seed(const uint32_t *lq_cnt, int64_t nl, int32_t gran_shift, int32_t ngran, uint32_t *gran_cnt, uint16_t *bin_map,
                     uint32_t *bin_lo, uint32_t *bin_cap, uint32_t *bin_off, unsigned long long *info, hipStream_t stream);

// Bucket-major scan: every touched bucket's rows are loaded once and scored
// against all the queries that probe that bucket.
struct LeafScanParams {
    const int32_t *ids;          // bucket contents (shard-filtered)
    const int32_t *leaf_queries; // the queries probing each bucket, bucket-major
    int64_t row_begin, row_end;
    const int32_t *qids;
    int32_t exclude_self;
    int32_t *tau;                // [nq] filter threshold (l2: radius, dot: product), order-preserving int encoding; tightened in flight
    const float4 *qc;            // [nq] per-query record of the scan, written with the query rows' fp16 conversion
    int32_t k;
    int32_t *emit;               // [nq][emit_cap] forwarded ids
    float *emit_v;               // [nq][emit_cap] pessimistic score bound of each forwarded row
    int32_t *emit_cnt;           // [nq] may exceed emit_cap: overflow
    int32_t emit_cap;
    int32_t debug;               // ablation bits, honoured only by kernels built with -DDPF_ABLATE
    const uint16_t *Q16;         // fp16 copies: queries [nq][ks*32], rows [rows][ks*32]
    const uint16_t *base16;
    const float *slot_norm;      // squared norms of the fp16 copies (fp32 sums), in bucket order beside ids[]
    const uint4 *units;          // work units of the batch, see k_units
    int64_t unit_begin, unit_end; // the slice of units[] this launch works through ...
    // ... or, when the host has not read the batch's unit count (totals != nullptr): totals[1] units in all,
    // a launch takes a slice of them given in thousandths; nothing when *abort is non-zero
    const int64_t *totals = nullptr;
    const int32_t *abort = nullptr;
    int32_t permille_lo = 0, permille_hi = 0;   // the launch takes the units [totals[1] * lo / 1000, totals[1] * hi / 1000)
    int32_t num_cus;
    unsigned int *tickets;       // num_counters ticket counters of this launch, 32 words apart, zeroed
    int32_t num_counters;
    int32_t ks;
    // l2, rows cut at a row length the scan has a kernel for (leafscan_f16_filter_dim): the up to 8 dimensions behind the
    // cut, fp16, zero padded - the tile's in bucket order beside ids[], the queries' by query; nullptr: whole rows
    const uint4 *slot_tail = nullptr;
    const uint4 *q_tail = nullptr;
    // int8 scan (l2, integer-valued rows and queries inside one window of 256 values): Q16 / base16 then point at
    // int8 rows of ks * 64 bytes (values minus the window's centre), qc holds the shifted queries' records and
    // slot_nj8[] = -floor(|x'|^2 / 2) of the shifted rows in bucket order beside ids[]
    int32_t i8 = 0;
    const int32_t *slot_nj8 = nullptr;
    // a launch runs only when (*gate != 0) == (gate_want != 0) (gate == nullptr: always): the int8 and the fp16
    // kernel of one slice are both enqueued and the word the query conversion wrote picks one
    const int32_t *gate = nullptr;
    int32_t gate_want = 0;
    // 1: base16 holds the rows in bucket order (row of id slot s at s * ks * 32 elements, beside ids[s]) instead of by id:
    // a tile's rows are one contiguous run
    int32_t rows_in_slot_order = 0;
    // what the index arithmetic of the tile and query loaders may reach (checked only by builds made with
    // -DDPF_DEBUG_BOUNDS, which count every access outside into *dbg_violations and clamp it)
    int64_t dbg_slots = 0, dbg_rows = 0, dbg_nq = 0, dbg_pairs = 0;
    unsigned long long *dbg_violations = nullptr;
};

struct RerankParams {
    const void *Q;            // [nq] padded rows
    int64_t q_stride_bytes;
    const void *base;         // rows row_begin .. row_end-1, padded
    int64_t base_stride_bytes;
    int32_t chunks16;         // 16-byte pieces per padded row
    int32_t k;
    int64_t row_begin, row_end;
    const int64_t *cand_off;  // [nq + 1]; nullptr: candidate lists sit at q * cand_stride
    const int32_t *cnt;       // [nq] or nullptr (then cand_off[q+1]-cand_off[q])
    int64_t cand_stride;      // used when cand_off is nullptr; cnt is clamped to it
    const int32_t *cand;
    int32_t *out_ids;         // [nq][k]
    double *out_scores;       // [nq][k]
};

void launch_hash(int dtype, const void *X, int64_t x_stride, int64_t n, const HashParams &p,
                 int32_t *hashes, int32_t *parts, hipStream_t stream);
// the same kernel over the rows row_list[0 .. *row_count) (device memory), n = capacity of the list
void launch_hash_rows(int dtype, const void *X, int64_t x_stride, int64_t n, const HashParams &p, int32_t *hashes, int32_t *parts,
                      const int32_t *row_list, const uint32_t *row_count, hipStream_t stream);
size_t hash_shmem_bytes(const HashParams &p);

// Matrix-core projection (dpf_project.hip): the same hashes and sub-index ids as launch_hash, bit for bit.
struct ProjectWs {
    void *codes = nullptr;           // angle: uint16 [n][ceil(F/16)] sign bits; pStable: int32 [n][F] bucket indices
    double *row_norm = nullptr;      // [n]
    uint32_t *row_flag = nullptr;    // [n] 1 = undecided, hashed by the exact kernel afterwards
    int32_t *flag_rows = nullptr;    // [n] the undecided rows
    uint32_t *flag_count = nullptr;  // [0] undecided rows of this call
    unsigned long long *flag_total = nullptr;   // running total over all calls (statistics)
};
size_t project_codes_bytes(const HashParams &p, int64_t n);
hipError_t launch_hash_mfma(int dtype, const void *X, int64_t x_stride, int64_t n, const HashParams &p, const ProjectWs &ws,
                            int32_t *hashes, int32_t *parts, hipStream_t stream);
void launch_probe(int mode, const ProbeParams &p, int64_t nq, hipStream_t stream);
// what launch_probe(0, ...) computes for the bucket-major path (lq_cnt and q_leaves set), one lane per (query, table,
// sub-index) instead of one lane per probe; false (nothing launched) when the configuration is outside its limits
bool launch_probe_lanes(ProbeParams p, int64_t nq, hipStream_t stream);
void launch_fill_lists(const ProbeParams &p, int64_t nq, hipStream_t stream);
void launch_seed_sample(const ProbeParams &p, int64_t nq, hipStream_t stream);
void launch_leafscan(int metric, const LeafScanParams &p, hipStream_t stream);
// flags[1] != 0 (a truncated bucket list) or a total beyond its capacity sets flags[3] when `guarded`
void launch_bucket_scans(const uint32_t *lq_cnt, const uint32_t *leaf_tile0, int64_t nl, int64_t *lq_off, int64_t *unit_off,
                         int64_t *totals, bool guarded, int64_t cap_pairs, int64_t cap_units, int32_t *flags, uint32_t *lq_off32,
                         hipStream_t stream);
// (bin_unit_base != nullptr: unit_off[] counts within the bucket's bin and bin_unit_base[bin_map[bucket >> gran_shift]] is added)
void launch_units(const uint32_t *lq_cnt, const int64_t *lq_off, const uint32_t *leaf_tile0, const uint4 *tiles, int64_t ntiles,
                  const int64_t *unit_off, uint4 *units, const int32_t *abort, hipStream_t stream,
                  const int64_t *bin_unit_base = nullptr, const uint16_t *bin_map = nullptr, int32_t gran_shift = 0);
void launch_tile_maxnorm(uint4 *tiles, int64_t ntiles, const int32_t *ids, const float *norms, const float *errs, int64_t row_begin,
                         int64_t row_end, float *slot_norms, hipStream_t stream, const uint4 *tails = nullptr, uint4 *slot_tails = nullptr,
                         uint32_t *bounds = nullptr);
// dst[s] = the fp16 row of ids[s] (zeros for ids outside [row_begin, row_end)): the rows once more, in bucket order
void launch_rows_to_slot_order(const int32_t *ids, int64_t nslots, const uint16_t *base16, int row_elems, int64_t row_begin,
                               int64_t row_end, uint16_t *dst, hipStream_t stream);
int leafscan_f16_steps(int dim);
int leafscan_f16_filter_dim(int dim, int metric);
int leafscan_tile_rows(int ks);
void launch_refine(int metric, const int32_t *emit, const float *emit_v, const int32_t *emit_cnt, int32_t cap, int k, int64_t nq,
                   const float *q16_err, int32_t *tau, hipStream_t stream);
// what the conversion of a QUERY batch writes besides the fp16 rows: each query's first threshold (from the k-th
// best exact score of its seed sample) and the 16-byte record the scan reads
struct TauParams {
    int metric = 1;
    const double *seed_scores = nullptr;   // [nq][k]
    const int32_t *seed_ids = nullptr;     // [nq][k]
    int k = 0;
    const int32_t *qids = nullptr;
    int32_t *tau = nullptr;                // [nq]
    float4 *qc = nullptr;                  // [nq]
};
// dim_all > dim: the dimensions [dim, dim_all) (at most 8) go to tails[r] instead of the row; norms cover the row,
// errs all dim_all dimensions
// mu (optional, [dim_all] floats): subtracted from every row before the conversion
void launch_rows_to_f16(const float *src, int64_t src_stride, int dim, int64_t n, uint16_t *dst, int dst_stride,
                         float *norms, float *errs, hipStream_t stream, const TauParams *tau = nullptr, int dim_all = 0,
                         uint4 *tails = nullptr, const float *mu = nullptr);
void launch_col_sums(const float *src, int64_t src_stride, int dim, int64_t n, float *sums, hipStream_t stream);
// int8 copies for the exact integer scan.  launch_i8_range: out[0] = smallest value (order-preserving int image of the
// float), out[1] = largest, out[2] != 0 when a value is no integer (out[] preset to INT32_MAX, INT32_MIN, 0).
// launch_rows_to_i8: dst rows of dst_stride bytes (a multiple of 64) = value - offset, zero padded; *bad is raised when a
// value minus the offset is no integer in [-128, 127]; base rows (qc == nullptr) leave |x'|^2 in norm2[r], query rows the
// scan's record {|q'|^2 / 2, 0, |q'|^2, query id} in qc[r].  launch_slot_nj8: -floor(norm2 / 2) in bucket order.
void launch_i8_range(const float *src, int64_t src_stride, int dim, int64_t n, int32_t *out, hipStream_t stream);
void launch_rows_to_i8(const float *src, int64_t src_stride, int dim, int64_t n, float offset, int8_t *dst, int dst_stride,
                       int32_t *norm2, float4 *qc, const int32_t *qids, int32_t *bad, hipStream_t stream);
void launch_slot_nj8(const int32_t *ids, int64_t nslots, const int32_t *norm2, int64_t row_begin, int64_t row_end,
                     int32_t *slot_nj8, hipStream_t stream);
// first thresholds (l2) and scan records of a batch from the fp16 copies of the queries and of their seed rows
void launch_seed_radius(const int32_t *cand, const int32_t *cnt, int64_t cand_stride, const uint16_t *Q16, const uint16_t *B16, int ks,
                        const float *q_norm, const float *q_err, const float *b_err, const uint4 *q_tail, const uint4 *b_tail,
                        int64_t row_begin, int64_t row_end, const TauParams &tau, int64_t nq, hipStream_t stream);
size_t unique_lists_lds(int32_t longest);
// what k_unique_lists needs to drop the entries of a forward list that the query's final threshold has passed by
struct PruneParams {
    const float *emit_v = nullptr;   // [nq][cap] the scan's bound per entry; nullptr: no pruning
    const int32_t *tau = nullptr;    // [nq] final thresholds
    const float4 *qc = nullptr;      // [nq] scan records (l2: A, dot: bound of |q~|; |q - q~|; |q~|^2; id) of the fp16 form ...
    const float4 *qc8 = nullptr;     // ... and of the int8 form (or nullptr)
    const uint4 *q_tail = nullptr;   // [nq] query tails of cut rows or nullptr
    const int32_t *gate = nullptr;   // != nullptr and *gate == 0: the int8 form of the scan ran
    int metric = 1;
    float c32 = 0.f, xr_max = 0.f, cx_max = 0.f, xt_max = 0.f;
};
void launch_unique_lists(int32_t *emit, const int32_t *emit_cnt, int32_t cap, int32_t longest, int32_t *unique_cnt, int64_t nq,
                         hipStream_t stream, const PruneParams *prune = nullptr);
float leafscan_c32_of(int stride);   // the scan's rounding allowance per unit of |q~|^2 + |x~|^2 for rows of `stride` fp16 elements
// flag[0] = overflowed lists (their queries in list[]), flag[2] = longest list; then the 40 bytes at `flag` (flags,
// the two scan totals and the int8 gate word) are copied to host_record (pinned host memory)
void launch_overflow_list(const int32_t *emit_cnt, int64_t nq, int32_t cap, int32_t *flag, int32_t *list, int32_t *host_record,
                          hipStream_t stream);
void launch_gather_queries(const int32_t *list, int64_t n, const void *Q, int64_t row_bytes, const int32_t *qids, void *subQ,
                           int32_t *sub_qids, hipStream_t stream);
void launch_scatter_results(const int32_t *list, int64_t n, int k, const int32_t *sub_ids, const double *sub_scores, int32_t *ids,
                            double *scores, hipStream_t stream);
void launch_scan_i64(const int64_t *in, int64_t n, int64_t *off, hipStream_t stream);
void launch_scan_i32(const int32_t *in, int64_t n, int64_t *off, hipStream_t stream);
void launch_compact(const int32_t *src, const int64_t *src_off, const int32_t *cnt,
                    const int64_t *dst_off, int32_t *dst, int64_t nq, hipStream_t stream);
void launch_rerank(int dtype, int metric, const RerankParams &p, int64_t nq, hipStream_t stream);
int rerank_max_chunks16();
bool rerank_short_applies(const RerankParams &p);
// few_rows: the lists hold a few tens of rows (rows of more than 256 bytes are then scored 8 at a time per wave, not 32)
void launch_rerank_short(int dtype, int metric, const RerankParams &p, int64_t nq, hipStream_t stream, bool few_rows = false);
void launch_merge_topk(int metric, const int32_t *in_ids, const double *in_scores, int parts, int64_t ids_stride,
                       int64_t scores_stride, int64_t nq, int k, int32_t *out_ids, double *out_scores, hipStream_t stream);
void launch_copy_rows(int dtype, const void *src, int64_t n, int dim, void *dst, int64_t dst_stride,
                      hipStream_t stream);
int configure_kernels();

}  // namespace dpf
