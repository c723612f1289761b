// C ABI of the DPF batch-query engine (see include/dpf.h for the contract and
// for the reference interfaces each entry point replaces).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/dpf.h"
#include "dpf_device.hpp"
#include "dpf_build.hpp"
#include "dpf_forest.hpp"

using namespace dpf;

namespace {

std::string g_create_error;

// ---------------------------------------------------------------------------
// The scratch of dpf_query_topk_device_static, the entry point that allocates nothing: one array per GPU
// in the library's device image (each GPU the process touches holds its copy until the process ends).
// Host code never uses the array's name as an address: dpf_create asks the runtime once per device where
// that device's copy lives (scratch_of) and everything below works with that pointer.
// ---------------------------------------------------------------------------
constexpr size_t SCRATCH_MAX_QUERIES = 1024;                       // queries per call
constexpr size_t SCRATCH_QPAD_BYTES = (size_t)4 << 20;             // padded query rows
constexpr size_t SCRATCH_HASH_WORDS = (size_t)128 << 10;           // nq * num_tables, hashes and sub-indexes each
constexpr size_t SCRATCH_ARENA_IDS = (size_t)6 << 20;              // candidate lists of the whole call
constexpr size_t SCRATCH_OFF_QPAD = 0;
constexpr size_t SCRATCH_OFF_HASHES = SCRATCH_OFF_QPAD + SCRATCH_QPAD_BYTES;
constexpr size_t SCRATCH_OFF_PARTS = SCRATCH_OFF_HASHES + SCRATCH_HASH_WORDS * 4;
constexpr size_t SCRATCH_OFF_RAW = SCRATCH_OFF_PARTS + SCRATCH_HASH_WORDS * 4;
constexpr size_t SCRATCH_OFF_BOUND = SCRATCH_OFF_RAW + SCRATCH_MAX_QUERIES * 8;
constexpr size_t SCRATCH_OFF_OFF = SCRATCH_OFF_BOUND + SCRATCH_MAX_QUERIES * 8;
constexpr size_t SCRATCH_OFF_CNT = SCRATCH_OFF_OFF + (SCRATCH_MAX_QUERIES + 32) * 8;
constexpr size_t SCRATCH_OFF_ARENA = SCRATCH_OFF_CNT + SCRATCH_MAX_QUERIES * 4;
constexpr size_t SCRATCH_BYTES = SCRATCH_OFF_ARENA + SCRATCH_ARENA_IDS * 4;       // 29.03 MiB
static_assert(SCRATCH_OFF_ARENA % 256 == 0 && SCRATCH_OFF_CNT % 256 == 0 && SCRATCH_OFF_OFF % 256 == 0, "256-byte aligned slices");

constexpr int SCRATCH_MAX_DEVICES = 64;
constexpr int MAX_BINS = BIN_MAX;                                     // bins of buckets of the binned list building
// Per GPU: where its copy of the array lives, an event recorded behind the last kernel of the last call that used it
// (the next call's stream waits for it, so that calls on different streams or from different handles never overlap
// on the device), and a lock held for the host part of a call (a second thread calling meanwhile is refused with
// DPF_EBUSY rather than left to race).  Event and pointer are made at dpf_create: the entry point itself creates nothing.
struct ScratchDev {
    void *ptr = nullptr;
    hipEvent_t done = nullptr;
    bool recorded = false;
    std::mutex busy;
};
ScratchDev g_scratch[SCRATCH_MAX_DEVICES];
std::mutex g_scratch_mutex;

}  // namespace

static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

namespace {

// This device's copy of the scratch array (the device must be current); nullptr + *err on failure.
void *scratch_of(int device, hipError_t *err)
{
    *err = hipSuccess;
    if (device < 0 || device >= SCRATCH_MAX_DEVICES) { *err = hipErrorInvalidDevice; return nullptr; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    ScratchDev &sd = g_scratch[device];
    if (!sd.ptr) {
        void *ptr = nullptr;
        *err = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (*err != hipSuccess) return nullptr;
        hipEvent_t ev = nullptr;
        *err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (*err != hipSuccess) return nullptr;
        sd.done = ev;
        sd.ptr = ptr;
    }
    return sd.ptr;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    void adopt(void *q, size_t bytes) { release(); p = q; cap = bytes; }
    // grows to `bytes`, keeping the first `keep` bytes
    hipError_t grow(size_t keep, size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        void *q = nullptr;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess) { e = hipMalloc(&q, bytes); want = bytes; }
        if (e != hipSuccess) return e;
        if (p && keep > 0) e = hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { (void)hipFree(q); return e; }
        release();
        p = q;
        cap = want;
        return hipSuccess;
    }
    template <typename T> T *as() const { return (T *)p; }
};

// pinned host memory, grown on demand
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// true for memory the device can copy from / to directly (hipHostMalloc, hipHostRegister, dpf_host_alloc)
bool host_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

}  // namespace

struct dpf_index {
    dpf_config cfg{};
    ForestGeometry geo{};
    std::string err;
    int64_t row_begin = 0, row_end = 0;     // resolved shard
    size_t elem = 4;
    int64_t dpad = 0;                       // padded row length in elements
    bool have_family = false, have_part = false;

    std::vector<double> fns, b, part_fns, part_b;
    std::vector<int32_t> w, chain_fn, part_w;
    DevBuf d_fnsT, d_b, d_w, d_chain, d_part_fns, d_part_b, d_part_w, d_fn_norm;
    DevBuf ws_proj_codes, ws_proj_norm, ws_proj_flag, ws_proj_rows;   // matrix-core projection (dpf_project.hip)
    unsigned char *d_proj_counters = nullptr;                         // [0] undecided rows of the last call (u32), [8] running total (u64),
                                                                      // [16] out-of-bounds indices counted by a -DDPF_DEBUG_BOUNDS build of the scan (u64)
    HashParams hp{};

    int64_t n = 0;
    DevBuf d_hashes, d_parts;                // hashes / sub-indexes of all n ids, [n][L]
    std::vector<int32_t> h_hashes, h_parts;  // host mirror of the two, refreshed on demand
    bool host_hashes_valid = true;
    std::vector<int64_t> part_counts;        // [L][P] ids per sub-index
    DevBuf d_base;
    int64_t base_rows = 0;                   // resident rows: ids row_begin .. row_begin+base_rows-1

    DevBuf d_roots, d_dnodes, d_leaf_off, d_ids;
    DevBuf ws_emit_ucnt;                     // distinct ids per forward list
    DevBuf ws_emit_v, ws_leaf_units, ws_unit_off, ws_q_leaves, ws_q_leaf_cnt;
    DevBuf d_leaf_tile0, ws_units;           // first tile of each bucket; work units of the batch
    int num_cus = 256;
    DevBuf d_tiles;                          // {bucket, first id slot, rows, 0} per tile
    DevBuf d_base16e, ws_q16e;               // distances between the rows and their fp16 copies
    DevBuf d_base16n, ws_q16n;               // their squared norms
    DevBuf d_slot_norm;                      // the rows' squared norms once more, in bucket order beside d_ids
    DevBuf d_base16t, d_slot_tail, ws_q16t;  // tails: by row, in bucket order beside d_ids, of the query batch
    DevBuf ws_qc;                            // per-query record of the scan
    DevBuf ws_ovf;                           // queries whose forward list overflowed
    int32_t emit_cap = 8192;                 // entries per forward list
    DevBuf sub_q, sub_qids, sub_ids, sub_scores, sub_ws[6];   // their batch on the gather path
    DevBuf d_base16, ws_q16;                 // fp16 copies of the resident rows / of the query batch
    DevBuf d_slot16;                         // DPF_FLAG_BUCKET_ORDER_ROWS: the fp16 rows once more, in bucket order beside d_ids
    bool slot16_ok = false;
    // int8 form of the scan: kept beside the fp16 copy when every resident row is integer-valued inside one window of
    // 256 values (l2 handles, rows of up to 256 dimensions); a batch whose queries fit the window too is scanned in int8
    DevBuf d_base8, d_norm8, d_slot_nj8, d_i8_probe;   // rows minus the window's centre, their |x'|^2, -floor(|x'|^2 / 2) in bucket order
    DevBuf ws_q8, ws_qc8;                    // the query batch's int8 rows and scan records
    bool i8_ok = false;
    DevBuf d_bounds;                         // k_tile_maxnorm's maxima over the tiles
    float tile_bounds[3] = {0.f, 0.f, 0.f};  // the largest |x~| bound and rounding-distance bound of a tile, the largest squared tail norm
    DevBuf d_mu;                             // the centre the fp16 copies of an l2 handle are taken around ([dpad] floats)
    bool centred = false;
    int ks8 = 0;                             // 64-byte steps per int8 row
    float i8_offset = 0.f;                   // the window's centre
    bool last_scan_int8 = false;
    int ks16 = -1;                           // 32-dimension fp16 steps per padded row (-1: no fp16 scan)
    int filter_dim = 0;                      // dimensions the fp16 rows keep; the rest (at most 8) travels as 16-byte tails
    int64_t ntiles = 0, nutiles = 0, num_leaves = 0;
    std::vector<uint32_t> h_leaf_len;        // rows per bucket of the uploaded snapshot
    bool bucket_major = false;               // the bucket-major rerank applies to this handle
    bool fitted = false;
    dpf_stats st{};

    // per-batch workspaces
    DevBuf ws_qpad, ws_hashes, ws_parts, ws_raw, ws_bound, ws_off, ws_cnt, ws_arena, ws_misc;
    DevBuf ws_hq, ws_hqids, ws_hids, ws_hscores;     // device copies for the host-pointer entry points
    DevBuf ws_stage;                                  // fit staging
    PinBuf h_stage;                                   // host-pointer entry points: pinned staging of pageable arguments / results
    DevBuf ws_lq_off, ws_lq_cursor, ws_leaf_queries, ws_seed_cand, ws_seed_cnt, ws_seed_ids,
        ws_seed_scores, ws_emit, ws_emit_cnt, ws_flag, ws_tau;
    hipEvent_t ev_scan[6]{};                          // begin and end of up to three passes of the scan
    int scan_launches = 0;
    bool ev_scan_used[3] = {false, false, false};
    int64_t prev_units = -1;                 // work units of the handle's last bucket-major batch (-1: none yet)
    unsigned long long *d_short_hash = nullptr;
    int64_t *pinned = nullptr;                        // host-pinned scratch (12 x int64)
    int32_t *h_sets = nullptr;                        // dpf_query_batch_view's result lists: pinned host memory, grown as needed
    size_t h_sets_cap = 0;
    DevBuf ws_compact;                                // the same lists packed back to back on the device
    bool sets_valid = false;                          // ws_compact / ws_misc / the padded query rows hold the Sets of dpf_query_batch_resident
    std::vector<int64_t> sets_off;                    // their offsets
    const void *sets_qpad = nullptr;                  // their padded query rows (ws_hq or ws_qpad)
    bool sizes_carried = false;                       // a bucket-major batch has sized leaf_queries[] / units[] already
    hipEvent_t ev[6]{};
    hipStream_t side_stream = nullptr;                // the list-building chain of a batch runs here, beside the first radii
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork0 = nullptr, ev_conv = nullptr;
    hipEvent_t ev_stage[2] = {nullptr, nullptr};      // the two halves of h_stage when results leave through it in pieces
    int last_path = 0;
    bool profiling = false;
    bool last_timed = false;                          // the last top-k batch recorded its events
    int64_t last_nq = 0;
    int32_t last_k = 0;
    bool last_valid = false;
    int32_t probe_mode = DPF_PROBE_FLIPS;             // dpf_set_probe_mode
    unsigned char *scratch_ptr = nullptr;             // this device's copy of the static scratch array
    // dpf_set_option values (0 / default = the library's own choice)
    int32_t opt_emit_cap = 0, opt_bucket_list_cap = 0, opt_first_permille = 150, opt_seed_cap = -1;   // seed sample: -1 = by the forest's bucket sizes
    int32_t auto_seed_cap = 256;             // what -1 stands for, set when the forest is built (seed_sample_for)
    int32_t opt_scan_passes = 0, opt_early_permille = 40;   // passes: 0 = by row length (3 from 16 steps on), 2 or 3
    bool opt_defer_build = false;                     // dpf_fit appends rows and hashes only; the forest is built by the first dpf_fit after it is cleared
    int64_t opt_expected_rows = 0;                    // ids the caller is going to fit in all: buffers are sized for them at once
    bool build_pending = false;                       // rows were appended under opt_defer_build and the forest does not cover them yet
    // Binned list building (batches with carried sizes): the buckets are grouped into nbins bins, ranges of whole granules
    // (1 << gran_shift buckets, ngran of them; d_bin_map: granule -> bin, d_bin_lo: first bucket of each bin) that held about
    // equally many incidences in the last batch sized on the host; the counting pass writes (bucket, query) pairs into
    // per-bin regions of ws_pairs (capacities d_bin_cap, offsets d_bin_off: twice what that batch needed), one workgroup
    // per bin then builds the lists in LDS
    int32_t gran_shift = 0, ngran = 0, nbins = 0;
    bool bins_seeded = false;
    DevBuf d_bin_cap, d_bin_off, d_bin_map, d_bin_lo, d_gran_cnt, ws_pairs, ws_bin_units, ws_bin_unit_base;
    int32_t opt_list_bins = 1;
    int32_t opt_probe_lanes = -1;                     // -1: by the number of (table, sub-index) pairs a query probes
    int32_t last_q_leaf_cap = 0;                      // bucket-list capacity of the last bucket-major batch
    bool opt_sync_sizes = false, opt_short_rerank = true, opt_exact_seeds = false, opt_keep_forwarded = false;
    int32_t opt_side_stream = -1;                     // 0: one stream; 1: conversions and lists on the side stream; 2: lists only; -1: by the forest
    int64_t opt_projection_min_rows = 4096;           // batches of at least this many vectors hash on the matrix cores (0: never)
};

#define HIPCHECK(ix, call)                                                              \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            (ix)->err = std::string(#call) + ": " + hipGetErrorString(e_);              \
            return DPF_EHIP;                                                            \
        }                                                                               \
    } while (0)

static int fail(dpf_index *ix, int code, const std::string &msg)
{
    if (ix) ix->err = msg; else g_create_error = msg;
    return code;
}

static void resolve_rows(dpf_index *ix)
{
    ix->row_begin = ix->cfg.row_begin < 0 ? 0 : ix->cfg.row_begin;
    ix->row_end = ix->cfg.row_end < 0 ? INT64_MAX : ix->cfg.row_end;
}

extern "C" const char *dpf_last_error(const dpf_index *ix)
{
    return ix ? ix->err.c_str() : g_create_error.c_str();
}

extern "C" int dpf_create(const dpf_config *cfg, dpf_index **out)
{
    if (!cfg || !out) return fail(nullptr, DPF_EINVAL, "dpf_create: null argument");
    *out = nullptr;
    dpf_index *ix = new dpf_index();
    ix->cfg = *cfg;
    ForestGeometry &g = ix->geo;
    g.num_tables = cfg->num_tables;
    g.partition_bits = cfg->partition_bits;
    g.bucket_bits = cfg->bucket_bits;
    g.dir_node_size = cfg->dir_node_size;
    g.chain_length = cfg->chain_length;
    g.bucket_overflow = cfg->bucket_overflow;
    std::string why;
    if (cfg->family != DPF_FAMILY_ANGLE && cfg->family != DPF_FAMILY_PSTABLE) why = "family must be angle or pStable";
    else if (cfg->dim < 1 || cfg->num_functions < 1 || cfg->num_tables < 1) why = "dim, num_functions and num_tables must be positive";
    else if (cfg->chain_length < 1 || (cfg->family == DPF_FAMILY_ANGLE && cfg->chain_length > 32)) why = "chain_length must be in [1, 32] for the angle family";
    else if (!(cfg->flags & DPF_FLAG_HASH_ONLY) && cfg->family == DPF_FAMILY_ANGLE && cfg->chain_length != 32)
        // RandomDrawTreeMap.java:953 consumes the LOW bits of the hash while angle hashes are
        // left-aligned (AngleHashFamily.scala:194): shorter chains put every id in slot 0.
        why = "the angle family needs chain_length = 32";
    else if (!(cfg->flags & DPF_FLAG_HASH_ONLY) && !g.derive()) why = "unsupported (bucket_bits, dir_node_size, chain_length, partition_bits, bucket_overflow) combination";
    else if (cfg->dtype != DPF_F32 && cfg->dtype != DPF_F64) why = "dtype must be DPF_F32 or DPF_F64";
    else if (cfg->top_k < 1 || cfg->top_k > 64) why = "top_k must be in [1, 64]";
    else if (cfg->metric != DPF_METRIC_DOT && cfg->metric != DPF_METRIC_L2) why = "metric must be dot or l2";
    else if (cfg->type_of_index < 0 || cfg->type_of_index > 3) why = "unknown type_of_index";
    else if (cfg->exclude_self < 0 || cfg->exclude_self > 2) why = "unknown exclude_self mode";
    else if (cfg->table_begin < 0 || cfg->table_begin > cfg->num_tables || cfg->table_end > cfg->num_tables ||
             (cfg->table_end > 0 && cfg->table_end < cfg->table_begin)) why = "table shard outside [0, num_tables]";
    if (why.empty() && !(cfg->flags & DPF_FLAG_HASH_ONLY) && cfg->partition_bits > 5)
        why = "partition_bits must be at most 5 (32 sub-indexes per table)";
    if (why.empty() && cfg->subindex_groups > 1 &&
        (cfg->subindex_group < 0 || cfg->subindex_group >= cfg->subindex_groups || cfg->subindex_groups > (1 << cfg->partition_bits)))
        why = "sub-index shard outside [0, subindex_groups) or more groups than sub-indexes";
    if (!why.empty()) { delete ix; return fail(nullptr, DPF_EINVAL, "dpf_create: " + why); }
    if (cfg->family == DPF_FAMILY_PSTABLE && cfg->chain_length > 1024) { delete ix; return fail(nullptr, DPF_EINVAL, "dpf_create: chain_length too large"); }
    ix->elem = cfg->dtype == DPF_F32 ? 4 : 8;
    const int64_t per16 = 16 / (int64_t)ix->elem;
    ix->dpad = (cfg->dim + per16 - 1) / per16 * per16;
    if (cfg->dtype == DPF_F32 && !(cfg->flags & DPF_FLAG_GATHER_ONLY)) {
        // the bucket-major scan works on fp16 copies of up to 1024 dimensions
        ix->filter_dim = (cfg->flags & DPF_FLAG_FULL_FILTER) ? cfg->dim : leafscan_f16_filter_dim(cfg->dim, cfg->metric);
        ix->ks16 = leafscan_f16_steps(ix->filter_dim);
        ix->bucket_major = ix->ks16 > 0;
    }
    if (ix->dpad * (int64_t)ix->elem / 16 > rerank_max_chunks16()) { delete ix; return fail(nullptr, DPF_EINVAL, "dpf_create: dim too large for the rerank kernel"); }
    resolve_rows(ix);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) { delete ix; return fail(nullptr, DPF_EHIP, "dpf_create: no HIP device (this library has no CPU path)"); }
    if (cfg->device < 0 || cfg->device >= ndev) { delete ix; return fail(nullptr, DPF_EINVAL, "dpf_create: bad device ordinal"); }
    e = hipSetDevice(cfg->device);
    if (e == hipSuccess && configure_kernels() != 0) e = hipErrorInvalidValue;
    if (e == hipSuccess) { int v = 0; e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device); if (v > 0) ix->num_cus = v; }
    if (e == hipSuccess) ix->scratch_ptr = (unsigned char *)scratch_of(cfg->device, &e);
    if (e == hipSuccess) e = hipMalloc((void **)&ix->d_short_hash, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ix->d_short_hash, 0, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&ix->d_proj_counters, 48);
    if (e == hipSuccess) e = hipMemset(ix->d_proj_counters, 0, 48);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ix->pinned, 12 * sizeof(int64_t), hipHostMallocDefault);
    for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&ix->ev[i]);
    for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&ix->ev_scan[i]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->side_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_fork0, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_conv, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ix->ev_stage[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        std::string m = std::string("dpf_create: ") + hipGetErrorString(e);
        delete ix;
        return fail(nullptr, DPF_EHIP, m);
    }
    HashParams &hp = ix->hp;
    hp.family = cfg->family; hp.dim = cfg->dim; hp.F = cfg->num_functions; hp.L = cfg->num_tables;
    hp.m = cfg->chain_length; hp.pb = cfg->partition_bits; hp.type_of_index = cfg->type_of_index;
    for (int i = 0; i < 32; i++) hp.sampling_index[i] = i;
    // significantBits.angleDistance / newMethod (significantBits.scala:100-127): the label
    // depends only on the popcount of the low 28 bits.
    static const double metric[9] = {16.0, 25.0, 33.0, 39.0, 46.0, 52.0, 58.0, 66.0, 72.0};
    for (int pc = 0; pc <= 28; pc++) {
        double angle = std::acos((double)pc / (std::sqrt(28.0) * std::sqrt((double)pc)));
        angle = angle * 360 / 2 / 3.141592653589793;
        int idx = 0;
        while (idx < 9 && angle > metric[idx]) idx++;
        hp.angle_label_lut[pc] = idx;
    }
    if (hash_shmem_bytes(hp) > 150 * 1024) { dpf_destroy(ix); return fail(nullptr, DPF_EINVAL, "dpf_create: dim + num_functions too large for the hash kernel's LDS tile"); }
    *out = ix;
    return DPF_OK;
}

extern "C" void dpf_destroy(dpf_index *ix)
{
    if (!ix) return;
    (void)hipSetDevice(ix->cfg.device);
    DevBuf *bufs[] = {&ix->d_fnsT, &ix->d_b, &ix->d_w, &ix->d_chain, &ix->d_part_fns, &ix->d_part_b, &ix->d_part_w,
                      &ix->d_base, &ix->d_roots, &ix->d_dnodes, &ix->d_leaf_off, &ix->d_ids, &ix->ws_qpad,
                      &ix->ws_hashes, &ix->ws_parts, &ix->ws_raw, &ix->ws_bound, &ix->ws_off, &ix->ws_cnt,
                      &ix->ws_arena, &ix->ws_misc, &ix->ws_hq, &ix->ws_hqids, &ix->ws_hids, &ix->ws_hscores, &ix->ws_stage,
                      &ix->d_hashes, &ix->d_parts};
    for (DevBuf *b : bufs) b->release();
    if (ix->d_short_hash) (void)hipFree(ix->d_short_hash);
    if (ix->d_proj_counters) (void)hipFree(ix->d_proj_counters);
    { DevBuf *pb[] = {&ix->d_fn_norm, &ix->ws_proj_codes, &ix->ws_proj_norm, &ix->ws_proj_flag, &ix->ws_proj_rows}; for (DevBuf *b : pb) b->release(); }
    if (ix->pinned) (void)hipHostFree(ix->pinned);
    ix->h_stage.release();
    if (ix->h_sets) (void)hipHostFree(ix->h_sets);
    ix->ws_compact.release();
    for (int i = 0; i < 6; i++) if (ix->ev[i]) (void)hipEventDestroy(ix->ev[i]);
    for (int i = 0; i < 6; i++) if (ix->ev_scan[i]) (void)hipEventDestroy(ix->ev_scan[i]);
    if (ix->ev_fork) (void)hipEventDestroy(ix->ev_fork);
    if (ix->ev_join) (void)hipEventDestroy(ix->ev_join);
    if (ix->ev_fork0) (void)hipEventDestroy(ix->ev_fork0);
    if (ix->ev_conv) (void)hipEventDestroy(ix->ev_conv);
    for (int i = 0; i < 2; i++) if (ix->ev_stage[i]) (void)hipEventDestroy(ix->ev_stage[i]);
    if (ix->side_stream) (void)hipStreamDestroy(ix->side_stream);
    DevBuf *more[] = {&ix->ws_emit_ucnt, &ix->ws_emit_v, &ix->ws_leaf_units, &ix->ws_unit_off, &ix->ws_q_leaves, &ix->ws_q_leaf_cnt, &ix->d_leaf_tile0, &ix->ws_units, &ix->d_tiles, &ix->ws_lq_off, &ix->ws_lq_cursor, &ix->ws_leaf_queries,
                      &ix->ws_seed_cand, &ix->ws_seed_cnt, &ix->ws_seed_ids, &ix->ws_seed_scores, &ix->ws_emit, &ix->ws_emit_cnt,
                      &ix->d_base16, &ix->d_slot16, &ix->d_bin_cap, &ix->d_bin_off, &ix->d_bin_map, &ix->d_bin_lo, &ix->d_gran_cnt, &ix->ws_pairs, &ix->ws_bin_units, &ix->ws_bin_unit_base, &ix->ws_q16, &ix->d_base16n, &ix->ws_q16n, &ix->d_base16e, &ix->ws_q16e, &ix->ws_flag, &ix->ws_tau, &ix->ws_qc, &ix->d_slot_norm, &ix->d_base16t, &ix->d_slot_tail, &ix->ws_q16t, &ix->ws_ovf, &ix->d_base8, &ix->d_norm8, &ix->d_slot_nj8, &ix->d_i8_probe, &ix->ws_q8, &ix->ws_qc8, &ix->d_mu, &ix->d_bounds, &ix->sub_q, &ix->sub_qids, &ix->sub_ids,
                      &ix->sub_scores, &ix->sub_ws[0], &ix->sub_ws[1], &ix->sub_ws[2], &ix->sub_ws[3], &ix->sub_ws[4], &ix->sub_ws[5]};
    for (DevBuf *b : more) b->release();
    delete ix;
}

template <typename T>
static int upload(dpf_index *ix, DevBuf &buf, const std::vector<T> &v)
{
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIPCHECK(ix, buf.ensure(bytes));
    if (!v.empty()) HIPCHECK(ix, hipMemcpy(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return DPF_OK;
}

extern "C" int dpf_set_hash_family(dpf_index *ix, const double *fns, const double *b, const int32_t *w,
                                   const int32_t *chain_fn)
{
    if (!ix) return DPF_EINVAL;
    if (!fns || !chain_fn) return fail(ix, DPF_EINVAL, "dpf_set_hash_family: null fns/chain_fn");
    const dpf_config &c = ix->cfg;
    if (c.family == DPF_FAMILY_PSTABLE && (!b || !w)) return fail(ix, DPF_EINVAL, "dpf_set_hash_family: pStable needs b and w");
    if (ix->n > 0) return fail(ix, DPF_EINVAL, "dpf_set_hash_family: index already holds vectors");
    const size_t F = c.num_functions, d = c.dim, L = c.num_tables, m = c.chain_length;
    for (size_t i = 0; i < L * m; i++)
        if (chain_fn[i] < 0 || chain_fn[i] >= (int32_t)F) return fail(ix, DPF_EINVAL, "dpf_set_hash_family: chain_fn entry outside the pool");
    if (c.family == DPF_FAMILY_PSTABLE)
        for (size_t f = 0; f < F; f++)
            if (w[f] == 0) return fail(ix, DPF_EINVAL, "dpf_set_hash_family: w must be non-zero");
    (void)hipSetDevice(c.device);
    ix->fns.assign(fns, fns + F * d);
    ix->chain_fn.assign(chain_fn, chain_fn + L * m);
    std::vector<double> fnsT(F * d);
    for (size_t f = 0; f < F; f++)
        for (size_t i = 0; i < d; i++) fnsT[i * F + f] = fns[f * d + i];
    int rc;
    if ((rc = upload(ix, ix->d_fnsT, fnsT)) != DPF_OK) return rc;
    if ((rc = upload(ix, ix->d_chain, ix->chain_fn)) != DPF_OK) return rc;
    if (c.family == DPF_FAMILY_PSTABLE) {
        ix->b.assign(b, b + F);
        ix->w.assign(w, w + F);
        if ((rc = upload(ix, ix->d_b, ix->b)) != DPF_OK) return rc;
        if ((rc = upload(ix, ix->d_w, ix->w)) != DPF_OK) return rc;
    }
    {   // Euclidean norm of every pool function: the matrix-core projection's guard band is proportional to it
        std::vector<double> nrm(F);
        for (size_t f = 0; f < F; f++) {
            double s = 0.0;
            for (size_t i = 0; i < d; i++) s += fns[f * d + i] * fns[f * d + i];
            nrm[f] = std::sqrt(s);
        }
        if ((rc = upload(ix, ix->d_fn_norm, nrm)) != DPF_OK) return rc;
    }
    ix->hp.fn_norm = ix->d_fn_norm.as<double>();
    ix->hp.fnsT = ix->d_fnsT.as<double>();
    ix->hp.chain_fn = ix->d_chain.as<int32_t>();
    ix->hp.b = ix->d_b.as<double>();
    ix->hp.w = ix->d_w.as<int32_t>();
    ix->have_family = true;
    return DPF_OK;
}

extern "C" int dpf_set_partitioner(dpf_index *ix, const double *part_fns, const double *part_b, const int32_t *part_w)
{
    if (!ix) return DPF_EINVAL;
    const dpf_config &c = ix->cfg;
    const size_t L = c.num_tables, pb = c.partition_bits;
    if (pb > 0 && !part_fns) return fail(ix, DPF_EINVAL, "dpf_set_partitioner: null part_fns");
    if (pb > 0 && c.family == DPF_FAMILY_PSTABLE && (!part_b || !part_w)) return fail(ix, DPF_EINVAL, "dpf_set_partitioner: pStable needs part_b and part_w");
    if (ix->n > 0) return fail(ix, DPF_EINVAL, "dpf_set_partitioner: index already holds vectors");
    (void)hipSetDevice(c.device);
    int rc;
    ix->part_fns.assign(part_fns, part_fns + L * pb * 32);
    if ((rc = upload(ix, ix->d_part_fns, ix->part_fns)) != DPF_OK) return rc;
    if (c.family == DPF_FAMILY_PSTABLE && pb > 0) {
        for (size_t i = 0; i < L * pb; i++)
            if (part_w[i] == 0) return fail(ix, DPF_EINVAL, "dpf_set_partitioner: part_w must be non-zero");
        ix->part_b.assign(part_b, part_b + L * pb);
        ix->part_w.assign(part_w, part_w + L * pb);
        if ((rc = upload(ix, ix->d_part_b, ix->part_b)) != DPF_OK) return rc;
        if ((rc = upload(ix, ix->d_part_w, ix->part_w)) != DPF_OK) return rc;
    }
    ix->hp.part_fns = ix->d_part_fns.as<double>();
    ix->hp.part_b = ix->d_part_b.as<double>();
    ix->hp.part_w = ix->d_part_w.as<int32_t>();
    ix->have_part = true;
    return DPF_OK;
}

extern "C" int dpf_set_sampling_index(dpf_index *ix, const int32_t *si)
{
    if (!ix) return DPF_EINVAL;
    if (!si) return fail(ix, DPF_EINVAL, "dpf_set_sampling_index: null");
    bool seen[32] = {false};
    for (int i = 0; i < 32; i++) {
        if (si[i] < 0 || si[i] > 31 || seen[si[i]]) return fail(ix, DPF_EINVAL, "dpf_set_sampling_index: not a permutation of 0..31");
        seen[si[i]] = true;
    }
    if (ix->n > 0) return fail(ix, DPF_EINVAL, "dpf_set_sampling_index: index already holds vectors");
    for (int i = 0; i < 32; i++) ix->hp.sampling_index[i] = si[i];
    return DPF_OK;
}

static int check_ready(dpf_index *ix, bool need_fit)
{
    if (!ix->have_family) return fail(ix, DPF_ENOFAMILY, "hash family not set (dpf_set_hash_family)");
    if (ix->cfg.partition_bits > 0 && !ix->have_part) return fail(ix, DPF_ENOFAMILY, "partitioner not set (dpf_set_partitioner)");
    if (need_fit && (ix->cfg.flags & DPF_FLAG_HASH_ONLY)) return fail(ix, DPF_EINVAL, "this handle was created hash-only");
    if (need_fit && ix->build_pending) return fail(ix, DPF_ENOTFIT, "rows were appended with DPF_OPT_DEFER_BUILD set: clear it and call dpf_fit (n = 0 will do) to build the forest");
    if (need_fit && !ix->fitted) return fail(ix, DPF_ENOTFIT, "need to fit the data first");
    return DPF_OK;
}

// Hashes and sub-index ids of n vectors (device rows, `stride` elements apart): large batches go through the
// matrix-core projection (same results bit for bit), small ones and callers that may not allocate through the
// exact vector-ALU kernel alone.
static int hash_rows(dpf_index *ix, const void *X, int64_t stride, int64_t n, int32_t *hashes, int32_t *parts, hipStream_t stream,
                     bool may_alloc = true)
{
    if (n <= 0) return DPF_OK;
    // the contraction pays from a few thousand vectors on and when a vector meets enough work (dim x functions);
    // below that the exact kernel alone is as fast and takes one launch instead of four
    const bool forced = ix->opt_projection_min_rows == 1;
    const bool worth = n >= ix->opt_projection_min_rows && (int64_t)ix->cfg.dim * ix->cfg.num_functions >= 40000;
    if (!may_alloc || ix->opt_projection_min_rows <= 0 || !(forced || worth)) {
        launch_hash(ix->cfg.dtype, X, stride, n, ix->hp, hashes, parts, stream);
        return DPF_OK;
    }
    HIPCHECK(ix, ix->ws_proj_codes.ensure(std::max<size_t>(project_codes_bytes(ix->hp, n), 16)));
    HIPCHECK(ix, ix->ws_proj_norm.ensure((size_t)n * 8));
    HIPCHECK(ix, ix->ws_proj_flag.ensure((size_t)n * 4));
    HIPCHECK(ix, ix->ws_proj_rows.ensure((size_t)n * 4));
    ProjectWs ws;
    ws.codes = ix->ws_proj_codes.p;
    ws.row_norm = ix->ws_proj_norm.as<double>();
    ws.row_flag = ix->ws_proj_flag.as<uint32_t>();
    ws.flag_rows = ix->ws_proj_rows.as<int32_t>();
    ws.flag_count = (uint32_t *)ix->d_proj_counters;
    ws.flag_total = (unsigned long long *)(ix->d_proj_counters + 8);
    HIPCHECK(ix, launch_hash_mfma(ix->cfg.dtype, X, stride, n, ix->hp, ws, hashes, parts, stream));
    return DPF_OK;
}

// The seed sample a handle draws per query when none is pinned.  The first thresholds come from the k-th best of the
// sample, which is taken from the query's own buckets (one per probed table); how well it stands for them depends on
// how many rows those buckets hold.  With m = the size-biased mean bucket size (the expected size of the bucket a
// stored row - or a query near one - falls into) the own buckets hold about T * m rows over the T tables this handle
// probes, and the sample is 5 sqrt(T m), at least 128 and at most 2048, a multiple of 64.  Measured optima this rule
// has to meet (batch time against sample size): one tree of 1M clustered 100-d rows (T m = 172): 128; 30 trees of
// 100-d rows (5.9 k): 256-384; 30 trees of SIFT-shaped rows (58 k, buckets of up to 31 k rows): 1024-2048, where 256 is
// 18 % slower.  Rows of 16 steps and more (512 dimensions up) take 1.5 sqrt(T m), at least 64: a seed costs as many bytes
// as a scanned row there, and their three scan slices with a refinement after each make up for weaker first thresholds
// (30 trees of 768-d rows, T m = 3.4 k: 96 / 128 / 192 / 256 / 384 / 512 -> 15.99 / 16.04 / 16.22 / 16.44 / 16.78 / 17.16 ms).
static int32_t seed_sample_for(const dpf_index *ix)
{
    double s1 = 0, s2 = 0;
    for (uint32_t len : ix->h_leaf_len) { s1 += (double)len; s2 += (double)len * (double)len; }
    if (s1 <= 0) return 256;
    const int t_end = ix->cfg.table_end > 0 ? ix->cfg.table_end : ix->cfg.num_tables;
    const int tables = std::max(1, t_end - ix->cfg.table_begin);
    const bool long_rows = ix->ks16 >= 16;
    const double want = (long_rows ? 1.5 : 5.0) * std::sqrt((double)tables * (s2 / s1));
    const int64_t v = ((int64_t)want + 63) / 64 * 64;
    return (int32_t)std::min<int64_t>(2048, std::max<int64_t>(long_rows ? 64 : 128, v));
}

// A second copy of the fp16 rows in bucket order (one row per bucket entry: num_tables times the table), so that the
// scan reads a tile's rows as one run instead of gathering them by id.  Gathered by id, a row that is no whole number
// of 128-byte lines (192 bytes at 100 dimensions) makes the fabric move the lines it shares with rows nobody asked for:
// 1.36x the bytes the scan needs on 1M x 100-d rows, 1.10x from the bucket-order copy.  Whether that pays depends on
// where the rows come from: a 192 MB table that stays in the 256 MiB Infinity Cache is re-read from there once per
// tree whatever the request shape (scan launch 0.91 ms by id, 0.97 from the copy, which streams 5.8 GB from HBM
// instead); tables beyond the cache come from HBM either way and the copy wins - 2M rows: 1.62 -> 1.59 ms per launch,
// 3M: 2.41 -> 2.21, 4M: 3.16 -> 2.81.  So, unless a flag says otherwise: rows that straddle lines, a table of more
// than 384 MiB, and room for the copy.
static int make_slot_order_rows(dpf_index *ix, int64_t num_ids)
{
    ix->slot16_ok = false;
    const int32_t fl = ix->cfg.flags;
    if ((fl & DPF_FLAG_NO_BUCKET_ORDER_ROWS) || num_ids <= 0 || ix->ks16 <= 0 || ix->ks16 > 8) return DPF_OK;
    const size_t bytes = (size_t)num_ids * ix->ks16 * 64;
    if (!(fl & DPF_FLAG_BUCKET_ORDER_ROWS)) {
        const size_t table = (size_t)ix->base_rows * ix->ks16 * 64;
        if (ix->ks16 % 2 == 0 || table <= ((size_t)384 << 20) || bytes > ((size_t)48 << 30)) { ix->d_slot16.release(); return DPF_OK; }
    }
    size_t free_b = 0, total_b = 0;
    HIPCHECK(ix, hipMemGetInfo(&free_b, &total_b));
    if (bytes > ix->d_slot16.cap && bytes + ((size_t)4 << 30) > free_b + ix->d_slot16.cap) { ix->d_slot16.release(); return DPF_OK; }   // no room: the scan gathers by id
    HIPCHECK(ix, ix->d_slot16.ensure(bytes));
    launch_rows_to_slot_order(ix->d_ids.as<int32_t>(), num_ids, ix->d_base16.as<uint16_t>(), ix->ks16 * 32, ix->row_begin,
                              ix->row_begin + ix->base_rows, ix->d_slot16.as<uint16_t>(), nullptr);
    HIPCHECK(ix, hipGetLastError());
    ix->slot16_ok = true;
    return DPF_OK;
}

// granules of buckets for the binned list building: the smallest power of two (16 or more) that gives at most
// BIN_MAX_GRANULES of them; the bins themselves are cut from a batch's counts (k_bin_seed)
static void setup_bins(dpf_index *ix)
{
    ix->bins_seeded = false;
    ix->gran_shift = 0; ix->ngran = 0; ix->nbins = 0;
    const int64_t nl = ix->num_leaves;
    if (nl <= 0 || nl >= ((int64_t)1 << BIN_LEAF_BITS)) return;   // more buckets than that: the lists are built the other way
    int shift = 4;
    while (((nl + ((int64_t)1 << shift) - 1) >> shift) > BIN_MAX_GRANULES) shift++;
    ix->gran_shift = shift;
    ix->ngran = (int32_t)((nl + ((int64_t)1 << shift) - 1) >> shift);
}

static int upload_snapshot(dpf_index *ix, ForestSnapshot &s)
{
    int rc;
    // a row-sharded handle keeps only its own ids in the buckets: the trie is replicated,
    // the bucket contents are not
    const int64_t lo = ix->row_begin, hi = std::min<int64_t>(ix->row_end, ix->n);
    if (lo > 0 || hi < ix->n) {
        std::vector<int32_t> ids;
        std::vector<uint32_t> off(s.leaf_off.size(), 0u);
        ids.reserve(s.ids.size() / 2 + 16);
        for (size_t l = 0; l + 1 < s.leaf_off.size(); l++) {
            for (uint32_t k = s.leaf_off[l]; k < s.leaf_off[l + 1]; k++)
                if (s.ids[k] >= lo && s.ids[k] < hi) ids.push_back(s.ids[k]);
            off[l + 1] = (uint32_t)ids.size();
        }
        s.ids.swap(ids);
        s.leaf_off.swap(off);
    }
    std::vector<uint32_t> tiles, leaf_tile0(s.leaf_off.size(), 0u);
    const uint32_t TR = (uint32_t)leafscan_tile_rows(ix->ks16);   // rows per work-unit tile of the scan
    for (size_t l = 0; l + 1 < s.leaf_off.size(); l++) {
        const uint32_t len = s.leaf_off[l + 1] - s.leaf_off[l];
        leaf_tile0[l] = (uint32_t)(tiles.size() / 4);
        for (uint32_t r = 0; r < len; r += TR) {          // work-unit tiles of the scan
            tiles.push_back((uint32_t)l); tiles.push_back(s.leaf_off[l] + r); tiles.push_back(std::min<uint32_t>(TR, len - r)); tiles.push_back(0u);
        }
    }
    leaf_tile0.back() = (uint32_t)(tiles.size() / 4);
    ix->nutiles = (int64_t)(tiles.size() / 4);
    if ((rc = upload(ix, ix->d_tiles, tiles)) != DPF_OK) return rc;
    if ((rc = upload(ix, ix->d_leaf_tile0, leaf_tile0)) != DPF_OK) return rc;
    ix->h_leaf_len.resize(s.leaf_off.size() - 1);
    for (size_t l = 0; l + 1 < s.leaf_off.size(); l++) ix->h_leaf_len[l] = s.leaf_off[l + 1] - s.leaf_off[l];
    ix->auto_seed_cap = seed_sample_for(ix);
    ix->ntiles = ix->nutiles;
    ix->num_leaves = (int64_t)s.leaf_off.size() - 1;
    setup_bins(ix);
    if ((rc = upload(ix, ix->d_roots, s.roots)) != DPF_OK) return rc;
    if ((rc = upload(ix, ix->d_dnodes, s.dnodes)) != DPF_OK) return rc;
    if ((rc = upload(ix, ix->d_leaf_off, s.leaf_off)) != DPF_OK) return rc;
    if ((rc = upload(ix, ix->d_ids, s.ids)) != DPF_OK) return rc;
    ix->st.num_dnodes = (int64_t)(s.dnodes.size() / ix->geo.dir_node_size) - 1;
    ix->st.num_leaves = (int64_t)s.leaf_off.size() - 1;
    ix->st.max_leaf = s.max_leaf;
    ix->st.misflag_events = s.misflag_events;
    ix->st.snapshot_bytes = (int64_t)(s.roots.size() * 4 + s.dnodes.size() * 4 + s.leaf_off.size() * 4 + s.ids.size() * 4);
    ix->part_counts = s.part_counts;
    if (ix->bucket_major && ix->ks16 > 0 && ix->ntiles > 0 && ix->base_rows > 0) {
        HIPCHECK(ix, ix->d_slot_norm.ensure(std::max<size_t>(ix->d_ids.cap, 16)));
        const bool tails = ix->filter_dim < ix->cfg.dim;
        if (tails) HIPCHECK(ix, ix->d_slot_tail.ensure(std::max<size_t>(ix->d_ids.cap, 16) * 4));
        HIPCHECK(ix, ix->d_bounds.ensure(16));
        HIPCHECK(ix, hipMemset(ix->d_bounds.p, 0, 16));
        launch_tile_maxnorm((uint4 *)ix->d_tiles.p, ix->nutiles, ix->d_ids.as<int32_t>(), ix->d_base16n.as<float>(), ix->d_base16e.as<float>(),
                            ix->row_begin, ix->row_begin + ix->base_rows, ix->d_slot_norm.as<float>(), nullptr,
                            tails ? (const uint4 *)ix->d_base16t.p : nullptr, tails ? (uint4 *)ix->d_slot_tail.p : nullptr,
                            ix->d_bounds.as<uint32_t>());
        HIPCHECK(ix, hipMemcpy(ix->tile_bounds, ix->d_bounds.p, 12, hipMemcpyDeviceToHost));
        if (ix->i8_ok) {
            HIPCHECK(ix, ix->d_slot_nj8.ensure(std::max<size_t>(s.ids.size() * 4, 16)));
            launch_slot_nj8(ix->d_ids.as<int32_t>(), (int64_t)s.ids.size(), ix->d_norm8.as<int32_t>(), ix->row_begin,
                            ix->row_begin + ix->base_rows, ix->d_slot_nj8.as<int32_t>(), nullptr);
        }
        if ((rc = make_slot_order_rows(ix, (int64_t)s.ids.size())) != DPF_OK) return rc;
        HIPCHECK(ix, hipDeviceSynchronize());
    }
    return DPF_OK;
}


static int hashes_to_host(dpf_index *ix)
{
    if (ix->host_hashes_valid) return DPF_OK;
    const size_t cnt = (size_t)ix->n * ix->cfg.num_tables;
    ix->h_hashes.resize(cnt);
    ix->h_parts.resize(cnt);
    if (cnt > 0) {
        HIPCHECK(ix, hipMemcpy(ix->h_hashes.data(), ix->d_hashes.p, cnt * 4, hipMemcpyDeviceToHost));
        HIPCHECK(ix, hipMemcpy(ix->h_parts.data(), ix->d_parts.p, cnt * 4, hipMemcpyDeviceToHost));
    }
    ix->host_hashes_valid = true;
    return DPF_OK;
}

// The forest derived on the device from d_hashes / d_parts (dpf_build.hip); same arrays as
// build_forest + upload_snapshot produce.
static int device_snapshot(dpf_index *ix, int64_t total)
{
    DeviceForest f;
    std::string err;
    int rc = build_forest_device(ix->geo, ix->d_hashes.as<int32_t>(), ix->d_parts.as<int32_t>(), total, f, err);
    if (rc != 0) { f.release(); return fail(ix, rc == -1 ? DPF_ECAPACITY : DPF_EHIP, err); }
    const int64_t lo = ix->row_begin, hi = std::min<int64_t>(ix->row_end, total);
    if ((lo > 0 || hi < total) && filter_forest_rows(f, lo, hi, err) != 0) { f.release(); return fail(ix, DPF_EHIP, err); }
    uint32_t *lt0 = nullptr;
    uint4 *tiles = nullptr;
    int64_t ntiles = 0;
    if (make_forest_tiles(f, leafscan_tile_rows(ix->ks16), &lt0, &tiles, &ntiles, err) != 0) {
        if (lt0) (void)hipFree(lt0);
        if (tiles) (void)hipFree(tiles);
        f.release();
        return fail(ix, DPF_EHIP, err);
    }
    const size_t PS = (size_t)ix->geo.num_parts * ix->geo.seg;
    ix->d_roots.adopt(f.roots, std::max<size_t>(PS * ix->cfg.num_tables * 4, 16));
    ix->d_dnodes.adopt(f.dnodes, (size_t)(f.num_dnodes + 1) * ix->geo.dir_node_size * 4);
    ix->d_leaf_off.adopt(f.leaf_off, ((size_t)f.num_leaves + 1) * 4);
    ix->d_ids.adopt(f.ids, std::max<size_t>((size_t)f.num_ids * 4, 16));
    const int64_t num_ids = f.num_ids;
    f.roots = f.dnodes = f.leaf_off = nullptr;
    f.ids = nullptr;
    ix->d_leaf_tile0.adopt(lt0, ((size_t)f.num_leaves + 1) * 4 + 16);
    ix->d_tiles.adopt(tiles, std::max<size_t>((size_t)ntiles * 16, 16));
    ix->nutiles = ix->ntiles = ntiles;
    ix->num_leaves = f.num_leaves;
    setup_bins(ix);
    std::vector<uint32_t> off((size_t)f.num_leaves + 1, 0u);
    HIPCHECK(ix, hipMemcpy(off.data(), ix->d_leaf_off.p, off.size() * 4, hipMemcpyDeviceToHost));
    ix->h_leaf_len.resize((size_t)f.num_leaves);
    for (size_t l = 0; l < ix->h_leaf_len.size(); l++) ix->h_leaf_len[l] = off[l + 1] - off[l];
    ix->auto_seed_cap = seed_sample_for(ix);
    ix->st.num_dnodes = f.num_dnodes;
    ix->st.num_leaves = f.num_leaves;
    ix->st.max_leaf = f.max_leaf;
    ix->st.misflag_events = f.misflag_events;
    ix->st.snapshot_bytes = (int64_t)(PS * ix->cfg.num_tables * 4 + (size_t)(f.num_dnodes + 1) * ix->geo.dir_node_size * 4 +
                                      ((size_t)f.num_leaves + 1) * 4 + (size_t)f.num_ids * 4);
    ix->part_counts = f.part_counts;
    if (ix->bucket_major && ix->ks16 > 0 && ix->ntiles > 0 && ix->base_rows > 0) {
        HIPCHECK(ix, ix->d_slot_norm.ensure(std::max<size_t>(ix->d_ids.cap, 16)));
        const bool tails = ix->filter_dim < ix->cfg.dim;
        if (tails) HIPCHECK(ix, ix->d_slot_tail.ensure(std::max<size_t>(ix->d_ids.cap, 16) * 4));
        HIPCHECK(ix, ix->d_bounds.ensure(16));
        HIPCHECK(ix, hipMemset(ix->d_bounds.p, 0, 16));
        launch_tile_maxnorm((uint4 *)ix->d_tiles.p, ix->nutiles, ix->d_ids.as<int32_t>(), ix->d_base16n.as<float>(), ix->d_base16e.as<float>(),
                            ix->row_begin, ix->row_begin + ix->base_rows, ix->d_slot_norm.as<float>(), nullptr,
                            tails ? (const uint4 *)ix->d_base16t.p : nullptr, tails ? (uint4 *)ix->d_slot_tail.p : nullptr,
                            ix->d_bounds.as<uint32_t>());
        HIPCHECK(ix, hipMemcpy(ix->tile_bounds, ix->d_bounds.p, 12, hipMemcpyDeviceToHost));
        if (ix->i8_ok) {
            HIPCHECK(ix, ix->d_slot_nj8.ensure(std::max<size_t>((size_t)num_ids * 4, 16)));
            launch_slot_nj8(ix->d_ids.as<int32_t>(), num_ids, ix->d_norm8.as<int32_t>(), ix->row_begin,
                            ix->row_begin + ix->base_rows, ix->d_slot_nj8.as<int32_t>(), nullptr);
        }
        { const int brc = make_slot_order_rows(ix, num_ids); if (brc != DPF_OK) return brc; }
        HIPCHECK(ix, hipDeviceSynchronize());
    }
    return DPF_OK;
}

// Second half of a fit, shared with dpf_load: the resident rows are in d_base and the hashes of all
// `total` ids in d_hashes / d_parts; make the fp16 copy and build the forest.
static int finish_fit(dpf_index *ix, int64_t total, int64_t new_rows)
{
    int rc;
    const int d = ix->cfg.dim;
    const size_t row_bytes = (size_t)ix->dpad * ix->elem;
    // The int8 copy, when the resident rows allow it: every value an integer, all of them inside one window of 256
    // values (SIFT's 0..255, for one); the window's centre is subtracted, which leaves distances as they are.
    ix->i8_ok = false;
    bool i8_window = false;
    if (ix->bucket_major && ix->ks16 > 0 && new_rows > 0 && ix->cfg.metric == DPF_METRIC_L2 && d <= 256 && !(ix->cfg.flags & DPF_FLAG_NO_INT8)) {
        const int32_t init[4] = {INT32_MAX, INT32_MIN, 0, 0};
        int32_t got[4];
        HIPCHECK(ix, ix->d_i8_probe.ensure(16));
        HIPCHECK(ix, hipMemcpy(ix->d_i8_probe.p, init, 16, hipMemcpyHostToDevice));
        launch_i8_range(ix->d_base.as<float>(), ix->dpad, d, new_rows, ix->d_i8_probe.as<int32_t>(), nullptr);
        HIPCHECK(ix, hipMemcpy(got, ix->d_i8_probe.p, 16, hipMemcpyDeviceToHost));
        auto dec = [](int32_t b2) { b2 = b2 >= 0 ? b2 : (b2 ^ 0x7fffffff); float f; memcpy(&f, &b2, 4); return f; };
        const float lo = dec(got[0]), hi = dec(got[1]);
        // (within +-2048 every integer is an fp16 value too: the thresholds the two forms of the scan share - seed radii,
        // refinements - then carry no rounding allowance for such a batch)
        if (got[2] == 0 && std::isfinite(lo) && std::isfinite(hi) && hi - lo <= 255.0f && std::fabs(lo) <= 2048.0f && std::fabs(hi) <= 2048.0f) {
            i8_window = true;
            ix->ks8 = (d + 63) / 64;
            // The window [c - 128, c + 127]: the value range of unsigned or signed bytes when the rows lie inside one of
            // them (queries of the same kind then always fit), else around the middle of the rows' range, so that queries
            // a little outside that range on either side still fit.
            if (lo >= 0.0f && hi <= 255.0f) ix->i8_offset = 128.0f;
            else if (lo >= -128.0f && hi <= 127.0f) ix->i8_offset = 0.0f;
            else ix->i8_offset = std::min(std::max(std::floor((lo + hi + 1.0f) * 0.5f), hi - 127.0f), lo + 128.0f);
        }
    }
    // l2 handles take their fp16 copies around the mean row: distances do not change, and the rounding distances of
    // the copies shrink with the values (rows of 1000 +- 1 would otherwise round to steps of 0.5).  Not where the
    // int8 copy exists: its rows are exact in fp16 as they are.
    ix->centred = false;
    if (ix->bucket_major && ix->ks16 > 0 && new_rows > 0 && ix->cfg.metric == DPF_METRIC_L2 && !i8_window && !(ix->cfg.flags & DPF_FLAG_NO_CENTRE)) {
        HIPCHECK(ix, ix->d_mu.ensure((size_t)ix->dpad * 4));
        HIPCHECK(ix, hipMemset(ix->d_mu.p, 0, (size_t)ix->dpad * 4));
        launch_col_sums(ix->d_base.as<float>(), ix->dpad, d, new_rows, ix->d_mu.as<float>(), nullptr);
        std::vector<float> mu((size_t)ix->dpad, 0.f);
        HIPCHECK(ix, hipMemcpy(mu.data(), ix->d_mu.p, (size_t)d * 4, hipMemcpyDeviceToHost));
        bool any = false;
        for (int c = 0; c < d; c++) {
            float m = mu[c] / (float)new_rows;
            if (!std::isfinite(m) || std::fabs(m) > 1.0e30f) m = 0.f;
            mu[c] = m;
            any = any || m != 0.f;
        }
        HIPCHECK(ix, hipMemcpy(ix->d_mu.p, mu.data(), (size_t)ix->dpad * 4, hipMemcpyHostToDevice));
        ix->centred = any;
    }
    if (ix->bucket_major && ix->ks16 > 0 && new_rows > 0) {
        HIPCHECK(ix, ix->d_base16.ensure((size_t)new_rows * ix->ks16 * 32 * 2));
        HIPCHECK(ix, ix->d_base16n.ensure((size_t)new_rows * 4));
        HIPCHECK(ix, ix->d_base16e.ensure((size_t)new_rows * 4));
        const bool tails = ix->filter_dim < d;
        if (tails) HIPCHECK(ix, ix->d_base16t.ensure((size_t)new_rows * 16));
        launch_rows_to_f16(ix->d_base.as<float>(), ix->dpad, std::min(d, ix->filter_dim), new_rows, ix->d_base16.as<uint16_t>(), ix->ks16 * 32,
                            ix->d_base16n.as<float>(), ix->d_base16e.as<float>(), nullptr, nullptr, d, tails ? (uint4 *)ix->d_base16t.p : nullptr,
                            ix->centred ? ix->d_mu.as<float>() : nullptr);
    }
    if (i8_window) {
        int32_t got[4];
        HIPCHECK(ix, ix->d_base8.ensure((size_t)new_rows * ix->ks8 * 64));
        HIPCHECK(ix, ix->d_norm8.ensure((size_t)new_rows * 4));
        launch_rows_to_i8(ix->d_base.as<float>(), ix->dpad, d, new_rows, ix->i8_offset, ix->d_base8.as<int8_t>(), ix->ks8 * 64,
                          ix->d_norm8.as<int32_t>(), nullptr, nullptr, ix->d_i8_probe.as<int32_t>() + 3, nullptr);
        HIPCHECK(ix, hipMemcpy(got, ix->d_i8_probe.p, 16, hipMemcpyDeviceToHost));
        ix->i8_ok = got[3] == 0;
    }
    HIPCHECK(ix, hipDeviceSynchronize());
    ix->base_rows = new_rows;
    ix->n = total;
    ix->st.num_vectors = total;
    ix->st.base_bytes = (int64_t)((size_t)new_rows * row_bytes);

    const auto t0 = std::chrono::steady_clock::now();
    if (ix->cfg.flags & DPF_FLAG_HOST_FOREST) {
        if ((rc = hashes_to_host(ix)) != DPF_OK) return rc;
        ForestSnapshot snap;
        int hw = (int)std::thread::hardware_concurrency();
        const char *env = getenv("OMP_NUM_THREADS");
        int threads = env ? atoi(env) : std::min(hw > 0 ? hw : 8, 16);
        if (threads < 1) threads = 1;
        if (build_forest(ix->geo, ix->h_hashes.data(), ix->h_parts.data(), total, threads, snap) != 0)
            return fail(ix, DPF_ECAPACITY, "dpf_fit: snapshot exceeds 32-bit arena indexing");
        if ((rc = upload_snapshot(ix, snap)) != DPF_OK) return rc;
    } else if ((rc = device_snapshot(ix, total)) != DPF_OK) {
        return rc;
    }
    ix->st.fit_ms_forest = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ix->fitted = total > 0;
    ix->last_valid = false;
    return DPF_OK;
}

extern "C" int dpf_fit(dpf_index *ix, const void *X, int64_t n)
{
    if (!ix) return DPF_EINVAL;
    int rc = check_ready(ix, false);
    if (rc != DPF_OK) return rc;
    if (n < 0 || (n > 0 && !X)) return fail(ix, DPF_EINVAL, "dpf_fit: bad arguments");
    if (ix->cfg.flags & DPF_FLAG_HASH_ONLY) return fail(ix, DPF_EINVAL, "dpf_fit: this handle was created hash-only");
    ix->sets_valid = false;
    const dpf_config &c = ix->cfg;
    const int L = c.num_tables, d = c.dim;
    const int64_t first = ix->n, total = first + n;
    if (total * L >= (int64_t)0x7fffffff) return fail(ix, DPF_ECAPACITY, "dpf_fit: more than 2^31 (id, table) pairs");
    HIPCHECK(ix, hipSetDevice(c.device));

    // grow the resident base shard
    const int64_t res_begin = ix->row_begin;
    const int64_t res_end = std::min<int64_t>(ix->row_end, total);
    const int64_t new_rows = std::max<int64_t>(0, res_end - res_begin);
    const size_t row_bytes = (size_t)ix->dpad * ix->elem;
    // (a caller that streams the base in announces the final size, so that the buffers are not regrown chunk by chunk)
    const int64_t final_total = std::max<int64_t>(total, ix->opt_expected_rows);
    if (final_total * L >= (int64_t)0x7fffffff) return fail(ix, DPF_ECAPACITY, "dpf_fit: more than 2^31 (id, table) pairs expected");
    const int64_t final_rows = std::max<int64_t>(new_rows, std::min<int64_t>(ix->row_end, final_total) - res_begin);
    if (new_rows > ix->base_rows) {
        size_t need = std::max<size_t>((size_t)final_rows * row_bytes, 16);
        if (need > ix->d_base.cap) {
            DevBuf nb;
            HIPCHECK(ix, nb.ensure(need));
            if (ix->base_rows > 0) HIPCHECK(ix, hipMemcpy(nb.p, ix->d_base.p, (size_t)ix->base_rows * row_bytes, hipMemcpyDeviceToDevice));
            ix->d_base.release();
            ix->d_base = nb;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHECK(ix, ix->d_hashes.grow((size_t)first * L * 4, std::max<size_t>((size_t)final_total * L * 4, 16)));
    HIPCHECK(ix, ix->d_parts.grow((size_t)first * L * 4, std::max<size_t>((size_t)final_total * L * 4, 16)));
    if (n > 0) ix->host_hashes_valid = false;
    const int64_t batch = 1 << 17;
    HIPCHECK(ix, ix->ws_stage.ensure((size_t)std::min<int64_t>(batch, std::max<int64_t>(n, 1)) * d * ix->elem));
    for (int64_t s = 0; s < n; s += batch) {
        const int64_t cnt = std::min(batch, n - s);
        const char *src = (const char *)X + (size_t)s * d * ix->elem;
        HIPCHECK(ix, hipMemcpy(ix->ws_stage.p, src, (size_t)cnt * d * ix->elem, hipMemcpyHostToDevice));
        const int64_t g0 = first + s, g1 = g0 + cnt;
        if ((rc = hash_rows(ix, ix->ws_stage.p, d, cnt, ix->d_hashes.as<int32_t>() + (size_t)g0 * L,
                            ix->d_parts.as<int32_t>() + (size_t)g0 * L, nullptr)) != DPF_OK) return rc;
        // rows of this batch that belong to the resident shard
        const int64_t a = std::max(g0, res_begin), bnd = std::min(g1, res_end);
        if (bnd > a) {
            const char *rsrc = (const char *)ix->ws_stage.p + (size_t)(a - g0) * d * ix->elem;
            char *rdst = (char *)ix->d_base.p + (size_t)(a - res_begin) * row_bytes;
            launch_copy_rows(c.dtype, rsrc, bnd - a, d, rdst, ix->dpad, nullptr);
        }
        HIPCHECK(ix, hipDeviceSynchronize());     // the staging buffer is reused by the next batch
    }
    ix->st.fit_ms_hash = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ix->opt_defer_build) {
        // streamed fit: the rows and hashes are in place; the copies the scan works on and the forest wait for the
        // last chunk (the first dpf_fit after DPF_OPT_DEFER_BUILD is cleared, which may carry no rows)
        ix->base_rows = new_rows;
        ix->n = total;
        ix->st.num_vectors = total;
        ix->st.base_bytes = (int64_t)((size_t)new_rows * row_bytes);
        ix->fitted = false;
        ix->build_pending = true;
        ix->last_valid = false;
        ix->st.fit_ms_total = ix->st.fit_ms_hash;
        return DPF_OK;
    }
    ix->build_pending = false;
    rc = finish_fit(ix, total, new_rows);
    ix->st.fit_ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// ---------------------------------------------------------------------------
// dpf_save / dpf_load: one file holds what a fitted handle cannot recompute
// without the caller's data: the configuration, the hash family and the
// partitioner, the hashes of every id and the resident rows.  The forest
// itself is rebuilt on load (it is a pure function of the hashes).
// ---------------------------------------------------------------------------
namespace {
const char SNAP_MAGIC[8] = {'D', 'P', 'F', 'S', 'N', 'A', 'P', '2'};

struct FileW {
    FILE *f; bool ok = true;
    void put(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    template <typename T> void val(const T &v) { put(&v, sizeof(T)); }
    template <typename T> void vec(const std::vector<T> &v) { const uint64_t n = v.size(); val(n); put(v.data(), n * sizeof(T)); }
};
struct FileR {
    FILE *f; bool ok = true;
    void get(void *p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; }
    template <typename T> void val(T &v) { get(&v, sizeof(T)); }
    template <typename T> void vec(std::vector<T> &v, uint64_t most)
    {
        uint64_t n = 0; val(n);
        if (!ok || n > most) { ok = false; return; }
        v.resize((size_t)n); get(v.data(), (size_t)n * sizeof(T));
    }
};
}  // namespace

extern "C" int dpf_get_config(const dpf_index *ix, dpf_config *out)
{
    if (!ix || !out) return DPF_EINVAL;
    *out = ix->cfg;
    return DPF_OK;
}

extern "C" int dpf_save(dpf_index *ix, const char *path)
{
    if (!ix || !path) return DPF_EINVAL;
    int rc = check_ready(ix, true);
    if (rc != DPF_OK) return rc;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    if ((rc = hashes_to_host(ix)) != DPF_OK) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ix, DPF_EINVAL, std::string("dpf_save: cannot open ") + path);
    FileW w{f};
    w.put(SNAP_MAGIC, 8);
    w.val(ix->cfg);
    w.put(ix->hp.sampling_index, sizeof(ix->hp.sampling_index));
    w.val(ix->n); w.val(ix->base_rows);
    w.vec(ix->fns); w.vec(ix->chain_fn); w.vec(ix->b); w.vec(ix->w);
    w.vec(ix->part_fns); w.vec(ix->part_b); w.vec(ix->part_w);
    w.vec(ix->h_hashes); w.vec(ix->h_parts);
    // resident rows, padding stripped
    const size_t row_bytes = (size_t)ix->dpad * ix->elem, out_row = (size_t)ix->cfg.dim * ix->elem;
    const int64_t batch = 1 << 16;
    std::vector<char> pad((size_t)std::min<int64_t>(batch, std::max<int64_t>(ix->base_rows, 1)) * row_bytes), flat;
    for (int64_t s0 = 0; s0 < ix->base_rows && w.ok; s0 += batch) {
        const int64_t cnt = std::min(batch, ix->base_rows - s0);
        hipError_t e = hipMemcpy(pad.data(), (const char *)ix->d_base.p + (size_t)s0 * row_bytes, (size_t)cnt * row_bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { fclose(f); ix->err = std::string("dpf_save: ") + hipGetErrorString(e); return DPF_EHIP; }
        flat.resize((size_t)cnt * out_row);
        for (int64_t i = 0; i < cnt; i++) memcpy(flat.data() + (size_t)i * out_row, pad.data() + (size_t)i * row_bytes, out_row);
        w.put(flat.data(), flat.size());
    }
    w.put(SNAP_MAGIC, 8);                                 // trailer: a truncated file does not load
    const bool ok = w.ok && fclose(f) == 0;
    if (!ok) return fail(ix, DPF_EINVAL, std::string("dpf_save: write failed on ") + path);
    return DPF_OK;
}

namespace {
// a read-only mapping of a snapshot file, walked with bounds checks
struct MapR {
    const unsigned char *base = nullptr;
    size_t size = 0, at = 0;
    bool ok = true;
    const void *take(size_t n)
    {
        if (!ok || n > size - at) { ok = false; return nullptr; }
        const void *p = base + at;
        at += n;
        return p;
    }
    template <typename T> void val(T &v) { const void *p = take(sizeof(T)); if (p) memcpy(&v, p, sizeof(T)); }
    // a length-prefixed array: pointer into the mapping + element count
    template <typename T> const T *vec(uint64_t &n)
    {
        n = 0;
        val(n);
        if (!ok || n > (size - at) / sizeof(T)) { ok = false; return nullptr; }
        return (const T *)take((size_t)n * sizeof(T));
    }
};
}  // namespace

extern "C" int dpf_load_shard(const char *path, int32_t device, const dpf_config *shard, dpf_index **out)
{
    if (!path || !out) return DPF_EINVAL;
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(nullptr, DPF_EINVAL, std::string("dpf_load: cannot open ") + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 16) { close(fd); return fail(nullptr, DPF_EINVAL, "dpf_load: not a snapshot file of this library version"); }
    void *mapping = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (mapping == MAP_FAILED) return fail(nullptr, DPF_EINVAL, std::string("dpf_load: cannot map ") + path);
    MapR r;
    r.base = (const unsigned char *)mapping;
    r.size = (size_t)sb.st_size;
    dpf_index *ix = nullptr;
    auto bail = [&](int code, const std::string &msg) {
        munmap(mapping, (size_t)sb.st_size);
        g_create_error = msg;
        if (ix) dpf_destroy(ix);
        return code;
    };
    const void *magic = r.take(8);
    if (!magic || memcmp(magic, SNAP_MAGIC, 8) != 0 || memcmp(r.base + r.size - 8, SNAP_MAGIC, 8) != 0)
        return bail(DPF_EINVAL, "dpf_load: not a snapshot file of this library version (or truncated)");
    dpf_config cfg{};
    int32_t si[32];
    int64_t n = 0, saved_rows = 0;
    r.val(cfg);
    { const void *p = r.take(sizeof(si)); if (p) memcpy(si, p, sizeof(si)); }
    r.val(n); r.val(saved_rows);
    if (!r.ok || n < 0 || saved_rows < 0 || saved_rows > n) return bail(DPF_EINVAL, "dpf_load: damaged header");
    const int64_t saved_begin = cfg.row_begin < 0 ? 0 : cfg.row_begin;       // first id of the rows the file holds
    if (device >= 0) cfg.device = device;
    if (shard) {
        cfg.table_begin = shard->table_begin; cfg.table_end = shard->table_end;
        cfg.subindex_group = shard->subindex_group; cfg.subindex_groups = shard->subindex_groups;
        cfg.row_begin = shard->row_begin; cfg.row_end = shard->row_end;
    }
    int rc = dpf_create(&cfg, &ix);
    if (rc != DPF_OK) { munmap(mapping, (size_t)sb.st_size); return rc; }
    uint64_t n_fns, n_chain, n_b, n_w, n_pf, n_pb, n_pw, n_hh, n_hp;
    const double *fns = r.vec<double>(n_fns);
    const int32_t *chain_fn = r.vec<int32_t>(n_chain);
    const double *b = r.vec<double>(n_b);
    const int32_t *w = r.vec<int32_t>(n_w);
    const double *part_fns = r.vec<double>(n_pf);
    const double *part_b = r.vec<double>(n_pb);
    const int32_t *part_w = r.vec<int32_t>(n_pw);
    const int32_t *hh = r.vec<int32_t>(n_hh);
    const int32_t *hp = r.vec<int32_t>(n_hp);
    const size_t F = (size_t)cfg.num_functions, d = (size_t)cfg.dim, L = (size_t)cfg.num_tables, m = (size_t)cfg.chain_length,
                 pb = (size_t)cfg.partition_bits;
    const size_t in_row = d * ix->elem;
    if (!r.ok || n_fns != F * d || n_chain != L * m || n_hh != (size_t)n * L || n_hp != (size_t)n * L ||
        (cfg.family == DPF_FAMILY_PSTABLE && (n_b != F || n_w != F)) || (pb > 0 && n_pf != L * pb * 32) ||
        (pb > 0 && cfg.family == DPF_FAMILY_PSTABLE && (n_pb != L * pb || n_pw != L * pb)) ||
        r.size - r.at != (size_t)saved_rows * in_row + 8)
        return bail(DPF_EINVAL, "dpf_load: damaged or truncated file");
    if ((rc = dpf_set_hash_family(ix, fns, n_b ? b : nullptr, n_w ? w : nullptr, chain_fn)) != DPF_OK ||
        (pb > 0 && (rc = dpf_set_partitioner(ix, part_fns, n_pb ? part_b : nullptr, n_pw ? part_w : nullptr)) != DPF_OK) ||
        (rc = dpf_set_sampling_index(ix, si)) != DPF_OK)
        return bail(rc, std::string("dpf_load: ") + ix->err);
    // the rows of this handle's shard, streamed from the mapping through the staging buffer
    const int64_t want_begin = ix->row_begin, want_end = std::min<int64_t>(ix->row_end, n);
    const int64_t new_rows = std::max<int64_t>(0, want_end - want_begin);
    if (new_rows > 0 && (want_begin < saved_begin || want_end > saved_begin + saved_rows))
        return bail(DPF_EINVAL, "dpf_load: the file does not hold the rows of the requested shard (save from a handle that holds every row)");
    const unsigned char *rows = r.base + r.at + (size_t)(want_begin - saved_begin) * in_row;
    const size_t row_bytes = (size_t)ix->dpad * ix->elem;
    hipError_t e = ix->d_base.ensure(std::max<size_t>((size_t)new_rows * row_bytes, 16));
    const int64_t batch = 1 << 16;
    if (e == hipSuccess) e = ix->ws_stage.ensure(std::max<size_t>((size_t)std::min<int64_t>(batch, std::max<int64_t>(new_rows, 1)) * in_row, 16));
    for (int64_t s0 = 0; s0 < new_rows && e == hipSuccess; s0 += batch) {
        const int64_t cnt = std::min(batch, new_rows - s0);
        e = hipMemcpy(ix->ws_stage.p, rows + (size_t)s0 * in_row, (size_t)cnt * in_row, hipMemcpyHostToDevice);
        if (e == hipSuccess)
            launch_copy_rows(cfg.dtype, ix->ws_stage.p, cnt, (int)d, (char *)ix->d_base.p + (size_t)s0 * row_bytes, ix->dpad, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    // the hashes of every id, straight from the mapping; the host mirror is refreshed only on demand (dpf_save)
    if (e == hipSuccess) e = ix->d_hashes.ensure(std::max<size_t>((size_t)n_hh * 4, 16));
    if (e == hipSuccess) e = ix->d_parts.ensure(std::max<size_t>((size_t)n_hp * 4, 16));
    if (e == hipSuccess && n_hh) e = hipMemcpy(ix->d_hashes.p, hh, (size_t)n_hh * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_hp) e = hipMemcpy(ix->d_parts.p, hp, (size_t)n_hp * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(DPF_EHIP, std::string("dpf_load: ") + hipGetErrorString(e));
    munmap(mapping, (size_t)sb.st_size);
    ix->host_hashes_valid = n == 0;
    if ((rc = finish_fit(ix, n, new_rows)) != DPF_OK) { g_create_error = "dpf_load: " + ix->err; dpf_destroy(ix); return rc; }
    *out = ix;
    return DPF_OK;
}

extern "C" int dpf_load(const char *path, int32_t device, dpf_index **out) { return dpf_load_shard(path, device, nullptr, out); }

static void snapshot_sizes(dpf_index *ix, int64_t *sizes)
{
    sizes[0] = (int64_t)ix->geo.num_parts * ix->geo.seg * ix->cfg.num_tables;
    sizes[1] = (ix->st.num_dnodes + 1) * ix->geo.dir_node_size;
    sizes[2] = ix->st.num_leaves + 1;
    sizes[3] = 0;
    for (uint32_t len : ix->h_leaf_len) sizes[3] += len;
}

extern "C" int dpf_snapshot_sizes(dpf_index *ix, int64_t *sizes)
{
    if (!ix || !sizes) return DPF_EINVAL;
    int rc = check_ready(ix, true);
    if (rc != DPF_OK) return rc;
    snapshot_sizes(ix, sizes);
    return DPF_OK;
}

extern "C" int dpf_snapshot_copy(dpf_index *ix, uint32_t *roots, uint32_t *dnodes, uint32_t *leaf_off, int32_t *ids)
{
    if (!ix) return DPF_EINVAL;
    int rc = check_ready(ix, true);
    if (rc != DPF_OK) return rc;
    int64_t sz[4];
    snapshot_sizes(ix, sz);
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    if (roots && sz[0] > 0) HIPCHECK(ix, hipMemcpy(roots, ix->d_roots.p, (size_t)sz[0] * 4, hipMemcpyDeviceToHost));
    if (dnodes && sz[1] > 0) HIPCHECK(ix, hipMemcpy(dnodes, ix->d_dnodes.p, (size_t)sz[1] * 4, hipMemcpyDeviceToHost));
    if (leaf_off && sz[2] > 0) HIPCHECK(ix, hipMemcpy(leaf_off, ix->d_leaf_off.p, (size_t)sz[2] * 4, hipMemcpyDeviceToHost));
    if (ids && sz[3] > 0) HIPCHECK(ix, hipMemcpy(ids, ix->d_ids.p, (size_t)sz[3] * 4, hipMemcpyDeviceToHost));
    return DPF_OK;
}

extern "C" int dpf_subindex_distribution(dpf_index *ix, double *out)
{
    if (!ix || !out) return DPF_EINVAL;
    const int P = ix->geo.num_parts, L = ix->cfg.num_tables;
    std::vector<int64_t> cnt((size_t)P, 0);
    for (int t = 0; t < L && ix->part_counts.size() == (size_t)L * P; t++)
        for (int p = 0; p < P; p++) cnt[(size_t)p] += ix->part_counts[(size_t)t * P + p];
    for (int p = 0; p < P; p++) out[p] = (double)cnt[(size_t)p] / (double)L;
    return DPF_OK;
}

// pads the queries to whole 16-byte pieces when needed; returns the device rows to use
static int pad_queries(dpf_index *ix, const void *d_Q, int64_t nq, hipStream_t stream, const void **out, int64_t *stride)
{
    if (ix->dpad == ix->cfg.dim) { *out = d_Q; *stride = ix->cfg.dim; return DPF_OK; }
    HIPCHECK(ix, ix->ws_qpad.ensure((size_t)nq * ix->dpad * ix->elem));
    launch_copy_rows(ix->cfg.dtype, d_Q, nq, ix->cfg.dim, ix->ws_qpad.p, ix->dpad, stream);
    *out = ix->ws_qpad.p;
    *stride = ix->dpad;
    return DPF_OK;
}

// The per-batch buffers of the gather path: the handle's own (grown on demand), or fixed slices of the
// static scratch array for the entry point that allocates nothing.
struct ProbeWs {
    int32_t *hashes = nullptr, *parts = nullptr;     // [nq][L]
    unsigned long long *raw = nullptr;               // [nq]
    int64_t *bound = nullptr, *off = nullptr;        // [nq], [nq + 1]
    int32_t *cnt = nullptr;                          // [nq]
    int32_t *arena = nullptr;                        // candidate lists
    int64_t arena_ids = -1;                          // capacity of a fixed arena; < 0: the handle's, grown to fit
};

static ProbeWs handle_ws(const dpf_index *ix)
{
    ProbeWs w;
    w.hashes = ix->ws_hashes.as<int32_t>(); w.parts = ix->ws_parts.as<int32_t>();
    w.raw = ix->ws_raw.as<unsigned long long>(); w.bound = ix->ws_bound.as<int64_t>();
    w.off = ix->ws_off.as<int64_t>(); w.cnt = ix->ws_cnt.as<int32_t>(); w.arena = ix->ws_arena.as<int32_t>();
    return w;
}

static void fill_probe_params(dpf_index *ix, ProbeParams &pp, const int32_t *d_qids, int32_t steps, const ProbeWs &w)
{
    const ForestGeometry &g = ix->geo;
    pp.roots = ix->d_roots.as<uint32_t>();
    pp.dnodes = ix->d_dnodes.as<uint32_t>();
    pp.leaf_off = ix->d_leaf_off.as<uint32_t>();
    pp.ids = ix->d_ids.as<int32_t>();
    pp.L = g.num_tables; pp.pb = g.partition_bits; pp.bucket_bits = g.bucket_bits; pp.seg = g.seg;
    pp.nbits = g.nbits; pp.dns = g.dir_node_size; pp.max_level = g.max_level;
    pp.steps = steps; pp.exclude_self = ix->cfg.exclude_self;
    pp.t_begin = ix->cfg.table_begin;
    pp.t_end = ix->cfg.table_end <= 0 ? g.num_tables : ix->cfg.table_end;
    pp.sub_mask = 0xffffffffu;
    if (ix->cfg.subindex_groups > 1) {
        pp.sub_mask = 0u;
        for (int s2 = 0; s2 < g.num_parts; s2++)
            if (s2 % ix->cfg.subindex_groups == ix->cfg.subindex_group) pp.sub_mask |= 1u << s2;
    }
    pp.probe_own = ix->probe_mode == DPF_PROBE_OWN ? 1 : 0;
    pp.qids = d_qids;
    pp.hashes = w.hashes;
    pp.parts = w.parts;
    pp.row_begin = ix->row_begin;
    pp.row_end = std::min<int64_t>(ix->row_end, ix->n);
    if (pp.row_end < pp.row_begin) pp.row_end = pp.row_begin;
    const int64_t range = pp.row_end - pp.row_begin;
    const int64_t max_words = 32768;                 // 128 KiB bitmap = 2^20 ids per pass
    int64_t words = (range + 31) / 32;
    pp.chunk_words = (int32_t)std::max<int64_t>(1, std::min(words, max_words));
    pp.raw = w.raw;
    pp.bound = w.bound;
    pp.short_hash_events = ix->d_short_hash;
    pp.cand_off = w.off;
    pp.cand = w.arena;
    pp.cnt = w.cnt;
    pp.lq_cnt = nullptr; pp.lq_off = nullptr; pp.lq_cursor = nullptr; pp.leaf_queries = nullptr;
    pp.q_leaves = nullptr; pp.q_leaf_cnt = nullptr; pp.q_leaf_cap = 0; pp.q_leaf_overflow = nullptr;
    pp.seed_cand = nullptr; pp.seed_cnt = nullptr; pp.seed_cap = 0;
}

// hash -> count -> scan -> (sync: arena size) -> dedupe.  Leaves the per-query
// sets in the arena at off[q], sizes in cnt.  fixed == nullptr: the handle's buffers, grown to fit;
// otherwise the given buffers, and a batch whose lists do not fit the arena is refused.
static int run_probe(dpf_index *ix, const int32_t *d_qids, const void *d_Qpad, int64_t q_stride, int64_t nq,
                     int32_t steps, hipStream_t stream, const ProbeWs *fixed = nullptr)
{
    const int L = ix->cfg.num_tables;
    if (!fixed) {
        HIPCHECK(ix, ix->ws_hashes.ensure((size_t)nq * L * 4));
        HIPCHECK(ix, ix->ws_parts.ensure((size_t)nq * L * 4));
        HIPCHECK(ix, ix->ws_raw.ensure((size_t)nq * 8));
        HIPCHECK(ix, ix->ws_bound.ensure((size_t)nq * 8));
        HIPCHECK(ix, ix->ws_off.ensure((size_t)(nq + 1) * 8));
        HIPCHECK(ix, ix->ws_cnt.ensure((size_t)nq * 4));
    }
    ProbeWs w = fixed ? *fixed : handle_ws(ix);
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[0], stream));
    {
        const int hrc = hash_rows(ix, d_Qpad, q_stride, nq, w.hashes, w.parts, stream, fixed == nullptr);
        if (hrc != DPF_OK) return hrc;
    }
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[1], stream));
    ProbeParams pp;
    fill_probe_params(ix, pp, d_qids, steps, w);
    launch_probe(0, pp, nq, stream);
    launch_scan_i64(pp.bound, nq, w.off, stream);
    HIPCHECK(ix, hipMemcpyAsync(ix->pinned, w.off + nq, 8, hipMemcpyDeviceToHost, stream));
    HIPCHECK(ix, hipStreamSynchronize(stream));
    const int64_t total = ix->pinned[0];
    if (fixed) {
        if (total > fixed->arena_ids)
            return fail(ix, DPF_ECAPACITY, "the batch's candidate lists (" + std::to_string(total) + " ids) exceed the static scratch arena (" +
                                           std::to_string(fixed->arena_ids) + " ids): use fewer queries per call, or dpf_query_topk_device");
    } else {
        HIPCHECK(ix, ix->ws_arena.ensure((size_t)std::max<int64_t>(total, 1) * 4));
        w = handle_ws(ix);
    }
    fill_probe_params(ix, pp, d_qids, steps, w);
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[2], stream));
    launch_probe(1, pp, nq, stream);
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[3], stream));
    HIPCHECK(ix, hipGetLastError());
    ix->last_nq = nq;
    ix->last_valid = !fixed;          // the statistics read the handle's own buffers
    return DPF_OK;
}

static void fill_rerank_params(dpf_index *ix, RerankParams &rp, const void *d_Qpad, int32_t k,
                               const int64_t *off, const int32_t *cnt, const int32_t *cand,
                               int32_t *d_ids, double *d_scores)
{
    rp.Q = d_Qpad;
    rp.q_stride_bytes = ix->dpad * (int64_t)ix->elem;
    rp.base = ix->d_base.p;
    rp.base_stride_bytes = ix->dpad * (int64_t)ix->elem;
    rp.chunks16 = (int32_t)(ix->dpad * (int64_t)ix->elem / 16);
    rp.k = k;
    rp.row_begin = ix->row_begin;
    rp.row_end = ix->row_begin + ix->base_rows;
    rp.cand_off = off; rp.cnt = cnt; rp.cand = cand; rp.cand_stride = 0;
    rp.out_ids = d_ids; rp.out_scores = d_scores;
}

// ws_flag: 16 bytes of flags ([0] overflowed lists, [1] truncated bucket lists, [2] longest forward list), the two
// scan totals (incidences, units) as int64 at byte 16, and from byte 128 the ticket counters of the two scan
// launches (each counter on a 128-byte line of its own)
static const int TICKET_COUNTERS = 16;
static const int MAX_SCAN_PASSES = 3;
// behind the ticket counters: the bins' pair counters of the binned list building (MAX_BINS words)
static const size_t BIN_CURSOR_OFF = 128 + MAX_SCAN_PASSES * (size_t)TICKET_COUNTERS * 128;
static const size_t FLAG_BYTES = BIN_CURSOR_OFF + (size_t)MAX_BINS * 4;
// the per-query forward counts sit behind them, so that one memset clears everything a batch starts from
// Everything a batch starts from at zero sits in ONE buffer, cleared by one memset: the per-bucket incidence
// histogram lq_cnt[num_leaves + 1] first (its size only changes with a fit), the flag block behind it.
static size_t lq_cnt_bytes_of(const dpf_index *ix) { return ((size_t)(ix->num_leaves + 1) * 4 + 255) & ~(size_t)255; }
static uint32_t *lq_cnt_of(const dpf_index *ix) { return (uint32_t *)ix->ws_flag.p; }
static char *flag_block_of(const dpf_index *ix) { return (char *)ix->ws_flag.p + lq_cnt_bytes_of(ix); }
static int32_t *emit_cnt_of(const dpf_index *ix) { return (int32_t *)(flag_block_of(ix) + FLAG_BYTES); }
// Forwarded candidates a query's list holds before the gather path answers the query instead.  The
// handle starts at 8192 and doubles the lists (up to 65536) after a batch in which a list overflowed;
// DPF_OPT_EMIT_CAP pins the value.
static int32_t emit_cap_setting(const dpf_index *ix) { return ix->opt_emit_cap > 0 ? ix->opt_emit_cap : ix->emit_cap; }

// Bucket-major rerank: hash -> count + (query, bucket) histogram -> scan -> bucket->queries lists
// + seed lists -> exact top-k of the seeds (thresholds) -> bucket-major fp32 filter -> exact
// fp64 rerank of the forwarded candidates.  Sets *done = false (nothing written) when a query
// forwarded more than EMIT_CAP candidates; the caller then runs the gather path.
static int run_topk_bucket_major(dpf_index *ix, const int32_t *d_qids, const void *qpad, int64_t qstride, int64_t nq,
                                 int32_t steps, int32_t k, int32_t *d_ids, double *d_scores, hipStream_t stream, bool *done)
{
    *done = false;
    const int32_t EMIT_CAP = emit_cap_setting(ix);
    const int32_t SEED_CAP = ix->opt_seed_cap >= 0 ? ix->opt_seed_cap : ix->auto_seed_cap;   // seed sample per query
    const int L = ix->cfg.num_tables;
    const int64_t nl = ix->num_leaves;
    HIPCHECK(ix, ix->ws_hashes.ensure((size_t)nq * L * 4));
    HIPCHECK(ix, ix->ws_parts.ensure((size_t)nq * L * 4));
    HIPCHECK(ix, ix->ws_raw.ensure((size_t)nq * 8));
    HIPCHECK(ix, ix->ws_bound.ensure((size_t)nq * 8));
    // (cleared sizes are whole 256-byte runs: one fill kernel each instead of a body and a tail)
    const size_t lq_cnt_bytes = lq_cnt_bytes_of(ix);
    const size_t flag_bytes = (FLAG_BYTES + (size_t)nq * 4 + 255) & ~(size_t)255;
    HIPCHECK(ix, ix->ws_lq_cursor.ensure((size_t)(nl + 1) * 4));
    HIPCHECK(ix, ix->ws_lq_off.ensure((size_t)(nl + 2) * 8));
    HIPCHECK(ix, ix->ws_seed_cand.ensure((size_t)nq * SEED_CAP * 4));
    HIPCHECK(ix, ix->ws_seed_cnt.ensure((size_t)nq * 4));
    HIPCHECK(ix, ix->ws_seed_ids.ensure((size_t)nq * k * 4));
    HIPCHECK(ix, ix->ws_seed_scores.ensure((size_t)nq * k * 8));
    HIPCHECK(ix, ix->ws_emit.ensure((size_t)nq * EMIT_CAP * 4));
    HIPCHECK(ix, ix->ws_flag.ensure(lq_cnt_bytes + flag_bytes));
    HIPCHECK(ix, ix->ws_tau.ensure((size_t)nq * 4));

    // what the scan reads of the queries (fp16 rows, norms, rounding distances, tails; int8 rows where the index has them)
    HIPCHECK(ix, ix->ws_q16.ensure((size_t)nq * ix->ks16 * 32 * 2));
    HIPCHECK(ix, ix->ws_q16n.ensure((size_t)nq * 4));
    HIPCHECK(ix, ix->ws_q16e.ensure((size_t)nq * 4));
    HIPCHECK(ix, ix->ws_qc.ensure((size_t)nq * 16));
    const bool tails = ix->filter_dim < ix->cfg.dim;
    if (tails) HIPCHECK(ix, ix->ws_q16t.ensure((size_t)nq * 16));
    const bool i8 = ix->i8_ok && ix->d_slot_nj8.p != nullptr;
    if (i8) {
        HIPCHECK(ix, ix->ws_q8.ensure((size_t)nq * ix->ks8 * 64));
        HIPCHECK(ix, ix->ws_qc8.ensure((size_t)nq * 16));
    }
    HIPCHECK(ix, ix->ws_emit_v.ensure((size_t)nq * EMIT_CAP * 4));
    int32_t *const i8_gate = (int32_t *)flag_block_of(ix) + 8;    // raised when a query does not fit the int8 window
    // l2: the first radii come from the fp16 copies (k_seed_radius, behind the query conversion: an upper bound is
    // all the scan needs, at half the bytes per seed row); dot, or on request: exact scores of the fp32 rows
    const bool seeds_fp16 = SEED_CAP > 0 && ix->cfg.metric == DPF_METRIC_L2 && !ix->opt_exact_seeds;
    // With carried sizes - nothing for the host to read in between - two chains of a batch run on the handle's side
    // stream beside the caller's: the conversions of the query rows (they need only the rows: beside the hashing) and,
    // behind the first probe pass, the lists (offset scans, k_fill_lists, work units: beside the first radii).
    const bool carried = ix->sizes_carried && !ix->opt_sync_sizes;
    const bool fork = carried && ix->opt_side_stream && !ix->profiling && ix->side_stream != nullptr;
    // (an early return between a fork and its join - a failed call, a workspace that cannot grow - must not leave
    // the side stream's kernels writing the handle's buffers behind the caller's back: the guard waits them out)
    struct SideGuard {
        hipStream_t side = nullptr;
        ~SideGuard() { if (side) { (void)hipStreamSynchronize(side); (void)hipGetLastError(); } }
    } side_guard;
    auto convert_queries = [&](hipStream_t cs, const TauParams *tp) {
        launch_rows_to_f16((const float *)qpad, ix->dpad, std::min(ix->cfg.dim, ix->filter_dim), nq, ix->ws_q16.as<uint16_t>(), ix->ks16 * 32,
                           ix->ws_q16n.as<float>(), ix->ws_q16e.as<float>(), cs, tp, ix->cfg.dim,
                           tails ? (uint4 *)ix->ws_q16t.p : nullptr, ix->centred ? ix->d_mu.as<float>() : nullptr);
        if (i8)
            launch_rows_to_i8((const float *)qpad, ix->dpad, ix->cfg.dim, nq, ix->i8_offset, ix->ws_q8.as<int8_t>(), ix->ks8 * 64, nullptr,
                              (float4 *)ix->ws_qc8.p, d_qids, i8_gate, cs);
    };

    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[0], stream));
    // (the memset comes first: the int8 conversion raises a word inside what it clears)
    HIPCHECK(ix, hipMemsetAsync(ix->ws_flag.p, 0, lq_cnt_bytes + flag_bytes, stream));
    // (the conversions beside the hashing: measured -2.5 % per batch on a single tree, whose hashing is short, +0.8 % on 30
    // trees, where they get in the hash kernel's way: by default only for handles that probe up to four tables)
    const int probed_tables = (ix->cfg.table_end > 0 ? ix->cfg.table_end : L) - ix->cfg.table_begin;
    const bool conv_early = fork && seeds_fp16 && (ix->opt_side_stream == 1 || (ix->opt_side_stream < 0 && probed_tables <= 4));
    if (conv_early) {
        HIPCHECK(ix, hipEventRecord(ix->ev_fork0, stream));
        HIPCHECK(ix, hipStreamWaitEvent(ix->side_stream, ix->ev_fork0, 0));
        side_guard.side = ix->side_stream;
        convert_queries(ix->side_stream, nullptr);
        HIPCHECK(ix, hipEventRecord(ix->ev_conv, ix->side_stream));
    }
    {
        const int hrc = hash_rows(ix, qpad, qstride, nq, ix->ws_hashes.as<int32_t>(), ix->ws_parts.as<int32_t>(), stream);
        if (hrc != DPF_OK) return hrc;
    }
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[1], stream));
    ProbeParams pp;
    fill_probe_params(ix, pp, d_qids, steps, handle_ws(ix));
    pp.lq_cnt = lq_cnt_of(ix);
    // the buckets each query reaches are remembered (up to a cap) so that the lists can be filled
    // without walking the forest again
    {
        int nsub = 0;
        for (int s2 = 0; s2 < ix->geo.num_parts; s2++) nsub += __builtin_popcount((unsigned)s2) <= steps;
        const int64_t most = (int64_t)L * 28 * nsub;          // every probe in a bucket of its own
        pp.q_leaf_cap = (int32_t)std::min<int64_t>(most, 4096);
        if (ix->opt_bucket_list_cap > 0) pp.q_leaf_cap = ix->opt_bucket_list_cap;
        ix->last_q_leaf_cap = pp.q_leaf_cap;
        HIPCHECK(ix, ix->ws_q_leaves.ensure((size_t)nq * pp.q_leaf_cap * 8));
        HIPCHECK(ix, ix->ws_q_leaf_cnt.ensure((size_t)nq * 4));
        pp.q_leaves = ix->ws_q_leaves.as<uint2>();
        pp.q_leaf_cnt = ix->ws_q_leaf_cnt.as<int32_t>();
        pp.q_leaf_overflow = (int32_t *)flag_block_of(ix) + 1;
    }
    bool use_bins = false;
    {
        // The counting pass: a lane per (query, table, sub-index) where a query has few of those (a single tree: a wave
        // per table leaves k_probe<0> mostly idle; 18 instead of 25 us on 10k queries), a lane per probe otherwise (30
        // trees, 4 sub-indexes each: 215 us against 230; the 7.7 M returning atomics on the incidence counters, not the
        // walk, are what the time goes to there - without them 99 us against 164).  DPF_OPT_PROBE_LANES pins either.
        int nsub = 0;
        for (int s2 = 0; s2 < ix->geo.num_parts; s2++) nsub += __builtin_popcount((unsigned)s2) <= steps;
        const int lanes_per_query = (pp.t_end - pp.t_begin) * nsub;
        const bool lanes = ix->opt_probe_lanes >= 0 ? ix->opt_probe_lanes != 0 : lanes_per_query <= 16;
        // With carried sizes and bins seeded by an earlier batch: the binned form (no incidence atomics; the lists are
        // built from the pairs bin by bin, below).
        if (carried && ix->bins_seeded && ix->opt_list_bins && ix->nbins > 0) {
            ProbeParams bp = pp;
            bp.bin_cursor = (uint32_t *)(flag_block_of(ix) + BIN_CURSOR_OFF);
            bp.bin_cap = ix->d_bin_cap.as<uint32_t>(); bp.bin_off = ix->d_bin_off.as<uint32_t>();
            bp.bin_map = ix->d_bin_map.as<uint16_t>();
            bp.pairs = ix->ws_pairs.as<uint2>();
            bp.gran_shift = ix->gran_shift; bp.nbins = ix->nbins;
            bp.bin_overflow = (int32_t *)flag_block_of(ix) + 3;
            use_bins = launch_probe_lanes(bp, nq, stream);
        }
        if (!use_bins && !(lanes && launch_probe_lanes(pp, nq, stream))) launch_probe(0, pp, nq, stream);
    }
    HIPCHECK(ix, ix->ws_unit_off.ensure((size_t)(nl + 2) * 8));
    // The two totals of the batch (bucket incidences, work units) size leaf_queries[] and units[].  The first
    // batch of a handle reads them back here; later batches do not wait: the kernels behind this point take the
    // totals from device memory, the buffers keep the headroom earlier batches left them, and a batch that
    // outgrows them (or truncated a bucket list) raises a flag, writes nothing and is run again the slow way.
    const int64_t cap_pairs = std::min<int64_t>((int64_t)(ix->ws_leaf_queries.cap / 4), 0xfffffffell);
    const int64_t cap_units = std::min<int64_t>((int64_t)(ix->ws_units.cap / 16), 0xfffffffell);
    int32_t *const flags = (int32_t *)flag_block_of(ix);
    int64_t *const d_totals = (int64_t *)(flag_block_of(ix) + 16);
    hipStream_t ls = stream;
    if (fork) {
        HIPCHECK(ix, hipEventRecord(ix->ev_fork, stream));
        HIPCHECK(ix, hipStreamWaitEvent(ix->side_stream, ix->ev_fork, 0));
        ls = ix->side_stream;
        side_guard.side = ix->side_stream;
    }
    if (use_bins) {
        HIPCHECK(ix, ix->ws_bin_units.ensure((size_t)ix->nbins * 8));
        HIPCHECK(ix, ix->ws_bin_unit_base.ensure((size_t)ix->nbins * 8));
        BinListParams bl;
        bl.bin_cursor = (const uint32_t *)(flag_block_of(ix) + BIN_CURSOR_OFF);
        bl.bin_cap = ix->d_bin_cap.as<uint32_t>(); bl.bin_off = ix->d_bin_off.as<uint32_t>();
        bl.bin_lo = ix->d_bin_lo.as<uint32_t>();
        bl.pairs = ix->ws_pairs.as<uint2>();
        bl.nbins = ix->nbins; bl.nl = nl; bl.nq = nq; bl.stage = ix->opt_list_bins == 2 ? 0 : 1;
        bl.leaf_tile0 = ix->d_leaf_tile0.as<uint32_t>();
        bl.cap_pairs = cap_pairs; bl.flags = flags;
        bl.lq_cnt = lq_cnt_of(ix); bl.lq_off = ix->ws_lq_off.as<int64_t>(); bl.unit_off = ix->ws_unit_off.as<int64_t>();
        bl.bin_units = ix->ws_bin_units.as<int64_t>(); bl.leaf_queries = ix->ws_leaf_queries.as<int32_t>();
        launch_bin_lists(bl, ls);
        launch_bin_prefix(bl.bin_cursor, bl.bin_units, ix->nbins, cap_units, ix->ws_bin_unit_base.as<int64_t>(), d_totals, flags, ls);
    } else {
        launch_bucket_scans(lq_cnt_of(ix), ix->d_leaf_tile0.as<uint32_t>(), nl, ix->ws_lq_off.as<int64_t>(),
                            ix->ws_unit_off.as<int64_t>(), d_totals, carried, cap_pairs, cap_units, flags,
                            ix->ws_lq_cursor.as<uint32_t>(), ls);
    }
    int64_t pairs = 0, nunits = 0;
    bool lists_truncated = false;
    if (!carried) {
        HIPCHECK(ix, hipMemcpyAsync(ix->pinned, flags, 32, hipMemcpyDeviceToHost, stream));   // flags + the two totals
        HIPCHECK(ix, hipStreamSynchronize(stream));
        pairs = ix->pinned[2]; nunits = ix->pinned[3];
        lists_truncated = (int32_t)((uint64_t)ix->pinned[0] >> 32) != 0;
        // unit descriptors index leaf_queries[] and the unit list with 32 bits
        if (pairs >= (int64_t)0xffffffffll || nunits >= (int64_t)0xffffffffll) return DPF_OK;
    }
    if (!carried) HIPCHECK(ix, ix->ws_leaf_queries.ensure((size_t)std::max<int64_t>(pairs + pairs / 2, 1024) * 4));
    if (!carried && !lists_truncated && ix->ngran > 0 && ix->opt_list_bins) {
        // the bins and their regions for the batches to come, cut from this batch's counts
        ix->bins_seeded = false;
        HIPCHECK(ix, ix->d_bin_cap.ensure((size_t)BIN_MAX * 4));
        HIPCHECK(ix, ix->d_bin_off.ensure((size_t)BIN_MAX * 4));
        HIPCHECK(ix, ix->d_bin_lo.ensure((size_t)(BIN_MAX + 1) * 4));
        HIPCHECK(ix, ix->d_bin_map.ensure((size_t)ix->ngran * 2));
        HIPCHECK(ix, ix->d_gran_cnt.ensure((size_t)ix->ngran * 4));
        unsigned long long *const info = (unsigned long long *)(ix->d_proj_counters + 32);
        launch_bin_seed(lq_cnt_of(ix), nl, ix->gran_shift, ix->ngran, ix->d_gran_cnt.as<uint32_t>(), ix->d_bin_map.as<uint16_t>(),
                        ix->d_bin_lo.as<uint32_t>(), ix->d_bin_cap.as<uint32_t>(), ix->d_bin_off.as<uint32_t>(), info, stream);
        HIPCHECK(ix, hipMemcpyAsync(ix->pinned, info, 16, hipMemcpyDeviceToHost, stream));
        HIPCHECK(ix, hipStreamSynchronize(stream));
        const int64_t total_cap = ix->pinned[0], nb = ix->pinned[1];
        ix->nbins = 0;
        if (nb > 0 && nb <= BIN_MAX && total_cap < (int64_t)0xffffffffll) {
            HIPCHECK(ix, ix->ws_pairs.ensure((size_t)total_cap * 8));
            ix->nbins = (int32_t)nb;
            ix->bins_seeded = true;
        }
    }
    pp.abort = carried ? flags + 3 : nullptr;
    pp.lq_off = ix->ws_lq_off.as<int64_t>();
    pp.lq_cursor = ix->ws_lq_cursor.as<uint32_t>();
    pp.lq_off32 = ix->ws_lq_cursor.as<uint32_t>();    // (the second walk, which needs the cursors, does not use these)
    pp.leaf_queries = ix->ws_leaf_queries.as<int32_t>();
    pp.seed_cand = ix->ws_seed_cand.as<int32_t>();
    pp.seed_cnt = ix->ws_seed_cnt.as<int32_t>();
    pp.seed_cap = SEED_CAP;
    if (lists_truncated) {
        HIPCHECK(ix, hipMemsetAsync(ix->ws_lq_cursor.p, 0, (size_t)(nl + 1) * 4, stream));
        launch_probe(2, pp, nq, stream);
    }
    else if (!use_bins) launch_fill_lists(pp, nq, ls);
    if (!carried) HIPCHECK(ix, ix->ws_units.ensure((size_t)std::max<int64_t>(nunits + nunits / 2, 1024) * 16));
    launch_units(lq_cnt_of(ix), ix->ws_lq_off.as<int64_t>(), ix->d_leaf_tile0.as<uint32_t>(),
                 (const uint4 *)ix->d_tiles.p, ix->nutiles, ix->ws_unit_off.as<int64_t>(), (uint4 *)ix->ws_units.p,
                 carried ? flags + 3 : nullptr, ls, use_bins ? ix->ws_bin_unit_base.as<int64_t>() : nullptr,
                 ix->d_bin_map.as<uint16_t>(), ix->gran_shift);
    if (fork) HIPCHECK(ix, hipEventRecord(ix->ev_join, ls));
    if (!lists_truncated) launch_seed_sample(pp, nq, stream);
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[2], stream));

    // thresholds: exact top-k of each query's seed sample
    RerankParams rp;
    fill_rerank_params(ix, rp, qpad, k, nullptr, ix->ws_seed_cnt.as<int32_t>(), ix->ws_seed_cand.as<int32_t>(),
                       ix->ws_seed_ids.as<int32_t>(), ix->ws_seed_scores.as<double>());
    rp.cand_stride = SEED_CAP;
    const bool short_lists = ix->opt_short_rerank;
    if (SEED_CAP > 0 && pp.t_end - pp.t_begin > 1) {
        // a near neighbour sits in the query's own bucket of several tables: every id of the sample once
        HIPCHECK(ix, ix->ws_emit_ucnt.ensure((size_t)nq * 4));
        launch_unique_lists(ix->ws_seed_cand.as<int32_t>(), ix->ws_seed_cnt.as<int32_t>(), SEED_CAP, SEED_CAP, ix->ws_emit_ucnt.as<int32_t>(), nq, stream);
        rp.cnt = ix->ws_emit_ucnt.as<int32_t>();
    }
    if (SEED_CAP > 0 && !seeds_fp16) {
        if (short_lists && rerank_short_applies(rp)) launch_rerank_short(ix->cfg.dtype, ix->cfg.metric, rp, nq, stream);
        else launch_rerank(ix->cfg.dtype, ix->cfg.metric, rp, nq, stream);
    }

    // fp16 copies of the queries (unless they were made beside the hashing), first thresholds
    {   // the conversion's last lane of every query row also writes the query's first threshold and scan record
        TauParams tp;
        tp.metric = ix->cfg.metric;
        if (SEED_CAP > 0 && !seeds_fp16) { tp.seed_scores = ix->ws_seed_scores.as<double>(); tp.seed_ids = ix->ws_seed_ids.as<int32_t>(); }
        tp.k = k; tp.qids = d_qids; tp.tau = ix->ws_tau.as<int32_t>(); tp.qc = (float4 *)ix->ws_qc.p;
        if (conv_early) HIPCHECK(ix, hipStreamWaitEvent(stream, ix->ev_conv, 0));
        else convert_queries(stream, seeds_fp16 ? nullptr : &tp);
        if (seeds_fp16)
            launch_seed_radius(rp.cand, rp.cnt, SEED_CAP, ix->ws_q16.as<uint16_t>(), ix->d_base16.as<uint16_t>(), ix->ks16,
                               ix->ws_q16n.as<float>(), ix->ws_q16e.as<float>(), ix->d_base16e.as<float>(),
                               tails ? (const uint4 *)ix->ws_q16t.p : nullptr, tails ? (const uint4 *)ix->d_base16t.p : nullptr,
                               ix->row_begin, ix->row_begin + ix->base_rows, tp, nq, stream);
    }
    if (fork) {                                                             // the scan needs both chains
        HIPCHECK(ix, hipStreamWaitEvent(stream, ix->ev_join, 0));
        side_guard.side = nullptr;
    } else if (conv_early) {
        side_guard.side = nullptr;
    }
    LeafScanParams lp;
    lp.ids = ix->d_ids.as<int32_t>();
    lp.leaf_queries = ix->ws_leaf_queries.as<int32_t>();
    lp.row_begin = ix->row_begin; lp.row_end = ix->row_begin + ix->base_rows;
    lp.qids = d_qids; lp.exclude_self = ix->cfg.exclude_self;
    lp.tau = ix->ws_tau.as<int32_t>();
    lp.k = k;
    lp.emit = ix->ws_emit.as<int32_t>(); lp.emit_v = ix->ws_emit_v.as<float>();
    lp.emit_cnt = emit_cnt_of(ix); lp.emit_cap = EMIT_CAP;
    lp.debug = 0;
    lp.Q16 = ix->ws_q16.as<uint16_t>(); lp.base16 = ix->d_base16.as<uint16_t>();
    if (ix->slot16_ok) { lp.base16 = ix->d_slot16.as<uint16_t>(); lp.rows_in_slot_order = 1; }
    lp.dbg_slots = (int64_t)(ix->d_ids.cap / 4); lp.dbg_rows = std::max<int64_t>(ix->base_rows, 1); lp.dbg_nq = nq;
    lp.dbg_pairs = (int64_t)(ix->ws_leaf_queries.cap / 4);
    lp.dbg_violations = (unsigned long long *)(ix->d_proj_counters + 16);
    lp.slot_norm = ix->d_slot_norm.as<float>(); lp.qc = (const float4 *)ix->ws_qc.p;
    lp.units = (const uint4 *)ix->ws_units.p;
    lp.num_cus = ix->num_cus; lp.ks = ix->ks16;
    if (ix->filter_dim < ix->cfg.dim) { lp.slot_tail = (const uint4 *)ix->d_slot_tail.p; lp.q_tail = (const uint4 *)ix->ws_q16t.p; }
    lp.num_counters = TICKET_COUNTERS;
    lp.tickets = (unsigned int *)(flag_block_of(ix) + 128);
    // Slices of the units with the thresholds refined from what has been forwarded in between: a first 15 %, then the
    // rest; for long rows (one wave per SIMD: nothing hides the cost of a forwarded row) a fiftieth goes before them.
    const int first_permille = ix->opt_first_permille;
    // (three also where the query's own buckets hold so many rows that the seed sample - 1024 and more by the rule of
    // seed_sample_for - is a small part of them: the first radii are then loose, and a refinement after a twenty-fifth of
    // the units pays: 30 trees over SIFT-shaped rows, buckets of up to 31 k rows, 2.41 -> 2.30 ms per batch with 354 instead
    // of 642 rows forwarded per query; 30 trees of 100-d clustered rows, sample 384: 2.39 -> 2.43 ms, left at two)
    const int passes = ix->opt_scan_passes ? ix->opt_scan_passes : (ix->ks16 >= 16 || ix->auto_seed_cap >= 1024 ? 3 : 2);
    // A batch of few units (a single-table forest: a dozen units per wave) gains nothing from a refinement that costs
    // two more launches: one slice.  Judged by this batch's count when the host has read it, by the last batch's otherwise.
    const int64_t units_hint = carried ? ix->prev_units : nunits;
    const bool tiny = ix->opt_scan_passes == 0 && units_hint >= 0 && units_hint < (int64_t)ix->num_cus * 12 * 16;
    int cuts[MAX_SCAN_PASSES + 1], ncuts = 0;
    cuts[ncuts++] = 0;
    if (!tiny && passes >= 3 && ix->opt_early_permille > 0 && ix->opt_early_permille < first_permille) cuts[ncuts++] = ix->opt_early_permille;
    if (!tiny && first_permille > 0 && first_permille < 1000) cuts[ncuts++] = first_permille;
    cuts[ncuts++] = 1000;
    if (carried) { lp.totals = d_totals; lp.abort = flags + 3; }
    for (int e = 0; e < 6; e += 2) ix->ev_scan_used[e / 2] = false;
    for (int s = 0; s + 1 < ncuts; s++) {
        lp.permille_lo = cuts[s]; lp.permille_hi = cuts[s + 1];
        lp.unit_begin = nunits * cuts[s] / 1000; lp.unit_end = nunits * cuts[s + 1] / 1000;
        if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev_scan[2 * s], stream));
        if (i8) {
            // both forms of the slice are enqueued; the word the query conversion wrote lets one of them run
            LeafScanParams l8 = lp;
            l8.i8 = 1; l8.ks = ix->ks8;
            l8.Q16 = (const uint16_t *)ix->ws_q8.p; l8.base16 = (const uint16_t *)ix->d_base8.p;
            l8.qc = (const float4 *)ix->ws_qc8.p; l8.slot_nj8 = ix->d_slot_nj8.as<int32_t>();
            l8.slot_tail = nullptr; l8.q_tail = nullptr;
            l8.gate = i8_gate; l8.gate_want = 0;
            l8.rows_in_slot_order = 0;
            launch_leafscan(ix->cfg.metric, l8, stream);
            lp.gate = i8_gate; lp.gate_want = 1;
        }
        launch_leafscan(ix->cfg.metric, lp, stream);
        if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev_scan[2 * s + 1], stream));
        ix->ev_scan_used[s] = ix->profiling;
        if (s + 2 < ncuts && (lp.unit_end > 0 || carried))
            launch_refine(ix->cfg.metric, lp.emit, lp.emit_v, lp.emit_cnt, EMIT_CAP, k, nq, ix->ws_q16e.as<float>(), lp.tau, stream);
        lp.tickets += TICKET_COUNTERS * 32;               // every launch has its own counters
    }
    HIPCHECK(ix, ix->ws_ovf.ensure((size_t)nq * 4));
    // one small kernel lists the overflowed queries and writes the batch's 40-byte record (flags, totals, int8 gate word) straight
    // into the pinned host words the host reads after the wait
    launch_overflow_list(emit_cnt_of(ix), nq, EMIT_CAP, flags, ix->ws_ovf.as<int32_t>(), (int32_t *)(ix->pinned + 4), stream);
    HIPCHECK(ix, hipStreamSynchronize(stream));
    if (carried) {
        if ((int32_t)((uint64_t)ix->pinned[5] >> 32) != 0) {         // abandoned on the device: again, reading the sizes first
            ix->sizes_carried = false;
            const int rc = run_topk_bucket_major(ix, d_qids, qpad, qstride, nq, steps, k, d_ids, d_scores, stream, done);
            ix->st.last_sizes_carried = 2;
            return rc;
        }
        pairs = ix->pinned[6]; nunits = ix->pinned[7];
    }
    ix->st.last_sizes_carried = carried ? 1 : 0;
    ix->last_scan_int8 = i8 && (int32_t)(ix->pinned[8] & 0xffffffff) == 0;
    ix->sizes_carried = !lists_truncated;
    ix->scan_launches = 0;
    for (int s2 = 0; s2 + 1 < ncuts; s2++) ix->scan_launches += nunits * cuts[s2 + 1] / 1000 > nunits * cuts[s2] / 1000 ? 1 : 0;
    if (ix->scan_launches == 0) ix->scan_launches = 1;
    ix->st.last_pairs = pairs;
    ix->prev_units = nunits;
    // Queries whose forward list overflowed are answered by the gather path afterwards, as a batch of
    // their own; when they are many the whole batch goes that way.
    const int64_t n_ovf = (int32_t)(ix->pinned[4] & 0xffffffff);
    const int32_t longest = (int32_t)(ix->pinned[5] & 0xffffffff);
    ix->st.last_overflowed = n_ovf;
    ix->st.last_longest_list = longest;
    if (n_ovf > 0 && ix->opt_emit_cap <= 0 && ix->emit_cap < 65536) ix->emit_cap *= 2;   // from the next batch on
    if (n_ovf * 8 > nq) return DPF_OK;

    // distinct ids of every forward list (a row is forwarded once per tree it was found in)
    const bool dedupe = unique_lists_lds(std::min(longest, EMIT_CAP)) <= 128 * 1024;   // the table of the longest list fits LDS
    if (dedupe) {
        HIPCHECK(ix, ix->ws_emit_ucnt.ensure((size_t)nq * 4));
        // (entries forwarded under a threshold the query has tightened past since are left out as well)
        PruneParams pr;
        if (!ix->opt_keep_forwarded) {
            pr.emit_v = ix->ws_emit_v.as<float>(); pr.tau = ix->ws_tau.as<int32_t>();
            pr.qc = (const float4 *)ix->ws_qc.p; pr.qc8 = i8 ? (const float4 *)ix->ws_qc8.p : nullptr;
            pr.q_tail = ix->filter_dim < ix->cfg.dim ? (const uint4 *)ix->ws_q16t.p : nullptr;
            pr.gate = i8 ? i8_gate : nullptr;
            pr.metric = ix->cfg.metric;
            pr.c32 = leafscan_c32_of(ix->ks16 * 32);
            pr.xr_max = ix->tile_bounds[0]; pr.cx_max = ix->tile_bounds[1]; pr.xt_max = ix->tile_bounds[2];
        }
        launch_unique_lists(ix->ws_emit.as<int32_t>(), emit_cnt_of(ix), EMIT_CAP, longest, ix->ws_emit_ucnt.as<int32_t>(), nq,
                            stream, &pr);
    }
    fill_rerank_params(ix, rp, qpad, k, nullptr, dedupe ? ix->ws_emit_ucnt.as<int32_t>() : emit_cnt_of(ix),
                       ix->ws_emit.as<int32_t>(), d_ids, d_scores);
    rp.cand_stride = EMIT_CAP;
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[4], stream));
    // (deduplicated and pruned, the lists are a few tens of rows long)
    if (short_lists && rerank_short_applies(rp)) launch_rerank_short(ix->cfg.dtype, ix->cfg.metric, rp, nq, stream, dedupe && !ix->opt_keep_forwarded);
    else launch_rerank(ix->cfg.dtype, ix->cfg.metric, rp, nq, stream);
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[5], stream));
    HIPCHECK(ix, hipGetLastError());
    if (n_ovf > 0) {
        const int64_t row_bytes = ix->dpad * (int64_t)ix->elem;
        HIPCHECK(ix, ix->sub_q.ensure((size_t)n_ovf * row_bytes));
        HIPCHECK(ix, ix->sub_qids.ensure((size_t)n_ovf * 4));
        HIPCHECK(ix, ix->sub_ids.ensure((size_t)n_ovf * k * 4));
        HIPCHECK(ix, ix->sub_scores.ensure((size_t)n_ovf * k * 8));
        launch_gather_queries(ix->ws_ovf.as<int32_t>(), n_ovf, qpad, row_bytes, d_qids, ix->sub_q.p, ix->sub_qids.as<int32_t>(), stream);
        // the gather path works in the per-batch buffers whose contents the statistics of this batch read
        DevBuf *mine[] = {&ix->ws_hashes, &ix->ws_parts, &ix->ws_raw, &ix->ws_bound, &ix->ws_off, &ix->ws_cnt};
        for (int i = 0; i < 6; i++) std::swap(*mine[i], ix->sub_ws[i]);
        const bool prof = ix->profiling;
        ix->profiling = false;
        int rc = run_probe(ix, d_qids ? ix->sub_qids.as<int32_t>() : nullptr, ix->sub_q.p, qstride, n_ovf, steps, stream);
        if (rc == DPF_OK) {
            RerankParams sp;
            fill_rerank_params(ix, sp, ix->sub_q.p, k, ix->ws_off.as<int64_t>(), ix->ws_cnt.as<int32_t>(), ix->ws_arena.as<int32_t>(),
                               ix->sub_ids.as<int32_t>(), ix->sub_scores.as<double>());
            launch_rerank(ix->cfg.dtype, ix->cfg.metric, sp, n_ovf, stream);
            launch_scatter_results(ix->ws_ovf.as<int32_t>(), n_ovf, k, ix->sub_ids.as<int32_t>(), ix->sub_scores.as<double>(), d_ids,
                                   d_scores, stream);
        }
        ix->profiling = prof;
        for (int i = 0; i < 6; i++) std::swap(*mine[i], ix->sub_ws[i]);
        if (rc != DPF_OK) return rc;
        HIPCHECK(ix, hipGetLastError());
    }
    ix->st.last_overflowed = n_ovf;
    ix->last_nq = nq;
    ix->last_valid = true;
    ix->last_path = 1;
    *done = true;
    return DPF_OK;
}

extern "C" int dpf_query_topk_device(dpf_index *ix, const int32_t *d_qids, const void *d_Q, int64_t nq,
                                     int32_t steps, int32_t k, int32_t *d_ids, double *d_scores, void *stream_)
{
    if (!ix) return DPF_EINVAL;
    int rc = check_ready(ix, true);
    if (rc != DPF_OK) return rc;
    if (nq < 0 || steps < 0 || k < 1 || k > 64 || (nq > 0 && (!d_Q || !d_ids || !d_scores)))
        return fail(ix, DPF_EINVAL, "dpf_query_topk_device: bad arguments");
    if (nq == 0) return DPF_OK;
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    ix->sets_valid = false;
    const void *qpad; int64_t qstride;
    // the rerank kernels read whole 16-byte pieces, so they always get padded rows
    if ((rc = pad_queries(ix, d_Q, nq, stream, &qpad, &qstride)) != DPF_OK) return rc;
    ix->last_k = k;
    ix->last_timed = ix->profiling;
    ix->st.total_topk_batches++;
    if (ix->bucket_major && ix->ntiles > 0) {
        bool done = false;
        if ((rc = run_topk_bucket_major(ix, d_qids, qpad, qstride, nq, steps, k, d_ids, d_scores, stream, &done)) != DPF_OK) return rc;
        if (done) {
            const int64_t kind = ix->st.last_sizes_carried;
            (kind == 0 ? ix->st.total_sized_on_host : kind == 1 ? ix->st.total_carried : ix->st.total_rerun)++;
            ix->st.total_overflowed += ix->st.last_overflowed;
            return DPF_OK;
        }
    }
    ix->st.total_gather_batches++;
    if ((rc = run_probe(ix, d_qids, qpad, qstride, nq, steps, stream)) != DPF_OK) return rc;
    RerankParams rp;
    fill_rerank_params(ix, rp, qpad, k, ix->ws_off.as<int64_t>(), ix->ws_cnt.as<int32_t>(), ix->ws_arena.as<int32_t>(), d_ids, d_scores);
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[4], stream));
    launch_rerank(ix->cfg.dtype, ix->cfg.metric, rp, nq, stream);
    if (ix->profiling) HIPCHECK(ix, hipEventRecord(ix->ev[5], stream));
    HIPCHECK(ix, hipGetLastError());
    ix->last_path = 0;
    return DPF_OK;
}

extern "C" int dpf_query_topk_device_static(dpf_index *ix, const int32_t *d_qids, const void *d_Q, int64_t nq,
                                            int32_t steps, int32_t k, int32_t *d_ids, double *d_scores, void *stream_)
{
    if (!ix) return DPF_EINVAL;
    int rc = check_ready(ix, true);
    if (rc != DPF_OK) return rc;
    if (nq < 0 || steps < 0 || k < 1 || k > 64 || (nq > 0 && (!d_Q || !d_ids || !d_scores)))
        return fail(ix, DPF_EINVAL, "dpf_query_topk_device_static: bad arguments");
    if (nq == 0) return DPF_OK;
    const int64_t row_bytes = ix->dpad * (int64_t)ix->elem;
    if ((size_t)nq > SCRATCH_MAX_QUERIES || (size_t)(nq * row_bytes) > SCRATCH_QPAD_BYTES ||
        (size_t)(nq * ix->cfg.num_tables) > SCRATCH_HASH_WORDS)
        return fail(ix, DPF_ECAPACITY, "dpf_query_topk_device_static: at most " + std::to_string(SCRATCH_MAX_QUERIES) + " queries, " +
                                       std::to_string(SCRATCH_QPAD_BYTES) + " bytes of padded query rows and " +
                                       std::to_string(SCRATCH_HASH_WORDS) + " (query, table) pairs per call");
    if (!ix->scratch_ptr) return fail(ix, DPF_EHIP, "dpf_query_topk_device_static: the scratch array was not located on this device");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    // One call at a time per GPU.  On the host: the lock.  On the device: this call's kernels are ordered behind the
    // last kernel of the previous call that used the array, whatever stream or handle that call came from; the event
    // is recorded again on every way out, behind whatever this call has enqueued.
    ScratchDev &sd = g_scratch[ix->cfg.device];
    std::unique_lock<std::mutex> turn(sd.busy, std::try_to_lock);
    if (!turn.owns_lock())
        return fail(ix, DPF_EBUSY, "dpf_query_topk_device_static: another call is using this GPU's scratch array (one call at a time per GPU)");
    if (sd.recorded) HIPCHECK(ix, hipStreamWaitEvent(stream, sd.done, 0));
    struct Mark {
        ScratchDev &sd; hipStream_t stream;
        ~Mark() { if (hipEventRecord(sd.done, stream) == hipSuccess) sd.recorded = true; else (void)hipGetLastError(); }
    } mark{sd, stream};
    unsigned char *const sp = ix->scratch_ptr;
    ProbeWs w;
    w.hashes = (int32_t *)(sp + SCRATCH_OFF_HASHES); w.parts = (int32_t *)(sp + SCRATCH_OFF_PARTS);
    w.raw = (unsigned long long *)(sp + SCRATCH_OFF_RAW); w.bound = (int64_t *)(sp + SCRATCH_OFF_BOUND);
    w.off = (int64_t *)(sp + SCRATCH_OFF_OFF); w.cnt = (int32_t *)(sp + SCRATCH_OFF_CNT);
    w.arena = (int32_t *)(sp + SCRATCH_OFF_ARENA); w.arena_ids = (int64_t)SCRATCH_ARENA_IDS;
    // the rerank kernel reads whole 16-byte pieces: padded copies of the query rows when the rows are no multiple
    const void *qpad = d_Q;
    int64_t qstride = ix->cfg.dim;
    if (ix->dpad != ix->cfg.dim) {
        launch_copy_rows(ix->cfg.dtype, d_Q, nq, ix->cfg.dim, sp + SCRATCH_OFF_QPAD, ix->dpad, stream);
        qpad = sp + SCRATCH_OFF_QPAD;
        qstride = ix->dpad;
    }
    // every slice is written in full before it is read (hashes, counts, offsets, set sizes, lists): nothing of
    // an earlier call's contents is relied on, so nothing needs clearing
    const bool prof = ix->profiling;
    ix->profiling = false;
    rc = run_probe(ix, d_qids, qpad, qstride, nq, steps, stream, &w);
    ix->profiling = prof;
    if (rc != DPF_OK) return rc;
    RerankParams rp;
    fill_rerank_params(ix, rp, qpad, k, w.off, w.cnt, w.arena, d_ids, d_scores);
    launch_rerank(ix->cfg.dtype, ix->cfg.metric, rp, nq, stream);
    HIPCHECK(ix, hipGetLastError());
    return DPF_OK;
}

// Host rows -> device, enqueued on `stream`.  Pinned memory is copied from where it lies; pageable memory goes through
// the handle's pinned staging area (at byte `at`) in 1 MiB pieces, so that the transfer of one piece runs while the
// next is being staged.  The staging area is reused only after the call's final stream wait.
static int stage_to_device(dpf_index *ix, void *dst, const void *src, size_t bytes, size_t at, hipStream_t stream)
{
    if (bytes == 0) return DPF_OK;
    if (host_pinned(src)) {
        HIPCHECK(ix, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return DPF_OK;
    }
    const size_t piece = (size_t)1 << 20;
    char *stage = (char *)ix->h_stage.p + at;
    for (size_t o = 0; o < bytes; o += piece) {
        const size_t len = std::min(piece, bytes - o);
        memcpy(stage + o, (const char *)src + o, len);
        HIPCHECK(ix, hipMemcpyAsync((char *)dst + o, stage + o, len, hipMemcpyHostToDevice, stream));
    }
    return DPF_OK;
}

extern "C" int dpf_query_topk(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps,
                              int32_t k, int32_t *ids, double *scores)
{
    if (!ix) return DPF_EINVAL;
    int rc = check_ready(ix, true);
    if (rc != DPF_OK) return rc;
    if (nq < 0 || (nq > 0 && (!Q || !ids || !scores)) || k < 1 || k > 64) return fail(ix, DPF_EINVAL, "dpf_query_topk: bad arguments");
    if (nq == 0) return DPF_OK;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    hipStream_t stream = nullptr;
    const size_t qbytes = (size_t)nq * ix->cfg.dim * ix->elem, idbytes = (size_t)nq * 4;
    const size_t obytes_i = (size_t)nq * k * 4, obytes_s = (size_t)nq * k * 8;
    // staging layout: query rows, query ids, result scores, result ids (each part 256-byte aligned)
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t at_ids = up(qbytes), at_os = at_ids + up(idbytes), at_oi = at_os + up(obytes_s);
    HIPCHECK(ix, ix->h_stage.ensure(at_oi + up(obytes_i)));
    HIPCHECK(ix, ix->ws_hq.ensure(qbytes));
    HIPCHECK(ix, ix->ws_hids.ensure(obytes_i));
    HIPCHECK(ix, ix->ws_hscores.ensure(obytes_s));
    if ((rc = stage_to_device(ix, ix->ws_hq.p, Q, qbytes, 0, stream)) != DPF_OK) return rc;
    const int32_t *dq = nullptr;
    if (qids) {
        HIPCHECK(ix, ix->ws_hqids.ensure(idbytes));
        if ((rc = stage_to_device(ix, ix->ws_hqids.p, qids, idbytes, at_ids, stream)) != DPF_OK) return rc;
        dq = ix->ws_hqids.as<int32_t>();
    }
    rc = dpf_query_topk_device(ix, dq, ix->ws_hq.p, nq, steps, k, ix->ws_hids.as<int32_t>(), ix->ws_hscores.as<double>(), stream);
    if (rc != DPF_OK) return rc;
    // results: straight into pinned caller memory, else into the staging area and from there with the CPU
    const bool pin_i = host_pinned(ids), pin_s = host_pinned(scores);
    void *ti = pin_i ? (void *)ids : (void *)((char *)ix->h_stage.p + at_oi);
    void *ts = pin_s ? (void *)scores : (void *)((char *)ix->h_stage.p + at_os);
    HIPCHECK(ix, hipMemcpyAsync(ti, ix->ws_hids.p, obytes_i, hipMemcpyDeviceToHost, stream));
    HIPCHECK(ix, hipMemcpyAsync(ts, ix->ws_hscores.p, obytes_s, hipMemcpyDeviceToHost, stream));
    HIPCHECK(ix, hipStreamSynchronize(stream));
    if (!pin_i) memcpy(ids, ti, obytes_i);
    if (!pin_s) memcpy(scores, ts, obytes_s);
    return DPF_OK;
}

// Pinned host memory for callers of the host-pointer entry points (a JVM wraps it in a direct ByteBuffer): arguments
// and results that lie in it are moved by the device's copy engines without a staging copy.
extern "C" int dpf_host_alloc(size_t bytes, void **out)
{
    if (!out) return DPF_EINVAL;
    *out = nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        g_create_error = "dpf_host_alloc: hipHostMalloc failed";
        return DPF_ENOMEM;
    }
    *out = p;
    return DPF_OK;
}

extern "C" void dpf_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int dpf_hash_batch(dpf_index *ix, const void *Q, int64_t nq, int32_t *hashes, int32_t *parts)
{
    if (!ix) return DPF_EINVAL;
    int rc = check_ready(ix, false);
    if (rc != DPF_OK) return rc;
    if (nq < 0 || (nq > 0 && (!Q || !hashes))) return fail(ix, DPF_EINVAL, "dpf_hash_batch: bad arguments");
    if (nq == 0) return DPF_OK;
    ix->sets_valid = false;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    const int L = ix->cfg.num_tables;
    const size_t qbytes = (size_t)nq * ix->cfg.dim * ix->elem;
    HIPCHECK(ix, ix->ws_hq.ensure(qbytes));
    HIPCHECK(ix, ix->ws_hashes.ensure((size_t)nq * L * 4));
    HIPCHECK(ix, ix->ws_parts.ensure((size_t)nq * L * 4));
    HIPCHECK(ix, hipMemcpy(ix->ws_hq.p, Q, qbytes, hipMemcpyHostToDevice));
    if ((rc = hash_rows(ix, ix->ws_hq.p, ix->cfg.dim, nq, ix->ws_hashes.as<int32_t>(), ix->ws_parts.as<int32_t>(), nullptr)) != DPF_OK) return rc;
    HIPCHECK(ix, hipGetLastError());
    HIPCHECK(ix, hipMemcpy(hashes, ix->ws_hashes.p, (size_t)nq * L * 4, hipMemcpyDeviceToHost));
    if (parts) HIPCHECK(ix, hipMemcpy(parts, ix->ws_parts.p, (size_t)nq * L * 4, hipMemcpyDeviceToHost));
    ix->last_valid = false;
    return DPF_OK;
}

// The Sets of a batch, left on the device: run_probe's per-query lists packed back to back in ws_compact, their
// offsets in ws_misc (device) and sets_off (host), the padded query rows in ws_hq / ws_qpad.  Any later call that
// reuses those workspaces clears sets_valid.
static int sets_to_device(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps, int64_t *cand_offsets)
{
    ix->sets_valid = false;
    cand_offsets[0] = 0;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    const size_t qbytes = (size_t)nq * ix->cfg.dim * ix->elem;
    const size_t at_ids = (qbytes + 255) & ~(size_t)255;
    HIPCHECK(ix, ix->h_stage.ensure(at_ids + (size_t)nq * 4 + 256));
    HIPCHECK(ix, ix->ws_hq.ensure(qbytes));
    int rc;
    if ((rc = stage_to_device(ix, ix->ws_hq.p, Q, qbytes, 0, nullptr)) != DPF_OK) return rc;
    const int32_t *dq = nullptr;
    if (qids) {
        HIPCHECK(ix, ix->ws_hqids.ensure((size_t)nq * 4));
        if ((rc = stage_to_device(ix, ix->ws_hqids.p, qids, (size_t)nq * 4, at_ids, nullptr)) != DPF_OK) return rc;
        dq = ix->ws_hqids.as<int32_t>();
    }
    // (the rerank kernels read whole 16-byte pieces: the Sets keep padded query rows for dpf_sets_topk)
    const void *qpad; int64_t qstride;
    if ((rc = pad_queries(ix, ix->ws_hq.p, nq, nullptr, &qpad, &qstride)) != DPF_OK) return rc;
    if ((rc = run_probe(ix, dq, qpad, qstride, nq, steps, nullptr)) != DPF_OK) return rc;
    // compact the per-query sets into one CSR array
    HIPCHECK(ix, ix->ws_misc.ensure((size_t)(nq + 1) * 8));
    launch_scan_i32(ix->ws_cnt.as<int32_t>(), nq, ix->ws_misc.as<int64_t>(), nullptr);
    HIPCHECK(ix, hipMemcpy(cand_offsets, ix->ws_misc.p, (size_t)(nq + 1) * 8, hipMemcpyDeviceToHost));
    const int64_t total = cand_offsets[nq];
    if (total > 0) {
        HIPCHECK(ix, ix->ws_compact.ensure((size_t)total * 4));
        launch_compact(ix->ws_arena.as<int32_t>(), ix->ws_off.as<int64_t>(), ix->ws_cnt.as<int32_t>(), ix->ws_misc.as<int64_t>(),
                       ix->ws_compact.as<int32_t>(), nq, nullptr);
        HIPCHECK(ix, hipGetLastError());
    }
    ix->sets_off.assign(cand_offsets, cand_offsets + nq + 1);
    ix->sets_qpad = qpad;
    ix->sets_valid = true;
    return DPF_OK;
}

static int sets_entry_checks(dpf_index *ix, const void *Q, int64_t nq, int32_t steps, int64_t *cand_offsets, const char *who, bool *empty)
{
    *empty = false;
    int rc = check_ready(ix, true);
    if (rc == DPF_ENOTFIT && cand_offsets && nq >= 0) {
        // Dense:422-424: an unfitted index yields empty sets
        for (int64_t i = 0; i <= nq; i++) cand_offsets[i] = 0;
        return rc;
    }
    if (rc != DPF_OK) return rc;
    if (nq < 0 || steps < 0 || !cand_offsets || (nq > 0 && !Q)) return fail(ix, DPF_EINVAL, std::string(who) + ": bad arguments");
    cand_offsets[0] = 0;
    *empty = nq == 0;
    return DPF_OK;
}

extern "C" int dpf_query_batch_resident(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps,
                                        int64_t *cand_offsets)
{
    if (!ix) return DPF_EINVAL;
    bool empty;
    int rc = sets_entry_checks(ix, Q, nq, steps, cand_offsets, "dpf_query_batch_resident", &empty);
    if (rc != DPF_OK) return rc;
    if (empty) { ix->sets_off.assign(1, 0); ix->sets_valid = true; return DPF_OK; }
    return sets_to_device(ix, qids, Q, nq, steps, cand_offsets);
}

extern "C" int dpf_sets_fetch(dpf_index *ix, int64_t q0, int64_t q1, int32_t *out)
{
    if (!ix) return DPF_EINVAL;
    if (!ix->sets_valid) return fail(ix, DPF_EINVAL, "dpf_sets_fetch: no resident Sets (dpf_query_batch_resident first; any other query or fit call drops them)");
    const int64_t nq = (int64_t)ix->sets_off.size() - 1;
    if (q0 < 0 || q1 < q0 || q1 > nq) return fail(ix, DPF_EINVAL, "dpf_sets_fetch: query range outside the batch");
    const int64_t a = ix->sets_off[(size_t)q0], b = ix->sets_off[(size_t)q1];
    if (b == a) return DPF_OK;
    if (!out) return fail(ix, DPF_EINVAL, "dpf_sets_fetch: null output");
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    HIPCHECK(ix, hipMemcpy(out, ix->ws_compact.as<int32_t>() + a, (size_t)(b - a) * 4, hipMemcpyDeviceToHost));
    return DPF_OK;
}

extern "C" int64_t dpf_sets_queries(const dpf_index *ix)
{
    return ix && ix->sets_valid ? (int64_t)ix->sets_off.size() - 1 : -1;
}

extern "C" int64_t dpf_sets_count(const dpf_index *ix, int64_t q0, int64_t q1)
{
    if (!ix || !ix->sets_valid) return -1;
    const int64_t nq = (int64_t)ix->sets_off.size() - 1;
    if (q0 < 0 || q1 < q0 || q1 > nq) return -1;
    return ix->sets_off[(size_t)q1] - ix->sets_off[(size_t)q0];
}

extern "C" int dpf_sets_topk(dpf_index *ix, int32_t k, int32_t *ids, double *scores)
{
    if (!ix) return DPF_EINVAL;
    if (!ix->sets_valid) return fail(ix, DPF_EINVAL, "dpf_sets_topk: no resident Sets (dpf_query_batch_resident first; any other query or fit call drops them)");
    const int64_t nq = (int64_t)ix->sets_off.size() - 1;
    if (k < 1 || k > 64 || (nq > 0 && (!ids || !scores))) return fail(ix, DPF_EINVAL, "dpf_sets_topk: bad arguments");
    if (nq == 0) return DPF_OK;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    HIPCHECK(ix, ix->ws_hids.ensure((size_t)nq * k * 4));
    HIPCHECK(ix, ix->ws_hscores.ensure((size_t)nq * k * 8));
    RerankParams rp;
    fill_rerank_params(ix, rp, ix->sets_qpad, k, ix->ws_misc.as<int64_t>(), nullptr, ix->ws_compact.as<int32_t>(), ix->ws_hids.as<int32_t>(),
                       ix->ws_hscores.as<double>());
    launch_rerank(ix->cfg.dtype, ix->cfg.metric, rp, nq, nullptr);
    HIPCHECK(ix, hipGetLastError());
    HIPCHECK(ix, hipMemcpy(ids, ix->ws_hids.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    HIPCHECK(ix, hipMemcpy(scores, ix->ws_hscores.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    return DPF_OK;
}

// The Sets in the handle's own pinned buffer (grown as needed, reused by the handle's next call).
extern "C" int dpf_query_batch_view(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps,
                                    int64_t *cand_offsets, const int32_t **cand_ids)
{
    if (!ix) return DPF_EINVAL;
    if (cand_ids) *cand_ids = nullptr;
    bool empty;
    int rc = sets_entry_checks(ix, Q, nq, steps, cand_offsets, "dpf_query_batch_view", &empty);
    if (rc != DPF_OK) return rc;
    if (!cand_ids) return fail(ix, DPF_EINVAL, "dpf_query_batch_view: bad arguments");
    static const int32_t none = 0;
    if (empty) { *cand_ids = &none; return DPF_OK; }
    if ((rc = sets_to_device(ix, qids, Q, nq, steps, cand_offsets)) != DPF_OK) return rc;
    const int64_t total = cand_offsets[nq];
    // (at c3 a batch's Sets are 2.9 GB of ids: into pinned memory they arrive at the link's rate, into fresh pageable
    // memory at a tenth of it)
    const size_t bytes = (size_t)std::max<int64_t>(total, 1) * 4;
    if (bytes > ix->h_sets_cap) {
        if (ix->h_sets) { (void)hipHostFree(ix->h_sets); ix->h_sets = nullptr; ix->h_sets_cap = 0; }
        void *q = nullptr;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ix, DPF_ENOMEM, "dpf_query_batch_view: no pinned host memory for " + std::to_string(bytes) + " bytes of lists (use dpf_query_batch_resident + dpf_sets_fetch in pieces)");
        }
        ix->h_sets = (int32_t *)q; ix->h_sets_cap = want;
    }
    if (total > 0) HIPCHECK(ix, hipMemcpy(ix->h_sets, ix->ws_compact.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    *cand_ids = ix->h_sets;
    return DPF_OK;
}

// The Sets in a block the caller owns (dpf_free).
extern "C" int dpf_query_batch(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps,
                               int64_t *cand_offsets, int32_t **cand_ids)
{
    if (!ix) return DPF_EINVAL;
    if (cand_ids) *cand_ids = nullptr;
    bool empty;
    int rc = sets_entry_checks(ix, Q, nq, steps, cand_offsets, "dpf_query_batch", &empty);
    if (rc != DPF_OK) return rc;
    if (!cand_ids) return fail(ix, DPF_EINVAL, "dpf_query_batch: bad arguments");
    if (empty) { *cand_ids = (int32_t *)malloc(4); return *cand_ids ? DPF_OK : fail(ix, DPF_ENOMEM, "dpf_query_batch: out of host memory"); }
    if ((rc = sets_to_device(ix, qids, Q, nq, steps, cand_offsets)) != DPF_OK) return rc;
    const int64_t total = cand_offsets[nq];
    int32_t *host = (int32_t *)malloc((size_t)std::max<int64_t>(total, 1) * 4);
    if (!host) return fail(ix, DPF_ENOMEM, "dpf_query_batch: out of host memory");
    // through the handle's pinned staging area, two pieces in flight: the copy engine fills one while the CPU empties the other
    const size_t piece = (size_t)8 << 20, bytes = (size_t)total * 4, pieces = (bytes + piece - 1) / piece;
    hipError_t e = total > 0 ? ix->h_stage.ensure(2 * piece) : hipSuccess;
    for (size_t i = 0; e == hipSuccess && i < pieces + 1; i++) {
        if (i < pieces) {                                 // (piece i - 2, which had this half, was emptied one round ago)
            e = hipMemcpyAsync((char *)ix->h_stage.p + (i % 2) * piece, (const char *)ix->ws_compact.p + i * piece,
                               std::min(piece, bytes - i * piece), hipMemcpyDeviceToHost, nullptr);
            if (e == hipSuccess) e = hipEventRecord(ix->ev_stage[i % 2], nullptr);
        }
        if (e == hipSuccess && i >= 1) {
            e = hipEventSynchronize(ix->ev_stage[(i - 1) % 2]);
            if (e == hipSuccess) memcpy((char *)host + (i - 1) * piece, (const char *)ix->h_stage.p + ((i - 1) % 2) * piece, std::min(piece, bytes - (i - 1) * piece));
        }
    }
    if (e != hipSuccess) { (void)hipDeviceSynchronize(); free(host); ix->err = std::string("dpf_query_batch: ") + hipGetErrorString(e); return DPF_EHIP; }
    *cand_ids = host;
    return DPF_OK;
}

extern "C" void dpf_free(void *p)
{
    free(p);
}

extern "C" int dpf_rerank(dpf_index *ix, const void *Q, int64_t nq, const int64_t *cand_offsets,
                          const int32_t *cand_ids, int32_t k, int32_t *ids, double *scores)
{
    if (!ix) return DPF_EINVAL;
    int rc = check_ready(ix, true);
    if (rc != DPF_OK) return rc;
    if (nq < 0 || k < 1 || k > 64 || (nq > 0 && (!Q || !cand_offsets || !ids || !scores))) return fail(ix, DPF_EINVAL, "dpf_rerank: bad arguments");
    if (nq == 0) return DPF_OK;
    const int64_t total = cand_offsets[nq];
    if (total < 0 || (total > 0 && !cand_ids)) return fail(ix, DPF_EINVAL, "dpf_rerank: bad candidate lists");
    ix->sets_valid = false;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    const size_t qbytes = (size_t)nq * ix->cfg.dim * ix->elem;
    HIPCHECK(ix, ix->ws_hq.ensure(qbytes));
    HIPCHECK(ix, ix->ws_hids.ensure((size_t)nq * k * 4));
    HIPCHECK(ix, ix->ws_hscores.ensure((size_t)nq * k * 8));
    HIPCHECK(ix, ix->ws_off.ensure((size_t)(nq + 1) * 8));
    HIPCHECK(ix, ix->ws_arena.ensure((size_t)std::max<int64_t>(total, 1) * 4));
    HIPCHECK(ix, hipMemcpy(ix->ws_hq.p, Q, qbytes, hipMemcpyHostToDevice));
    HIPCHECK(ix, hipMemcpy(ix->ws_off.p, cand_offsets, (size_t)(nq + 1) * 8, hipMemcpyHostToDevice));
    if (total > 0) HIPCHECK(ix, hipMemcpy(ix->ws_arena.p, cand_ids, (size_t)total * 4, hipMemcpyHostToDevice));
    const void *qpad; int64_t qstride;
    if ((rc = pad_queries(ix, ix->ws_hq.p, nq, nullptr, &qpad, &qstride)) != DPF_OK) return rc;
    RerankParams rp;
    fill_rerank_params(ix, rp, qpad, k, ix->ws_off.as<int64_t>(), nullptr, ix->ws_arena.as<int32_t>(), ix->ws_hids.as<int32_t>(), ix->ws_hscores.as<double>());
    launch_rerank(ix->cfg.dtype, ix->cfg.metric, rp, nq, nullptr);
    HIPCHECK(ix, hipGetLastError());
    HIPCHECK(ix, hipMemcpy(ids, ix->ws_hids.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    HIPCHECK(ix, hipMemcpy(scores, ix->ws_hscores.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    ix->last_valid = false;
    return DPF_OK;
}

extern "C" int dpf_merge_topk_device(dpf_index *ix, const int32_t *in_ids, const double *in_scores, int32_t parts,
                                     int64_t nq, int32_t k, int32_t *d_ids, double *d_scores, void *stream)
{
    if (!ix) return DPF_EINVAL;
    if (parts < 1 || parts > 64 || k < 1 || nq < 0 || !in_ids || !in_scores || !d_ids || !d_scores)
        return fail(ix, DPF_EINVAL, "dpf_merge_topk_device: bad arguments");
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    launch_merge_topk(ix->cfg.metric, in_ids, in_scores, parts, nq * k, nq * k, nq, k, d_ids, d_scores, (hipStream_t)stream);
    HIPCHECK(ix, hipGetLastError());
    return DPF_OK;
}

extern "C" int dpf_merge_topk_packed_device(dpf_index *ix, const void *in, int64_t part_bytes, int32_t parts, int64_t nq, int32_t k,
                                            int32_t *d_ids, double *d_scores, void *stream)
{
    if (!ix) return DPF_EINVAL;
    if (parts < 1 || parts > 64 || k < 1 || nq < 0 || !in || !d_ids || !d_scores || part_bytes % 8 != 0 ||
        part_bytes < nq * k * 12 || ((uintptr_t)in & 7) != 0)
        return fail(ix, DPF_EINVAL, "dpf_merge_topk_packed_device: bad arguments");
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    const double *scores = (const double *)in;
    const int32_t *ids = (const int32_t *)((const char *)in + (size_t)nq * k * 8);
    launch_merge_topk(ix->cfg.metric, ids, scores, parts, part_bytes / 4, part_bytes / 8, nq, k, d_ids, d_scores, (hipStream_t)stream);
    HIPCHECK(ix, hipGetLastError());
    return DPF_OK;
}

extern "C" int dpf_set_option(dpf_index *ix, int32_t option, int64_t value)
{
    if (!ix) return DPF_EINVAL;
    switch (option) {
    case DPF_OPT_EMIT_CAP:
        if (value < 0 || value > 65536) return fail(ix, DPF_EINVAL, "dpf_set_option: forward-list capacity must be in [0, 65536]");
        ix->opt_emit_cap = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_BUCKET_LIST_CAP:
        if (value < 0 || value > 4096) return fail(ix, DPF_EINVAL, "dpf_set_option: bucket-list capacity must be in [0, 4096]");
        ix->opt_bucket_list_cap = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_SYNC_SIZES:
        ix->opt_sync_sizes = value != 0;
        return DPF_OK;
    case DPF_OPT_FIRST_PASS_PERMILLE:
        if (value < 0 || value > 1000) return fail(ix, DPF_EINVAL, "dpf_set_option: first-pass share must be in [0, 1000]");
        ix->opt_first_permille = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_SCAN_PASSES:
        if (value != 0 && value != 2 && value != 3) return fail(ix, DPF_EINVAL, "dpf_set_option: scan passes must be 0 (by row length), 2 or 3");
        ix->opt_scan_passes = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_EARLY_PASS_PERMILLE:
        if (value < 0 || value > 1000) return fail(ix, DPF_EINVAL, "dpf_set_option: early-pass share must be in [0, 1000]");
        ix->opt_early_permille = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_SEED_SAMPLE:
        if (value != 0 && value != -1 && (value < 16 || value > 4096))
            return fail(ix, DPF_EINVAL, "dpf_set_option: seed sample must be -1 (by bucket sizes), 0 (none) or in [16, 4096]");
        ix->opt_seed_cap = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_SHORT_RERANK:
        ix->opt_short_rerank = value != 0;
        return DPF_OK;
    case DPF_OPT_KEEP_FORWARDED:
        ix->opt_keep_forwarded = value != 0;
        return DPF_OK;
    case DPF_OPT_SIDE_STREAM:
        if (value < -1 || value > 2) return fail(ix, DPF_EINVAL, "dpf_set_option: side stream must be -1 (by the forest), 0, 1 or 2");
        ix->opt_side_stream = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_EXACT_SEEDS:
        ix->opt_exact_seeds = value != 0;
        return DPF_OK;
    case DPF_OPT_DEFER_BUILD:
        ix->opt_defer_build = value != 0;
        return DPF_OK;
    case DPF_OPT_EXPECTED_ROWS:
        if (value < 0) return fail(ix, DPF_EINVAL, "dpf_set_option: the row count must not be negative");
        ix->opt_expected_rows = value;
        return DPF_OK;
    case DPF_OPT_LIST_BINS:
        if (value < 0 || value > 2) return fail(ix, DPF_EINVAL, "dpf_set_option: list bins must be 0, 1 or 2");
        ix->opt_list_bins = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_PROBE_LANES:
        if (value < -1 || value > 1) return fail(ix, DPF_EINVAL, "dpf_set_option: probe lanes must be -1 (by the forest), 0 or 1");
        ix->opt_probe_lanes = (int32_t)value;
        return DPF_OK;
    case DPF_OPT_PROJECTION_MIN_ROWS:
        if (value < 0) return fail(ix, DPF_EINVAL, "dpf_set_option: the row count must not be negative");
        ix->opt_projection_min_rows = value;
        return DPF_OK;
    default:
        return fail(ix, DPF_EINVAL, "dpf_set_option: unknown option");
    }
}

extern "C" int dpf_set_probe_mode(dpf_index *ix, int32_t mode)
{
    if (!ix) return DPF_EINVAL;
    if (mode != DPF_PROBE_FLIPS && mode != DPF_PROBE_OWN) return fail(ix, DPF_EINVAL, "dpf_set_probe_mode: unknown mode");
    ix->probe_mode = mode;
    return DPF_OK;
}

extern "C" int dpf_set_profiling(dpf_index *ix, int32_t on)
{
    if (!ix) return DPF_EINVAL;
    ix->profiling = on != 0;
    return DPF_OK;
}

static int last_totals(dpf_index *ix, int64_t *raw_total, int64_t *uniq_total)
{
    *raw_total = 0; *uniq_total = 0;
    if (!ix->last_valid || ix->last_nq <= 0) return DPF_OK;
    std::vector<unsigned long long> raw((size_t)ix->last_nq);
    std::vector<int32_t> cnt((size_t)ix->last_nq);
    HIPCHECK(ix, hipMemcpy(raw.data(), ix->ws_raw.p, raw.size() * 8, hipMemcpyDeviceToHost));
    // gather path: per-query set sizes; bucket-major path: candidates forwarded to the exact rerank
    HIPCHECK(ix, hipMemcpy(cnt.data(), ix->last_path == 1 ? (void *)emit_cnt_of(ix) : ix->ws_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
    for (auto v : raw) *raw_total += (int64_t)v;
    for (auto v : cnt) *uniq_total += v;
    return DPF_OK;
}

extern "C" int dpf_get_stats(dpf_index *ix, dpf_stats *out)
{
    if (!ix || !out) return DPF_EINVAL;
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    HIPCHECK(ix, hipDeviceSynchronize());
    unsigned long long sh = 0;
    HIPCHECK(ix, hipMemcpy(&sh, ix->d_short_hash, 8, hipMemcpyDeviceToHost));
    ix->st.short_hash_events = (int64_t)sh;
    {
        unsigned long long und = 0;
        HIPCHECK(ix, hipMemcpy(&und, ix->d_proj_counters + 8, 8, hipMemcpyDeviceToHost));
        ix->st.hash_undecided_rows = (int64_t)und;
        HIPCHECK(ix, hipMemcpy(&und, ix->d_proj_counters + 16, 8, hipMemcpyDeviceToHost));
        ix->st.debug_bound_violations = (int64_t)und;
    }
    int rc = last_totals(ix, &ix->st.last_raw_candidates, &ix->st.last_unique_candidates);
    if (rc != DPF_OK) return rc;
    ix->st.last_path = ix->last_valid ? ix->last_path : 0;
    ix->st.last_forwarded = 0;
    ix->st.last_ms_scan = 0;
    ix->st.last_scan_launches = 0;
    if (ix->last_valid && ix->last_path == 1) {
        ix->st.last_forwarded = ix->st.last_unique_candidates;
        ix->st.last_unique_candidates = 0;          // the sets are not materialised on this path
        float sum = 0;
        bool ok = ix->last_timed;
        for (int s2 = 0; s2 < 3 && ok; s2++) {
            float t = 0;
            if (!ix->ev_scan_used[s2]) continue;
            ok = hipEventElapsedTime(&t, ix->ev_scan[2 * s2], ix->ev_scan[2 * s2 + 1]) == hipSuccess;
            sum += t;
        }
        if (ok) ix->st.last_ms_scan = sum;
        ix->st.last_scan_launches = ix->scan_launches;
    }
    if (ix->last_timed && ix->last_valid) {
        float ms = 0;
        ix->st.last_ms_dedupe = 0;
        if (hipEventElapsedTime(&ms, ix->ev[0], ix->ev[1]) == hipSuccess) ix->st.last_ms_hash = ms;
        if (hipEventElapsedTime(&ms, ix->ev[1], ix->ev[2]) == hipSuccess) ix->st.last_ms_probe = ms;
        if (ix->last_path == 0 && hipEventElapsedTime(&ms, ix->ev[2], ix->ev[3]) == hipSuccess) ix->st.last_ms_dedupe = ms;
        if (ix->last_k > 0 && hipEventElapsedTime(&ms, ix->ev[4], ix->ev[5]) == hipSuccess) ix->st.last_ms_rerank = ms;
        (void)hipGetLastError();      // an unrecorded event is not an error of this call
    }
    ix->st.base16_bytes = ix->bucket_major && ix->ks16 > 0 ? ix->base_rows * (int64_t)ix->ks16 * 64 : 0;
    ix->st.base8_bytes = ix->i8_ok ? ix->base_rows * (int64_t)ix->ks8 * 64 : 0;
    ix->st.last_scan_int8 = ix->last_valid && ix->last_path == 1 && ix->last_scan_int8 ? 1 : 0;
    ix->st.tile_rows = ix->bucket_major && ix->ks16 > 0 ? leafscan_tile_rows(ix->ks16) : 0;
    ix->st.num_tiles = ix->ntiles;
    ix->st.resident_ids = 0;
    for (uint32_t len : ix->h_leaf_len) ix->st.resident_ids += len;
    ix->st.bucket_order_bytes = ix->slot16_ok ? ix->st.resident_ids * (int64_t)ix->ks16 * 64 : 0;
    *out = ix->st;
    return DPF_OK;
}

extern "C" int dpf_last_incidences(dpf_index *ix, uint32_t *bucket_counts, uint64_t *ids_visited, int32_t *buckets_reached,
                                   uint32_t *bucket_lists, int32_t list_cap)
{
    if (!ix) return DPF_EINVAL;
    if (!ix->last_valid || ix->last_path != 1 || ix->last_nq <= 0) return fail(ix, DPF_EINVAL, "dpf_last_incidences: the last batch did not take the bucket-major path");
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    HIPCHECK(ix, hipDeviceSynchronize());
    if (bucket_counts && ix->num_leaves > 0) HIPCHECK(ix, hipMemcpy(bucket_counts, lq_cnt_of(ix), (size_t)ix->num_leaves * 4, hipMemcpyDeviceToHost));
    if (ids_visited) HIPCHECK(ix, hipMemcpy(ids_visited, ix->ws_raw.p, (size_t)ix->last_nq * 8, hipMemcpyDeviceToHost));
    if (buckets_reached) HIPCHECK(ix, hipMemcpy(buckets_reached, ix->ws_q_leaf_cnt.p, (size_t)ix->last_nq * 4, hipMemcpyDeviceToHost));
    if (bucket_lists && list_cap > 0) {
        const int32_t cap = ix->last_q_leaf_cap;
        std::vector<uint2> row((size_t)std::max(cap, 1));
        std::vector<int32_t> cnt((size_t)ix->last_nq);
        HIPCHECK(ix, hipMemcpy(cnt.data(), ix->ws_q_leaf_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
        for (int64_t q = 0; q < ix->last_nq; q++) {
            const int32_t n = std::min(std::min(cnt[(size_t)q], cap), list_cap);
            if (n > 0) HIPCHECK(ix, hipMemcpy(row.data(), ix->ws_q_leaves.as<uint2>() + (size_t)q * cap, (size_t)n * 8, hipMemcpyDeviceToHost));
            for (int32_t i = 0; i < list_cap; i++) bucket_lists[(size_t)q * list_cap + i] = i < n ? row[(size_t)i].x : 0xffffffffu;
        }
    }
    return DPF_OK;
}

extern "C" int dpf_last_forward_counts(dpf_index *ix, int32_t *forwarded, int32_t *kept)
{
    if (!ix) return DPF_EINVAL;
    if (!ix->last_valid || ix->last_path != 1 || ix->last_nq <= 0) return fail(ix, DPF_EINVAL, "dpf_last_forward_counts: the last batch did not take the bucket-major path");
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    HIPCHECK(ix, hipDeviceSynchronize());
    if (forwarded) HIPCHECK(ix, hipMemcpy(forwarded, emit_cnt_of(ix), (size_t)ix->last_nq * 4, hipMemcpyDeviceToHost));
    if (kept && ix->ws_emit_ucnt.p) HIPCHECK(ix, hipMemcpy(kept, ix->ws_emit_ucnt.p, (size_t)ix->last_nq * 4, hipMemcpyDeviceToHost));
    return DPF_OK;
}

extern "C" int dpf_last_rerank_profile(dpf_index *ix, double *ms, int64_t *algorithmic_bytes)
{
    if (!ix) return DPF_EINVAL;
    if (!ix->last_valid || ix->last_k <= 0) return fail(ix, DPF_EINVAL, "dpf_last_rerank_profile: no top-k batch has run");
    if (!ix->last_timed) return fail(ix, DPF_EINVAL, "dpf_last_rerank_profile: the last batch ran without dpf_set_profiling(1)");
    HIPCHECK(ix, hipSetDevice(ix->cfg.device));
    HIPCHECK(ix, hipEventSynchronize(ix->ev[5]));
    float t = 0;
    if (ix->last_path == 1) {
        for (int s2 = 0; s2 < 3; s2++) {
            float t2 = 0;
            if (!ix->ev_scan_used[s2]) continue;
            HIPCHECK(ix, hipEventElapsedTime(&t2, ix->ev_scan[2 * s2], ix->ev_scan[2 * s2 + 1]));
            t += t2;
        }
    } else {
        HIPCHECK(ix, hipEventElapsedTime(&t, ix->ev[4], ix->ev[5]));
    }
    int64_t raw = 0, uniq = 0;
    int rc = last_totals(ix, &raw, &uniq);
    if (rc != DPF_OK) return rc;
    if (ms) *ms = t;
    const int64_t row = (int64_t)ix->cfg.dim * (int64_t)ix->elem;
    if (ix->last_path == 1) {
        // bucket-major scan: every touched bucket's rows with their ids and norms once; one query row,
        // its index, threshold and 16-byte record per (query, bucket) incidence; one (id, bound) pair
        // per forwarded candidate
        std::vector<uint32_t> c((size_t)ix->num_leaves);
        HIPCHECK(ix, hipMemcpy(c.data(), lq_cnt_of(ix), c.size() * 4, hipMemcpyDeviceToHost));
        int64_t touched_rows = 0;
        for (size_t l = 0; l < c.size(); l++)
            if (c[l]) touched_rows += ix->h_leaf_len[l];
        const int64_t srow = ix->last_scan_int8 ? (int64_t)ix->ks8 * 64 : (int64_t)ix->ks16 * 64;
        const int64_t tail = !ix->last_scan_int8 && ix->filter_dim < ix->cfg.dim ? 16 : 0;   // the 16-byte tails of cut rows and query rows
        if (algorithmic_bytes) *algorithmic_bytes = touched_rows * (srow + 8 + tail) + ix->st.last_pairs * (srow + 24 + tail) + uniq * 8;
        return DPF_OK;
    }
    // gather rerank: candidate rows + candidate ids + query rows + (score, id) results
    if (algorithmic_bytes) *algorithmic_bytes = uniq * (row + 4) + ix->last_nq * row + ix->last_nq * (int64_t)ix->last_k * 12;
    return DPF_OK;
}
