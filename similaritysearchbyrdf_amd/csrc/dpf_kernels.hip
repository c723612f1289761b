// gfx950 kernels of the Dynamic Partition Forest batch-query path.
//
//   k_hash        LSH projection + hash assembly + sub-index id      (fp64 VALU, exact order)
//   k_probe       multi-probe / multi-step trie walk; counts or sets candidate bits
//   k_scan_*      exclusive scan (per-query candidate bounds, per-bucket query counts)
//   k_rerank      candidate-row gather + distance + per-query top-k  (HBM-bound)
//   k_merge_topk  merge of partial top-k lists
//   k_compact     gathers the per-query sets into one CSR array
//
// Reference citations (paths relative to the reference checkout):
//   Angle   src/main/scala/mclab/lsh/hashFamilies/AngleHashFamily.scala
//   PStable src/main/scala/mclab/lsh/hashFamilies/PStableHashFamily.scala
//   Sim     src/main/scala/mclab/lsh/vector/SimilarityCalculator.scala
//   Part    src/main/scala/mclab/utils/Partitioner.scala
//   RDT     src/main/java/mclab/mapdb/RandomDrawTreeMap.java
//   Dense   src/main/scala/mclab/deploy/DensevectorRDFInit.scala
//
// The whole file is compiled with -ffp-contract=off: the hash bits must match a
// JVM that never fuses a multiply with an add (Sim:46-47).  Where a fused
// multiply-add is wanted (rerank) it is written as fma().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "dpf_device.hpp"
#include "dpf_hash_common.hpp"

#pragma clang fp contract(off)

#ifndef LS16_WAVES4
#define LS16_WAVES4 3
#endif

namespace dpf {

// ---------------------------------------------------------------------------
// k_hash
// ---------------------------------------------------------------------------
// One workgroup hashes a tile of G*QT vectors against all F pool functions,
// then assembles the L table hashes and sub-index ids of each vector.
//
// Thread (fl, qg): function lane fl in [0, FT), query group qg in [0, G),
// G = 256 / FT.  Each thread carries QT running sums in registers and walks
// the dimension in order, so every dot product is the exact left-to-right
// fp64 sum of Sim:40-49 -- bit-identical hash bits, no guard band needed.
// The pool is read through its transpose fnsT[dim][F] (coalesced across fl,
// L2-resident); the vector tile sits in LDS and is read as a broadcast.
// With row_list the kernel hashes the rows row_list[0 .. *row_count) instead of rows 0 .. n-1 (the rows the
// matrix-core projection could not decide, see dpf_project.hip); the grid covers the worst case and blocks past
// the count leave at once.
template <typename T, int QT, bool LIST>
__global__ __launch_bounds__(256) void k_hash(const T *__restrict__ X, int64_t x_stride, int64_t n,
                                              HashParams p, int FT,
                                              int32_t *__restrict__ hashes, int32_t *__restrict__ parts,
                                              const int32_t *__restrict__ row_list, const uint32_t *__restrict__ row_count)
{
    extern __shared__ double sm[];
    const int G = 256 / FT;
    const int tile = G * QT;
    double *xs = sm;                          // [tile][dim]
    double *dots = sm + (size_t)tile * p.dim; // [tile][F]
    const int tid = threadIdx.x;
    if (LIST) n = (int64_t)*row_count < n ? (int64_t)*row_count : n;
    auto row_of = [&](int64_t at) { return LIST ? (int64_t)row_list[at] : at; };
    // (a list is worked through by a small grid in strides; the plain form has one tile per workgroup)
    for (int64_t q0 = (int64_t)blockIdx.x * tile; q0 < n; q0 += (int64_t)gridDim.x * tile) {
    if (LIST) __syncthreads();

    for (int idx = tid; idx < tile * p.dim; idx += 256) {
        int q = idx / p.dim, i = idx - q * p.dim;
        xs[idx] = q0 + q < n ? (double)X[row_of(q0 + q) * x_stride + i] : 0.0;
    }
    __syncthreads();

    const int fl = tid % FT, qg = tid / FT;
    const double *xq = xs + (size_t)qg * QT * p.dim;
    for (int f0 = 0; f0 < p.F; f0 += FT) {
        const int f = f0 + fl;
        if (f < p.F) {
            double acc[QT];
#pragma unroll
            for (int j = 0; j < QT; j++) acc[j] = 0.0;
            const double *col = p.fnsT + f;
            for (int i = 0; i < p.dim; i++) {
                const double a = col[(size_t)i * p.F];
#pragma unroll
                for (int j = 0; j < QT; j++) {
                    const double prod = a * xq[j * p.dim + i];
                    acc[j] = acc[j] + prod;
                }
            }
#pragma unroll
            for (int j = 0; j < QT; j++) dots[(size_t)(qg * QT + j) * p.F + f] = acc[j];
        }
    }
    __syncthreads();

    for (int idx = tid; idx < tile * p.L; idx += 256) {
        const int q = idx / p.L, t = idx - q * p.L;
        if (q0 + q >= n) continue;
        const int64_t row = row_of(q0 + q);
        const double *dq = dots + (size_t)q * p.F;
        int32_t h, part;
        assemble_hash(p, t,
                      [&](int f) { return dq[f] <= 0 ? 0u : 1u; },                                  // Angle:184
                      [&](int f) { double v = dq[f] + p.b[f]; v = v / (double)p.w[f]; return d2i_sat(v); },   // PStable:130
                      h, part);
        hashes[row * p.L + t] = h;
        if (parts) parts[row * p.L + t] = part;
    }
    if (!LIST) break;
    }
}

void launch_hash_rows(int dtype, const void *X, int64_t x_stride, int64_t n, const HashParams &p, int32_t *hashes, int32_t *parts,
                      const int32_t *row_list, const uint32_t *row_count, hipStream_t stream)
{
    if (n <= 0) return;
    int FT = p.F <= 64 ? 64 : (p.F <= 128 ? 128 : 256);
    const int QT = 4;
    int tile = (256 / FT) * QT;
    size_t shmem = (size_t)tile * (p.dim + p.F) * sizeof(double);
    unsigned grid = (unsigned)((n + tile - 1) / tile);
    if (row_list) grid = std::min(grid, 512u);             // usually (almost) empty: keep the launch cheap
#define DPF_HASH(T, LIST)                                                                                          \
    hipLaunchKernelGGL((k_hash<T, QT, LIST>), dim3(grid), dim3(256), shmem, stream, (const T *)X, x_stride, n, p, FT, \
                       hashes, parts, row_list, row_count)
    if (dtype == 0) { if (row_list) DPF_HASH(float, true); else DPF_HASH(float, false); }
    else { if (row_list) DPF_HASH(double, true); else DPF_HASH(double, false); }
#undef DPF_HASH
}

void launch_hash(int dtype, const void *X, int64_t x_stride, int64_t n, const HashParams &p,
                 int32_t *hashes, int32_t *parts, hipStream_t stream)
{
    launch_hash_rows(dtype, X, x_stride, n, p, hashes, parts, nullptr, nullptr, stream);
}

size_t hash_shmem_bytes(const HashParams &p)
{
    int FT = p.F <= 64 ? 64 : (p.F <= 128 ? 128 : 256);
    int tile = (256 / FT) * 4;
    return (size_t)tile * (p.dim + p.F) * sizeof(double);
}

// ---------------------------------------------------------------------------
// k_probe
// ---------------------------------------------------------------------------

// RDT:946-992 on the flattened trie: <= max_level+1 dependent 4-byte reads.
__device__ inline uint32_t walk(const uint32_t *__restrict__ dnodes, uint32_t root, uint32_t pro,
                                int max_level, int nbits, int dns)
{
    uint32_t node = root;
    if (node == 0) return NO_LEAF;
    for (int level = max_level; level >= 0; --level) {
        const uint32_t slot = (pro >> (level * nbits)) & (uint32_t)(dns - 1);
        const uint32_t w = dnodes[(size_t)node * dns + slot];
        if (w == 0) return NO_LEAF;
        if (w & 1u) return w >> 1;
        node = w >> 1;
    }
    return NO_LEAF;
}

// RDT:982: candidate == query id is dropped only for ids in the boxed-Integer
// cache [-128, 127] (reference inequality on java.lang.Integer).
__device__ inline bool self_excluded(int32_t cand, int32_t qid, int mode)
{
    if (cand != qid) return false;
    if (mode == 1) return true;
    if (mode == 2) return false;
    return qid >= -128 && qid <= 127;
}

// One workgroup per query.  Waves take tables round-robin.  For one table a
// wave enumerates the sub-indexes within `steps` bit flips of the query's own
// (RDT:613-621), two per round: lanes 0-31 walk the <= 28 single-bit-flip probes
// (RDT:753-756; the unflipped hash itself is never probed) of the first, lanes
// 32-63 of the second.  Probes that end in the same bucket are collapsed with
// shuffles (the reference concatenates duplicates and takes a Set later,
// Dense:420), then the wave streams each distinct bucket's ids.
//
// MODE 0: count ids only (raw[q], used to size the candidate arena); optionally
//         histogram the (query, bucket) incidences per bucket.
// MODE 1: set a bit per id in an LDS bitmap covering [lo, hi), then emit the
//         set bits as the query's sorted unique candidate list.
// MODE 2: append the query to the list of every bucket it reaches, and copy the
//         ids of table 0's buckets into the query's seed list.
// (the 256-thread forms are compiled for eight waves per SIMD: without the hint the kernel's 106 scalar registers allow
// seven, and the walks are latency bound: k_probe<0> 245 -> 221 us on 10k queries over 30 tables)
template <int MODE>
__global__ __launch_bounds__(MODE == 1 ? 1024 : 256, MODE == 1 ? 4 : 8) void k_probe(ProbeParams p)
{
    extern __shared__ uint32_t lds[];
    __shared__ unsigned long long s_raw;
    __shared__ uint32_t s_wave_tot[16];
    __shared__ uint32_t s_emitted;
    __shared__ uint32_t s_seed, s_nleaf;

    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int half = lane >> 5, i = lane & 31;
    const int32_t qid = p.qids ? p.qids[q] : -1;
    const int P = 1 << p.pb;

    if (tid == 0) { s_raw = 0; s_emitted = 0; s_seed = 0; s_nleaf = 0; }

    const int64_t range = p.row_end - p.row_begin;
    const int64_t chunk_bits = (int64_t)p.chunk_words * 32;
    const int nchunks = MODE != 1 ? 1 : (int)((range + chunk_bits - 1) / chunk_bits);
    unsigned long long raw_local = 0;

    for (int c = 0; c < nchunks; c++) {
        const int64_t lo = p.row_begin + (int64_t)c * chunk_bits;
        const int64_t hi = (lo + chunk_bits < p.row_end) ? lo + chunk_bits : p.row_end;
        const int nwords = (int)((hi - lo + 31) >> 5);
        if (MODE == 1) {
            for (int w = tid; w < nwords; w += blockDim.x) lds[w] = 0;
        }
        __syncthreads();

        // A wave works on two tables at a time: each lane then has four independent walks in flight (two
        // sub-indexes of each table), whose dependent 4-byte reads overlap.
        // what modes 1 and 2 do with one lane's bucket (every lane calls it, with NO_LEAF when it has none; mode 0
        // handles a lane's four buckets together, below)
        auto take = [&](uint32_t leaf, bool own_bucket) {
            uint32_t beg = 0, end = 0;
            if (leaf != NO_LEAF) { beg = p.leaf_off[leaf]; end = p.leaf_off[leaf + 1]; }
            if (MODE == 2) {
                if (end > beg) {
                    const uint32_t pos = atomicAdd(&p.lq_cursor[leaf], 1u);
                    p.leaf_queries[p.lq_off[leaf] + pos] = (int32_t)q;
                }
                // seed sample: ids of the first buckets any wave reaches, up to seed_cap
                // (duplicates across tables are harmless: the exact rerank keeps one)
                unsigned long long mask = __ballot(end > beg);
                while (mask && s_seed < (uint32_t)p.seed_cap) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const uint32_t b = __shfl(beg, src), e = __shfl(end, src);
                    for (uint32_t k0 = b; k0 < e; k0 += 64) {
                        const uint32_t k = k0 + lane;
                        const int32_t id = k < e ? p.ids[k] : -1;
                        const bool ok = k < e && id >= p.row_begin && id < p.row_end &&
                                        !self_excluded(id, qid, p.exclude_self);
                        const unsigned long long m = __ballot(ok);
                        if (m == 0) continue;
                        uint32_t at0 = 0;
                        if (lane == 0) at0 = atomicAdd(&s_seed, (uint32_t)__popcll(m));
                        at0 = __shfl(at0, 0);
                        if (at0 >= (uint32_t)p.seed_cap) break;
                        const uint32_t at = at0 + __popcll(m & ((1ull << lane) - 1ull));
                        if (ok && at < (uint32_t)p.seed_cap) p.seed_cand[q * p.seed_cap + at] = id;
                    }
                }
            } else {
                unsigned long long mask = __ballot(leaf != NO_LEAF);
                while (mask) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const uint32_t b = __shfl(beg, src), e = __shfl(end, src);
                    for (uint32_t k = b + lane; k < e; k += 64) {
                        const int32_t id = p.ids[k];
                        if (id >= lo && id < hi && !self_excluded(id, qid, p.exclude_self)) {
                            const uint32_t bit = (uint32_t)(id - lo);
                            atomicOr(&lds[bit >> 5], 1u << (bit & 31));
                        }
                    }
                }
            }
        };
        for (int t0 = p.t_begin + 2 * wave; t0 < p.t_end; t0 += 2 * nwaves) {
            uint32_t key[2], submask[2] = {0u, 0u};
            int part[2], nprobe[2];
            const uint32_t *roots[2];
#pragma unroll
            for (int x = 0; x < 2; x++) {
                const int t = t0 + x;
                key[x] = 0; part[x] = 0; nprobe[x] = 0; roots[x] = p.roots;
                if (t >= p.t_end) continue;
                const uint32_t h = (uint32_t)p.hashes[q * p.L + t];
                part[x] = p.parts[q * p.L + t];
                // RDT:753: 32 - numberOfLeadingZeros(h) - 4 probes; negative for 0 <= h < 8.  The keyed walk
                // (RDT:630-675) has no probe list: it looks up the hash itself, whatever its value.
                nprobe[x] = p.probe_own ? 1 : 32 - (h ? __clz(h) : 32) - 4;
                if (nprobe[x] <= 0) {
                    if (nprobe[x] < 0 && lane == 0 && c == 0 && MODE == 0) atomicAdd(p.short_hash_events, 1ull);
                    continue;
                }
                roots[x] = p.roots + ((size_t)t * P) * p.seg + (h >> p.bucket_bits);
                // the sub-indexes within `steps` bit flips of the query's own
                for (int s2 = 0; s2 < P; s2++)
                    if (__popc((uint32_t)(s2 ^ part[x])) <= p.steps && ((p.sub_mask >> s2) & 1u)) submask[x] |= 1u << s2;
                // lane 31 of each half walks the unflipped hash: it is not a probe (RDT:753-758),
                // it only names the query's own bucket for the collapse below
                key[x] = i == 31 ? h : h ^ (1u << i);
            }
            // four sub-indexes of each table per round: the lower half-wave walks the first and third, the
            // upper the second and fourth
            while (submask[0] | submask[1]) {
                int sw[2][2];
                uint32_t leaf[2][2], node[2][2];
#pragma unroll
                for (int x = 0; x < 2; x++) {
                    int sub[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        sub[k] = -1;
                        if (submask[x]) { sub[k] = __ffs((int)submask[x]) - 1; submask[x] &= submask[x] - 1; }
                    }
                    sw[x][0] = half ? sub[1] : sub[0];
                    sw[x][1] = half ? sub[3] : sub[2];
                    const bool act = p.probe_own ? (i == 31 && nprobe[x] > 0) : (i < nprobe[x] || (i == 31 && nprobe[x] > 0));
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        leaf[x][k] = NO_LEAF;
                        node[x][k] = (act && sw[x][k] >= 0) ? roots[x][(size_t)sw[x][k] * p.seg] : 0u;
                    }
                }
                for (int level = p.max_level; level >= 0 && (node[0][0] | node[0][1] | node[1][0] | node[1][1]); --level) {
                    uint32_t w[2][2];
#pragma unroll
                    for (int x = 0; x < 2; x++) {
                        const uint32_t slot = (key[x] >> (level * p.nbits)) & (uint32_t)(p.dns - 1);
#pragma unroll
                        for (int k = 0; k < 2; k++) w[x][k] = node[x][k] ? p.dnodes[(size_t)node[x][k] * p.dns + slot] : 0u;
                    }
#pragma unroll
                    for (int x = 0; x < 2; x++)
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            if (w[x][k] & 1u) { leaf[x][k] = w[x][k] >> 1; node[x][k] = 0; }
                            else node[x][k] = w[x][k] >> 1;
                        }
                }
                // Collapse probes that reached the same bucket (same sub-index only: buckets of
                // different sub-indexes are disjoint).  A flipped bit either changes the path at its
                // level - then the probe ends in a subtree no other probe enters - or it lies below
                // the level where the path ends (or in bits the trie does not use) and the probe ends
                // in the query's own bucket.  So only the own bucket can be reached more than once.
                bool own_hit[2][2];
#pragma unroll
                for (int x = 0; x < 2; x++)
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t own = __shfl(leaf[x][k], (lane & 32) + 31);
                        if (i == 31 && !p.probe_own) leaf[x][k] = NO_LEAF;
                        const unsigned long long same = __ballot(leaf[x][k] != NO_LEAF && leaf[x][k] == own) >> (lane & 32);
                        if (leaf[x][k] == own && (same & ((1ull << i) - 1ull) & 0xffffffffull)) leaf[x][k] = NO_LEAF;
                        own_hit[x][k] = leaf[x][k] != NO_LEAF && leaf[x][k] == own && sw[x][k] == part[x];
                    }
                if constexpr (MODE == 0) {
                    // the four buckets of a lane together: their bounds in one round of loads, then the four
                    // counter updates in flight at once; the old count is the query's place in the bucket's list
                    uint32_t lb[4], le[4], place[4];
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const uint32_t lf = leaf[v >> 1][v & 1];
                        lb[v] = 0; le[v] = 0;
                        if (lf != NO_LEAF) { lb[v] = p.leaf_off[lf]; le[v] = p.leaf_off[lf + 1]; }
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        raw_local += le[v] - lb[v];
                        place[v] = 0;
#if defined(DPF_PROBE_ABLATE) && (DPF_PROBE_ABLATE & 1)
                        if (p.lq_cnt && le[v] > lb[v] && leaf[v >> 1][v & 1] == 0xfffffff0u) place[v] = atomicAdd(&p.lq_cnt[0], 1u);
#else
                        if (p.lq_cnt && le[v] > lb[v]) place[v] = atomicAdd(&p.lq_cnt[leaf[v >> 1][v & 1]], 1u);
#endif
                    }
                    if (p.lq_cnt && p.q_leaves) {
                        // one LDS counter update per lane for all of its buckets
                        const uint32_t mine = (le[0] > lb[0]) + (le[1] > lb[1]) + (le[2] > lb[2]) + (le[3] > lb[3]);
                        uint32_t pos = mine ? atomicAdd(&s_nleaf, mine) : 0u;
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            if (le[v] > lb[v]) {
                                // top bit: the bucket the query itself would be stored in (same sub-index, same path)
#if defined(DPF_PROBE_ABLATE) && (DPF_PROBE_ABLATE & 2)
                                if (pos < (uint32_t)p.q_leaf_cap && leaf[v >> 1][v & 1] == 0xfffffff0u)
#else
                                if (pos < (uint32_t)p.q_leaf_cap)
#endif
                                    p.q_leaves[q * p.q_leaf_cap + pos] = make_uint2(leaf[v >> 1][v & 1], place[v] | (own_hit[v >> 1][v & 1] ? 0x80000000u : 0u));
                                pos++;
                            }
                        }
                    }
                } else {
#pragma unroll
                for (int x = 0; x < 2; x++) {
                    take(leaf[x][0], own_hit[x][0]);
                    take(leaf[x][1], own_hit[x][1]);
                }
                }
            }
        }
        if (MODE != 1) break;
        __syncthreads();

        // emit set bits in ascending order: each wave owns a contiguous block of words
        const int words_per_wave = (nwords + nwaves - 1) / nwaves;
        const int wbeg = wave * words_per_wave;
        const int wend = (wbeg + words_per_wave < nwords) ? wbeg + words_per_wave : nwords;
        uint32_t cnt = 0;
        for (int w = wbeg + lane; w < wend; w += 64) cnt += __popc(lds[w]);
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) s_wave_tot[wave] = cnt;
        __syncthreads();
        uint32_t base = s_emitted;
        for (int w = 0; w < wave; w++) base += s_wave_tot[w];
        int32_t *out = p.cand + p.cand_off[q];
        for (int w0 = wbeg; w0 < wend; w0 += 64) {
            const int w = w0 + lane;
            uint32_t word = (w < wend) ? lds[w] : 0u;
            uint32_t incl = __popc(word);
            const uint32_t mine = incl;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            uint32_t pos = base + incl - mine;
            const int32_t idbase = (int32_t)(lo + (int64_t)w * 32);
            while (word) {
                const int bpos = __ffs((int)word) - 1;
                word &= word - 1;
                out[pos++] = idbase + bpos;
            }
            base += __shfl(incl, 63);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < nwaves; w++) tot += s_wave_tot[w];
            s_emitted += tot;
        }
        __syncthreads();
    }

    if (MODE == 0) {
        for (int o = 32; o > 0; o >>= 1) raw_local += __shfl_xor(raw_local, o);
        if (lane == 0 && raw_local) atomicAdd(&s_raw, raw_local);
        __syncthreads();
        if (tid == 0) {
            const unsigned long long r = s_raw;
            p.raw[q] = r;
            p.bound[q] = (int64_t)(r < (unsigned long long)range ? r : (unsigned long long)range);
            if (p.q_leaves) {
#if defined(DPF_PROBE_ABLATE) && (DPF_PROBE_ABLATE & 2)
                p.q_leaf_cnt[q] = 0;
#else
                p.q_leaf_cnt[q] = (int32_t)s_nleaf;
                if (s_nleaf > (uint32_t)p.q_leaf_cap) *p.q_leaf_overflow = 1;
#endif
            }
        }
    } else if (MODE == 1) {
        if (tid == 0) p.cnt[q] = (int32_t)s_emitted;
    } else {
        __syncthreads();
        if (tid == 0) p.seed_cnt[q] = s_seed < (uint32_t)p.seed_cap ? (int32_t)s_seed : p.seed_cap;
    }
}

// ---------------------------------------------------------------------------
// k_probe_lanes: the counting pass of the bucket-major path (what k_probe<0> computes when it also
// histograms the incidences and remembers each query's buckets), one LANE per (query, table, sub-index).
//
// The <= 28 probes of one (query, table, sub-index) - the hash with one bit flipped each (RDT:753-756) - reach
// few distinct buckets, and which ones follows from the hash's OWN path through the trie: a flipped bit
// that belongs to a level the own path passes through as a directory sends the probe into the sibling slot
// of that directory (flipped bit b of the level: slot ^ (1 << b)) and from there down along the hash's own
// lower bits; it ends in a subtree no other probe enters.  Every other flipped bit - below the level at which
// the own path ends in a bucket, or in the bits above the trie's (max_level + 1) * nbits - leaves the path as
// it is and ends in the hash's own bucket.  So the lane walks the own path once, reads at every directory on
// it the nbits sibling slots (one 4 * dns-byte directory = one or a few cache lines), follows the siblings
// that are directories down, and takes the own bucket once if any probe reaches it.  k_probe<0> spends a
// lane per probe and most of its loads on addresses the neighbouring lanes load too; the address path of the
// memory pipeline is paid per instruction, not per distinct address.
// ---------------------------------------------------------------------------
constexpr int PL_THREADS = 256;
// What a lane finds goes through a queue in LDS and is counted from there by ALL lanes of the workgroup, one queue
// entry per lane: the buckets one lane finds at one level are siblings - neighbouring bucket numbers, so their
// offsets and incidence counters share cache lines - and the memory pipeline merges the lanes of one instruction
// that touch one line.  Counted by the finding lane itself, bucket by bucket, every atomic of an instruction went
// to a line of its own: 380 us instead of 100 on 10k queries over 30 tables.
// BINS: the incidences are not counted into the global histogram (7.7 M returning atomics on c3, which the fabric
// retires at about 43 G per second) but collected in LDS and written as (bucket, query) pairs into the region of the
// bucket's bin - a range of consecutive buckets; the bins are cut so that they hold about equally many pairs - with one
// reservation per (flush, touched bin); k_bin_lists then builds counts, offsets and lists bin by bin in LDS.
constexpr int PL_MAX_BINS = BIN_MAX;
constexpr int PL_PAIRS = 3072;                               // pairs a workgroup parks before it writes them out
template <int NB, bool BINS>   // NB: bits per trie level (dirNodeSize = 1 << NB)
__global__ __launch_bounds__(PL_THREADS) void k_probe_lanes(ProbeParams p, int64_t nq, int qpw, int T)
{
    constexpr int PER = NB + 1;                              // buckets a lane can find at one level: NB siblings and its own
    static_assert(PL_THREADS * PER <= PL_PAIRS, "one level's finds fit the parked pairs");
    __shared__ uint32_t s_nleaf[32];
    __shared__ unsigned long long s_raw[32];
    __shared__ uint32_t s_queue[PL_THREADS * PER];           // bucket | own-bucket flag (bit 31) ...
    __shared__ uint8_t s_queue_q[PL_THREADS * PER];          // ... and the query (of this workgroup's) it belongs to
    __shared__ uint32_t s_qn;
    __shared__ uint32_t s_pair_leaf[BINS ? PL_PAIRS : 1];    // parked pairs: bucket ...
    __shared__ uint8_t s_pair_q[BINS ? PL_PAIRS : 1];        // ... and query
    __shared__ uint32_t s_np;
    __shared__ uint32_t s_bin_cnt[BINS ? PL_MAX_BINS : 1], s_bin_base[BINS ? PL_MAX_BINS : 1];
    const int tid = threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.x * qpw;
    const int nql = (int)((nq - q0 < qpw) ? nq - q0 : qpw);
    if (tid < 32) { s_nleaf[tid] = 0; s_raw[tid] = 0; }
    if (tid == 0) { s_qn = 0; s_np = 0; }
    __syncthreads();
    const int P = 1 << p.pb;
    constexpr uint32_t mask = (1u << NB) - 1u;
    constexpr int dns = 1 << NB;
    const int trie_bits = (p.max_level + 1) * NB;
    // (BINS) the parked pairs into their bins' regions: count per bin, one reservation per touched bin, place
    auto flush_pairs = [&]() {
        if constexpr (BINS) {
            const uint32_t np = s_np;
            if (np == 0) return;                             // (uniform)
            for (int b = tid; b < p.nbins; b += PL_THREADS) s_bin_cnt[b] = 0;
            __syncthreads();
            for (uint32_t e = tid; e < np; e += PL_THREADS) {    // (the bin rides in the bits above the bucket number)
                const uint32_t leaf = s_pair_leaf[e], b = p.bin_map[leaf >> p.gran_shift];
                s_pair_leaf[e] = leaf | (b << BIN_LEAF_BITS);
                atomicAdd(&s_bin_cnt[b], 1u);
            }
            __syncthreads();
            for (int b = tid; b < p.nbins; b += PL_THREADS) {
                const uint32_t c = s_bin_cnt[b];
                if (c) {
                    const uint32_t base = atomicAdd(&p.bin_cursor[b], c);
                    s_bin_base[b] = base;
                    if (base + c > p.bin_cap[b]) *p.bin_overflow = 1;
                    s_bin_cnt[b] = 0;
                }
            }
            __syncthreads();
            for (uint32_t e = tid; e < np; e += PL_THREADS) {
                const uint32_t w = s_pair_leaf[e], leaf = w & ((1u << BIN_LEAF_BITS) - 1u), b = w >> BIN_LEAF_BITS;
                const uint32_t at = s_bin_base[b] + atomicAdd(&s_bin_cnt[b], 1u);
                if (at < p.bin_cap[b]) p.pairs[(size_t)p.bin_off[b] + at] = make_uint2(leaf, (uint32_t)(q0 + s_pair_q[e]));
            }
            __syncthreads();
            if (tid == 0) s_np = 0;
            __syncthreads();
        }
    };
    for (int x0 = 0; x0 < nql * T; x0 += PL_THREADS) {       // (uniform over the workgroup: barriers inside)
        const int x = x0 + tid;
        const bool live = x < nql * T;
        const int ql = live ? x / T : 0, r = x - ql * T;
        const int t = p.t_begin + (live ? r / p.nsub : 0), j = live ? r - (r / p.nsub) * p.nsub : 0;
        const int64_t q = q0 + ql;
        uint32_t h = 0, node = 0;
        int nprobe = 0;
        bool own_sub = false;
        if (live) {
            h = (uint32_t)p.hashes[q * p.L + t];
            const int part = p.parts[q * p.L + t];
            // RDT:753: 32 - numberOfLeadingZeros(h) - 4 probes; negative for 0 <= h < 8.  The keyed walk (RDT:630-675)
            // has no probe list: it looks up the hash itself, whatever its value.
            nprobe = p.probe_own ? 1 : 32 - (h ? __clz(h) : 32) - 4;
            if (nprobe < 0 && j == 0) atomicAdd(p.short_hash_events, 1ull);
            const int sub = part ^ (int)p.sub_delta[j];
            own_sub = sub == part;
            if (nprobe > 0 && ((p.sub_mask >> sub) & 1u)) node = p.roots[((size_t)t * P + sub) * p.seg + (h >> p.bucket_bits)];
        }
        for (int level = p.max_level; level >= 0; --level) {
            if (!__syncthreads_or(node != 0)) break;         // (also: the queue of the level before has been read out)
            uint32_t lf[PER];
#pragma unroll
            for (int b = 0; b < PER; b++) lf[b] = NO_LEAF;
            if (node != 0) {
                const int shift = level * NB;
                const uint32_t slot = (h >> shift) & mask;
                const uint32_t *dn = p.dnodes + (size_t)node * dns;
                // the own slot and the sibling slots of this level's flipped bits, in flight together
                uint32_t w[NB];
#pragma unroll
                for (int b = 0; b < NB; b++)
                    w[b] = (!p.probe_own && shift + b < nprobe) ? dn[slot ^ (1u << b)] : 0u;
                const uint32_t w_own = dn[slot];
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if (w[b] & 1u) lf[b] = w[b] >> 1;
#if defined(DPF_PROBE_ABLATE) && (DPF_PROBE_ABLATE & 4)
                    else if (w[b] == 0xfffffff0u) {
#else
                    else if (w[b]) {
#endif
                        // a sibling that is a directory: down along the hash's own lower bits
                        uint32_t n2 = w[b] >> 1;
                        for (int l2 = level - 1; l2 >= 0; --l2) {
                            const uint32_t w2 = p.dnodes[(size_t)n2 * dns + ((h >> (l2 * NB)) & mask)];
                            if (w2 & 1u) { lf[b] = w2 >> 1; break; }
                            if (w2 == 0) break;
                            n2 = w2 >> 1;
                        }
                    }
                }
                // the own path: on through a directory; a bucket is reached by the probes whose flipped bit lies below
                // this level or above the trie's bits (and by the keyed walk)
                node = 0;
                if (w_own & 1u) {
                    if (p.probe_own || level > 0 || nprobe > trie_bits) lf[NB] = (w_own >> 1) | (own_sub ? 0x80000000u : 0u);
                } else {
                    node = w_own >> 1;
                }
                int mine = 0;
#pragma unroll
                for (int b = 0; b < PER; b++) mine += lf[b] != NO_LEAF;
                if (mine) {
                    uint32_t at = atomicAdd(&s_qn, (uint32_t)mine);
#pragma unroll
                    for (int b = 0; b < PER; b++)
                        if (lf[b] != NO_LEAF) { s_queue[at] = lf[b]; s_queue_q[at] = (uint8_t)ql; at++; }
                }
            }
            __syncthreads();
            // every lane counts one queue entry at a time
            const uint32_t n = s_qn;
            if constexpr (BINS) {
                if (s_np + n > (uint32_t)PL_PAIRS) flush_pairs();     // (uniform: both numbers are the workgroup's)
            }
            for (uint32_t e = tid; e < n; e += PL_THREADS) {
                const uint32_t ent = s_queue[e], leaf = ent & 0x7fffffffu;
                const int eq = s_queue_q[e];
#if defined(DPF_PROBE_ABLATE) && (DPF_PROBE_ABLATE & 8)
                const uint32_t lb = 0, le = 1;
#else
                const uint32_t lb = p.leaf_off[leaf], le = p.leaf_off[leaf + 1];
#endif
                if (le > lb) {
                    uint32_t place = 0;
                    if constexpr (BINS) {
                        const uint32_t at = atomicAdd(&s_np, 1u);
                        s_pair_leaf[at] = leaf;
                        s_pair_q[at] = (uint8_t)eq;
                    } else {
#if defined(DPF_PROBE_ABLATE) && (DPF_PROBE_ABLATE & 1)
                        place = leaf == 0xfffffff0u ? atomicAdd(&p.lq_cnt[0], 1u) : 0u;
#else
                        place = atomicAdd(&p.lq_cnt[leaf], 1u);
#endif
                    }
                    const uint32_t pos = atomicAdd(&s_nleaf[eq], 1u);
                    atomicAdd(&s_raw[eq], (unsigned long long)(le - lb));
                    // top bit: the bucket the query itself would be stored in (own sub-index, own path)
                    if (pos < (uint32_t)p.q_leaf_cap)
                        p.q_leaves[(q0 + eq) * p.q_leaf_cap + pos] = make_uint2(leaf, place | (ent & 0x80000000u));
                }
            }
            __syncthreads();
            if (tid == 0) s_qn = 0;
        }
        __syncthreads();
    }
    __syncthreads();
    flush_pairs();
    const int64_t range = p.row_end - p.row_begin;
    if (tid < nql) {
        const int64_t q = q0 + tid;
        const unsigned long long rr = s_raw[tid];
        p.raw[q] = rr;
        p.bound[q] = (int64_t)(rr < (unsigned long long)range ? rr : (unsigned long long)range);
#if defined(DPF_PROBE_ABLATE) && (DPF_PROBE_ABLATE & 2)
        p.q_leaf_cnt[q] = 0;
#else
        p.q_leaf_cnt[q] = (int32_t)s_nleaf[tid];
        if (s_nleaf[tid] > (uint32_t)p.q_leaf_cap) *p.q_leaf_overflow = 1;
#endif
    }
}

bool launch_probe_lanes(ProbeParams p, int64_t nq, hipStream_t stream)
{
    if (nq <= 0) return true;
    if (!p.q_leaves || p.nbits < 5 || p.nbits > 8 || p.dns != (1 << p.nbits) || p.pb > 5) return false;
    const bool bins = p.bin_cursor != nullptr;
    if (bins ? (p.nbins < 1 || p.nbins > PL_MAX_BINS || !p.pairs || !p.bin_cap || !p.bin_off || !p.bin_overflow) : !p.lq_cnt) return false;
    p.nsub = 0;
    for (int d = 0; d < (1 << p.pb); d++)
        if (__builtin_popcount((unsigned)d) <= p.steps) p.sub_delta[p.nsub++] = (uint8_t)d;
    const int T = (p.t_end - p.t_begin) * p.nsub;
    if (T <= 0) return false;
    // whole queries per workgroup (their bucket counters live in LDS); few per workgroup when a query has few lanes,
    // so that small forests still spread over the chip
    int qpw = PL_THREADS / T;
    if (qpw < 1) qpw = 1;
    if (qpw > 32) qpw = 32;
    const int64_t grid = (nq + qpw - 1) / qpw;
#define DPF_PL(NB)                                                                                                      \
    do {                                                                                                                \
        if (bins) hipLaunchKernelGGL((k_probe_lanes<NB, true>), dim3((unsigned)grid), dim3(PL_THREADS), 0, stream, p, nq, qpw, T); \
        else hipLaunchKernelGGL((k_probe_lanes<NB, false>), dim3((unsigned)grid), dim3(PL_THREADS), 0, stream, p, nq, qpw, T);     \
    } while (0)
    switch (p.nbits) {
    case 5: DPF_PL(5); break;
    case 6: DPF_PL(6); break;
    case 7: DPF_PL(7); break;
    default: DPF_PL(8); break;
    }
#undef DPF_PL
    return true;
}

// k_fill_lists + k_seed_sample: what k_probe<2> does, from the bucket lists k_probe<0> recorded instead of a
// second walk of the forest.  k_fill_lists writes the query into its place in every bucket's list (it needs the
// lists' offsets); k_seed_sample copies ids of the query's first buckets into its seed sample (it needs only what
// k_probe<0> left, so the host may run it beside the offset scans).  Only launched when no query's list was truncated.
__global__ __launch_bounds__(256) void k_fill_lists(ProbeParams p)
{
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    if (p.abort && *p.abort) return;                      // the lists have no room for this batch: it is run again
    const int n = min(p.q_leaf_cnt[q], p.q_leaf_cap);
    const uint2 *mine = p.q_leaves + q * p.q_leaf_cap;
    for (int j0 = tid; j0 < n; j0 += 1024) {
        // four entries per thread and round: the entries, then their buckets' list offsets, in flight together
        uint2 lp[4];                                      // {bucket, place in the bucket's list | own-bucket flag}
        uint32_t off[4];
#pragma unroll
        for (int u = 0; u < 4; u++) lp[u] = j0 + 256 * u < n ? mine[j0 + 256 * u] : make_uint2(0u, 0u);
#pragma unroll
        for (int u = 0; u < 4; u++) off[u] = j0 + 256 * u < n ? p.lq_off32[lp[u].x] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (j0 + 256 * u >= n) continue;
            p.leaf_queries[off[u] + (lp[u].y & 0x7fffffffu)] = (int32_t)q;
        }
    }
}

__global__ __launch_bounds__(256) void k_seed_sample(ProbeParams p)
{
    __shared__ uint32_t s_seed, s_nown;
    __shared__ uint32_t s_own[256], s_ob[256], s_oe[256];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(p.q_leaf_cnt[q], p.q_leaf_cap);
    const int32_t qid = p.qids ? p.qids[q] : -1;
    if (tid == 0) { s_seed = 0; s_nown = 0; }
    __syncthreads();
    const uint2 *mine = p.q_leaves + q * p.q_leaf_cap;
    // the query's own buckets: the entries k_probe<0> flagged
    for (int j = tid; j < n; j += 256) {
        const uint2 e = mine[j];
        if (e.y & 0x80000000u) {
            const uint32_t at = atomicAdd(&s_nown, 1u);
            if (at < 256u) s_own[at] = e.x;
        }
    }
    __syncthreads();
    // Seed sample.  First an equal share from each of the query's OWN buckets (one per table: the rows that
    // collide with the query on the whole hash are its likeliest neighbours, and a true neighbour sits in
    // several of them), then whole buckets in list order until the sample is full.
    auto copy_from = [&](uint32_t leaf, uint32_t most) {
        const uint32_t b = p.leaf_off[leaf], end = p.leaf_off[leaf + 1], e = most < end - b ? b + most : end;
        for (uint32_t k0 = b; k0 < e; k0 += 64) {
            const uint32_t k = k0 + lane;
            const int32_t id = k < e ? p.ids[k] : -1;
            const bool ok = k < e && id >= p.row_begin && id < p.row_end && !self_excluded(id, qid, p.exclude_self);
            const unsigned long long m = __ballot(ok);
            if (m == 0) continue;
            uint32_t at0 = 0;
            if (lane == 0) at0 = atomicAdd(&s_seed, (uint32_t)__popcll(m));
            at0 = __shfl(at0, 0);
            if (at0 >= (uint32_t)p.seed_cap) break;
            const uint32_t at = at0 + __popcll(m & ((1ull << lane) - 1ull));
            if (ok && at < (uint32_t)p.seed_cap) p.seed_cand[q * p.seed_cap + at] = id;
        }
    };
    const uint32_t nown = min(s_nown, 256u);
    if (nown > 0) {
        // all own buckets at once: their bounds in one round of loads, then one thread per (bucket, entry)
        const uint32_t share = ((uint32_t)p.seed_cap + nown - 1u) / nown;
        if ((uint32_t)tid < nown) {
            const uint32_t leaf = s_own[tid], b = p.leaf_off[leaf], e = p.leaf_off[leaf + 1];
            s_ob[tid] = b;
            s_oe[tid] = share < e - b ? b + share : e;
        }
        __syncthreads();
        const uint32_t total = nown * share;
        for (uint32_t s0 = 0; s0 < total; s0 += 256u) {
            const uint32_t s = s0 + (uint32_t)tid, j = s / share;
            const uint32_t k = s < total ? s_ob[j] + (s - j * share) : 0xffffffffu;
            const bool there = s < total && k < s_oe[j];
            const int32_t id = there ? p.ids[k] : -1;
            const bool ok = there && id >= p.row_begin && id < p.row_end && !self_excluded(id, qid, p.exclude_self);
            const unsigned long long m = __ballot(ok);
            if (m == 0) continue;
            uint32_t at0 = 0;
            if (lane == 0) at0 = atomicAdd(&s_seed, (uint32_t)__popcll(m));
            at0 = __shfl(at0, 0);
            const uint32_t at = at0 + __popcll(m & ((1ull << lane) - 1ull));
            if (ok && at < (uint32_t)p.seed_cap) p.seed_cand[q * p.seed_cap + at] = id;
        }
    }
    __syncthreads();
    for (int j = wave; j < n && s_seed < (uint32_t)p.seed_cap; j += 4)
        if (!(mine[j].y & 0x80000000u)) copy_from(mine[j].x, 0xffffffffu);
    __syncthreads();
    if (tid == 0) p.seed_cnt[q] = s_seed < (uint32_t)p.seed_cap ? (int32_t)s_seed : p.seed_cap;
}

void launch_seed_sample(const ProbeParams &p, int64_t nq, hipStream_t stream)
{
    if (nq > 0) hipLaunchKernelGGL(k_seed_sample, dim3((unsigned)nq), dim3(256), 0, stream, p);
}

void launch_fill_lists(const ProbeParams &p, int64_t nq, hipStream_t stream)
{
    if (nq > 0) hipLaunchKernelGGL(k_fill_lists, dim3((unsigned)nq), dim3(256), 0, stream, p);
}

void launch_probe(int mode, const ProbeParams &p, int64_t nq, hipStream_t stream)
{
    if (nq <= 0) return;
    if (mode == 0) {
        hipLaunchKernelGGL((k_probe<0>), dim3((unsigned)nq), dim3(256), 0, stream, p);
    } else if (mode == 2) {
        hipLaunchKernelGGL((k_probe<2>), dim3((unsigned)nq), dim3(256), 0, stream, p);
    } else {
        size_t shmem = (size_t)p.chunk_words * sizeof(uint32_t);
        hipLaunchKernelGGL((k_probe<1>), dim3((unsigned)nq), dim3(1024), shmem, stream, p);
    }
}

// ---------------------------------------------------------------------------
// k_scan_*: off[0] = 0, off[i+1] = off[i] + in[i], in three launches:
//   sums   each workgroup adds up its SCAN_TILE inputs and parks the sum in off[] at
//          the tile's first slot (no scratch buffer: off[] is rewritten by the last phase)
//   tiles  one workgroup turns the parked sums into exclusive tile bases, writes off[n]
//   apply  each workgroup picks up its base and scans its tile
// ---------------------------------------------------------------------------
static constexpr int SCAN_THREADS = 256, SCAN_PER = 8, SCAN_TILE = SCAN_THREADS * SCAN_PER;

__device__ __forceinline__ int64_t block_sum256(int64_t v, int64_t *sh)
{
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <typename TIN>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(const TIN *__restrict__ in, int64_t n, int64_t *__restrict__ off)
{
    __shared__ int64_t sh[4];
    const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILE;
    int64_t s = 0;
    for (int j = 0; j < SCAN_PER; j++) {
        const int64_t i = t0 + j * SCAN_THREADS + threadIdx.x;
        if (i < n) s += (int64_t)in[i];
    }
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) off[t0] = s;
}

__global__ __launch_bounds__(1024) void k_scan_tiles(int64_t n, int64_t ntiles, int64_t *__restrict__ off)
{
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (ntiles + 1023) / 1024;
    const int64_t b = tid * per, e = (b + per < ntiles) ? b + per : ntiles;
    int64_t s = 0;
    for (int64_t i = b; i < e; i++) s += off[i * SCAN_TILE];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = part[tid] - s;
    for (int64_t i = b; i < e; i++) { const int64_t v = off[i * SCAN_TILE]; off[i * SCAN_TILE] = run; run += v; }
    if (tid == 1023) off[n] = part[1023];
}

template <typename TIN>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const TIN *__restrict__ in, int64_t n, int64_t *__restrict__ off)
{
    __shared__ int64_t wsum[4];
    const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILE;
    const int64_t base = off[t0];
    // thread t owns inputs t*SCAN_PER .. t*SCAN_PER+SCAN_PER-1 of the tile
    int64_t v[SCAN_PER], s = 0;
    const int64_t i0 = t0 + (int64_t)threadIdx.x * SCAN_PER;
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++) { v[j] = i0 + j < n ? (int64_t)in[i0 + j] : 0; s += v[j]; }
    int64_t inc = s;                                  // inclusive scan over the wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const int64_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();                                  // also orders the read of off[t0] before the writes below
    int64_t run = base + inc - s;
    for (int u = 0; u < w; u++) run += wsum[u];
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++) { if (i0 + j < n) off[i0 + j] = run; run += v[j]; }
}

template <typename TIN>
static void launch_scan(const TIN *in, int64_t n, int64_t *off, hipStream_t stream)
{
    const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (ntiles > 0) hipLaunchKernelGGL((k_scan_sums<TIN>), dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, in, n, off);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, stream, n, ntiles, off);
    if (ntiles > 0) hipLaunchKernelGGL((k_scan_apply<TIN>), dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, in, n, off);
}
void launch_scan_i64(const int64_t *in, int64_t n, int64_t *off, hipStream_t stream) { launch_scan(in, n, off, stream); }
void launch_scan_i32(const int32_t *in, int64_t n, int64_t *off, hipStream_t stream) { launch_scan(in, n, off, stream); }

// ---------------------------------------------------------------------------
// k_compact: dst[dst_off[q] + j] = src[src_off[q] + j], j < cnt[q]
// ---------------------------------------------------------------------------
__global__ void k_compact(const int32_t *__restrict__ src, const int64_t *__restrict__ src_off,
                          const int32_t *__restrict__ cnt, const int64_t *__restrict__ dst_off,
                          int32_t *__restrict__ dst)
{
    const int64_t q = blockIdx.x;
    const int32_t n = cnt[q];
    const int32_t *s = src + src_off[q];
    int32_t *d = dst + dst_off[q];
    for (int32_t j = threadIdx.x; j < n; j += blockDim.x) d[j] = s[j];
}

void launch_compact(const int32_t *src, const int64_t *src_off, const int32_t *cnt,
                    const int64_t *dst_off, int32_t *dst, int64_t nq, hipStream_t stream)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_compact, dim3((unsigned)nq), dim3(256), 0, stream, src, src_off, cnt, dst_off, dst);
}

// ---------------------------------------------------------------------------
// k_rerank
// ---------------------------------------------------------------------------

// order of the final list: metric dot = descending inner product (Dense:490),
// metric l2 = ascending squared distance; equal scores by ascending id.
template <int METRIC>
__device__ inline bool better(double sa, int32_t ia, double sb, int32_t ib)
{
    if (sa != sb) return METRIC == 0 ? (sa > sb) : (sa < sb);
    return ia < ib;
}

template <int METRIC>
__device__ inline double worst_score()
{
    return METRIC == 0 ? -HUGE_VAL : HUGE_VAL;
}

// A wave keeps its best K entries sorted, entry e in lane e.
template <int METRIC>
__device__ inline void topk_insert(double s, int32_t id, double &ls, int32_t &lid, int K, int lane)
{
    const bool shift = (lane < K) && better<METRIC>(s, id, ls, lid);
    const unsigned long long mask = __ballot(shift);
    if (mask == 0) return;
    // the same id reached through two buckets scores identically: keep one (Set semantics)
    if (__ballot((lane < K) && lid == id && ls == s)) return;
    const int pos = __ffsll((long long)mask) - 1;
    const double ps = __shfl_up(ls, 1);
    const int32_t pid = __shfl_up(lid, 1);
    if (lane == pos) { ls = s; lid = id; }
    else if (lane > pos && lane < K) { ls = ps; lid = pid; }
}

template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float4 type; static constexpr int N = 4; };
template <> struct Vec16<double> { typedef double2 type; static constexpr int N = 2; };

template <typename T, int METRIC>
__device__ inline double accum16(const typename Vec16<T>::type &q, const typename Vec16<T>::type &x, double acc);

template <> __device__ inline double accum16<float, 0>(const float4 &q, const float4 &x, double acc)
{
    acc = fma((double)q.x, (double)x.x, acc); acc = fma((double)q.y, (double)x.y, acc);
    acc = fma((double)q.z, (double)x.z, acc); acc = fma((double)q.w, (double)x.w, acc);
    return acc;
}
template <> __device__ inline double accum16<float, 1>(const float4 &q, const float4 &x, double acc)
{
    double d0 = (double)q.x - (double)x.x, d1 = (double)q.y - (double)x.y;
    double d2 = (double)q.z - (double)x.z, d3 = (double)q.w - (double)x.w;
    acc = fma(d0, d0, acc); acc = fma(d1, d1, acc); acc = fma(d2, d2, acc); acc = fma(d3, d3, acc);
    return acc;
}
template <> __device__ inline double accum16<double, 0>(const double2 &q, const double2 &x, double acc)
{
    acc = fma(q.x, x.x, acc); acc = fma(q.y, x.y, acc);
    return acc;
}
template <> __device__ inline double accum16<double, 1>(const double2 &q, const double2 &x, double acc)
{
    double d0 = q.x - x.x, d1 = q.y - x.y;
    acc = fma(d0, d0, acc); acc = fma(d1, d1, acc);
    return acc;
}

// One workgroup (4 waves) per query.  A group of LPR lanes owns one candidate
// row per step: lane gl of the group loads 16-byte pieces gl, gl+LPR, ... of the
// row (whole rows, consecutive lanes on consecutive 16 B: fully coalesced),
// accumulates its share of the fp64 score, and the group reduces with
// shuffles.  UR rows per group are in flight at once.  Scores are fp64 so that
// the order of near-equal candidates matches an fp64 CPU evaluation.
template <typename T, int METRIC, int LPR, int NJ, int UR>
__global__ __launch_bounds__(256) void k_rerank(RerankParams p)
{
    typedef typename Vec16<T>::type V;
    __shared__ double s_score[4 * 64];
    __shared__ int32_t s_id[4 * 64];

    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int GPW = 64 / LPR;
    const int grp = lane / LPR, gl = lane % LPR;
    const int K = p.k;
    const int nc = p.chunks16;            // 16-byte pieces per row

    const char *qrow = (const char *)p.Q + (size_t)q * p.q_stride_bytes;
    V qv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int c = gl + j * LPR;
        if (c < nc) qv[j] = *(const V *)(qrow + (size_t)c * 16);
        else { V z = {}; qv[j] = z; }
    }

    double ls = worst_score<METRIC>();
    int32_t lid = 0x7fffffff;
    double thr_s = ls; int32_t thr_id = lid;

    const int64_t coff = p.cand_off ? p.cand_off[q] : q * p.cand_stride;
    int32_t cnt = p.cnt ? p.cnt[q] : (int32_t)(p.cand_off[q + 1] - coff);
    if (!p.cand_off && cnt > p.cand_stride) cnt = (int32_t)p.cand_stride;
    const int32_t *cand = p.cand + coff;
    const char *base = (const char *)p.base;

    for (int32_t b0 = wave * GPW * UR; b0 < cnt; b0 += 4 * GPW * UR) {
        int32_t id[UR];
        V x[UR][NJ];
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const int32_t idx = b0 + u * GPW + grp;
            id[u] = idx < cnt ? cand[idx] : -1;
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const bool ok = id[u] >= p.row_begin && id[u] < p.row_end;
            const char *row = base + (size_t)(ok ? (id[u] - p.row_begin) : 0) * p.base_stride_bytes;
            if (!ok) id[u] = -1;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int c = gl + j * LPR;
                if (ok && c < nc) x[u][j] = *(const V *)(row + (size_t)c * 16);
                else { V z = {}; x[u][j] = qv[j]; if (METRIC == 0) x[u][j] = z; }
            }
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; j++) acc = accum16<T, METRIC>(qv[j], x[u][j], acc);
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            const bool cand_ok = (gl == 0) && (id[u] >= 0) && better<METRIC>(acc, id[u], thr_s, thr_id);
            unsigned long long m = __ballot(cand_ok);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const double s = __shfl(acc, src);
                const int32_t sid = __shfl(id[u], src);
                topk_insert<METRIC>(s, sid, ls, lid, K, lane);
                thr_s = __shfl(ls, K - 1);
                thr_id = __shfl(lid, K - 1);
            }
        }
    }

    // merge the four waves' lists through LDS; wave 0 inserts the others' entries
    s_score[wave * 64 + lane] = ls;
    s_id[wave * 64 + lane] = lid;
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; w++) {
            for (int e = 0; e < K; e++) {
                const double s = s_score[w * 64 + e];
                const int32_t sid = s_id[w * 64 + e];
                if (sid == 0x7fffffff) break;
                topk_insert<METRIC>(s, sid, ls, lid, K, lane);
            }
        }
        if (lane < K) {
            const bool has = lid != 0x7fffffff;
            p.out_ids[q * K + lane] = has ? lid : -1;
            p.out_scores[q * K + lane] = has ? ls : nan("");
        }
    }
}

template <typename T, int METRIC>
static void launch_rerank_tm(const RerankParams &p, int64_t nq, hipStream_t stream)
{
    const int nc = p.chunks16;
    dim3 grid((unsigned)nq), block(256);
#define DPF_RR(LPR, NJ, UR) hipLaunchKernelGGL((k_rerank<T, METRIC, LPR, NJ, UR>), grid, block, 0, stream, p)
    if (nc <= 8) DPF_RR(8, 1, 4);
    else if (nc <= 16) DPF_RR(16, 1, 8);
    else if (nc <= 32) DPF_RR(32, 1, 8);
    else if (nc <= 64) DPF_RR(64, 1, 8);
    else if (nc <= 128) DPF_RR(64, 2, 4);
    else if (nc <= 192) DPF_RR(64, 3, 4);
    else if (nc <= 256) DPF_RR(64, 4, 2);
    else if (nc <= 384) DPF_RR(64, 6, 2);
    else if (nc <= 512) DPF_RR(64, 8, 1);
    else DPF_RR(64, 16, 1);
#undef DPF_RR
}

int rerank_max_chunks16() { return 1024; }

// Short candidate lists (the seed samples and the forwarded lists of the bucket-major path: a few
// hundred rows per query): one WAVE per query, 8 lanes per candidate row, UR*8 rows in flight, the
// next step's ids requested before this step's rows are scored, so a query costs a handful of
// memory round trips instead of a workgroup's launch, LDS merge and barrier.  The arithmetic
// reproduces k_rerank's bit for bit for rows of up to 32 16-byte pieces: there lane c of an
// LPR-lane group sums its piece from zero and the group adds by xor-shuffles LPR/2, ..., 1; here
// lane gl holds pieces gl, gl + 8, ... as separate partial sums and adds them in the same tree
// (offsets >= 8 inside the lane, 4, 2, 1 across lanes).
template <typename T, int METRIC, int NJ, int UR>
__global__ __launch_bounds__(256) void k_rerank_short(RerankParams p, int64_t nq)
{
    typedef typename Vec16<T>::type V;
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int grp = lane >> 3, gl = lane & 7;
    const int K = p.k;
    const int nc = p.chunks16;

    const char *qrow = (const char *)p.Q + (size_t)q * p.q_stride_bytes;
    V qv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int c = gl + j * 8;
        if (c < nc) qv[j] = *(const V *)(qrow + (size_t)c * 16);
        else { V z = {}; qv[j] = z; }
    }
    double ls = worst_score<METRIC>();
    int32_t lid = 0x7fffffff;
    double thr_s = ls; int32_t thr_id = lid;

    const int64_t coff = q * p.cand_stride;
    int32_t cnt = p.cnt[q];
    if (cnt > p.cand_stride) cnt = (int32_t)p.cand_stride;
    const int32_t *cand = p.cand + coff;
    const char *base = (const char *)p.base;

    int32_t idn[UR];
#pragma unroll
    for (int u = 0; u < UR; u++) idn[u] = u * 8 + grp < cnt ? cand[u * 8 + grp] : -1;
    for (int32_t b0 = 0; b0 < cnt; b0 += 8 * UR) {
        int32_t id[UR];
        V x[UR][NJ];
#pragma unroll
        for (int u = 0; u < UR; u++) {
            id[u] = idn[u];
            const bool ok = id[u] >= p.row_begin && id[u] < p.row_end;
            const char *row = base + (size_t)(ok ? (id[u] - p.row_begin) : 0) * p.base_stride_bytes;
            if (!ok) id[u] = -1;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int c = gl + j * 8;
                if (ok && c < nc) x[u][j] = *(const V *)(row + (size_t)c * 16);
                else { V z = {}; x[u][j] = qv[j]; if (METRIC == 0) x[u][j] = z; }
            }
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const int32_t idx = b0 + 8 * UR + u * 8 + grp;
            idn[u] = idx < cnt ? cand[idx] : -1;
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            double part[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) part[j] = accum16<T, METRIC>(qv[j], x[u][j], 0.0);
#pragma unroll
            for (int o = NJ / 2; o > 0; o >>= 1)
#pragma unroll
                for (int i = 0; i < o; i++) part[i] += part[i + o];
            double acc = part[0];
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            const bool cand_ok = (gl == 0) && (id[u] >= 0) && better<METRIC>(acc, id[u], thr_s, thr_id);
            unsigned long long m = __ballot(cand_ok);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const double sc = __shfl(acc, src);
                const int32_t sid = __shfl(id[u], src);
                topk_insert<METRIC>(sc, sid, ls, lid, K, lane);
                thr_s = __shfl(ls, K - 1);
                thr_id = __shfl(lid, K - 1);
            }
        }
    }
    if (lane < K) {
        const bool has = lid != 0x7fffffff;
        p.out_ids[q * K + lane] = has ? lid : -1;
        p.out_scores[q * K + lane] = has ? ls : nan("");
    }
}

// true when launch_rerank_short serves this row length with k_rerank's exact arithmetic
bool rerank_short_applies(const RerankParams &p) { return p.chunks16 <= 32 && !p.cand_off && p.cnt; }

template <typename T, int METRIC>
static void launch_rerank_short_tm(const RerankParams &p, int64_t nq, hipStream_t stream, bool few_rows)
{
    dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    const int nc = p.chunks16;
    if (nc <= 8) hipLaunchKernelGGL((k_rerank_short<T, METRIC, 1, 8>), grid, block, 0, stream, p, nq);
    else if (nc <= 16) hipLaunchKernelGGL((k_rerank_short<T, METRIC, 2, 8>), grid, block, 0, stream, p, nq);
    // (lists of a few tens of rows - the forward lists after pruning: 10 to 35 rows per query - gain nothing from 32 rows
    // in flight per wave and lose the occupancy their registers cost: 8 rows per round, 41 -> 35 us on 10k queries over 30
    // trees, 61 -> 53 us over one tree; 64 rows in flight: 59 / 81 us)
    else if (few_rows) hipLaunchKernelGGL((k_rerank_short<T, METRIC, 4, 1>), grid, block, 0, stream, p, nq);
    else hipLaunchKernelGGL((k_rerank_short<T, METRIC, 4, 4>), grid, block, 0, stream, p, nq);
}

void launch_rerank_short(int dtype, int metric, const RerankParams &p, int64_t nq, hipStream_t stream, bool few_rows)
{
    if (nq <= 0) return;
    if (dtype == 0) {
        if (metric == 0) launch_rerank_short_tm<float, 0>(p, nq, stream, few_rows);
        else launch_rerank_short_tm<float, 1>(p, nq, stream, few_rows);
    } else {
        if (metric == 0) launch_rerank_short_tm<double, 0>(p, nq, stream, few_rows);
        else launch_rerank_short_tm<double, 1>(p, nq, stream, few_rows);
    }
}


void launch_rerank(int dtype, int metric, const RerankParams &p, int64_t nq, hipStream_t stream)
{
    if (nq <= 0) return;
    if (dtype == 0) {
        if (metric == 0) launch_rerank_tm<float, 0>(p, nq, stream);
        else launch_rerank_tm<float, 1>(p, nq, stream);
    } else {
        if (metric == 0) launch_rerank_tm<double, 0>(p, nq, stream);
        else launch_rerank_tm<double, 1>(p, nq, stream);
    }
}

// ---------------------------------------------------------------------------
// Bucket-major candidate filter
// ---------------------------------------------------------------------------
// The per-query gather above reads a bucket's rows once per query that probes
// the bucket.  In a batch many queries probe the same buckets, so the scan turns
// the loop around: a wave owns a 64-row tile of one bucket and walks the list of
// queries that reach the bucket.  The rows come from HBM once per tile instead of
// once per (query, bucket).
//
// The score here only FILTERS: a row is forwarded to the exact fp64 rerank when
// its approximate score is within the proven error bound of the query's
// threshold, an upper bound of the query's exact k-th best score (any subset's
// k-th best bounds the full set's k-th best).  Every true top-k member therefore
// survives; the exact kernel orders them.
// order-preserving float <-> int32 (an involution), so thresholds can be tightened with integer atomics
__device__ __host__ inline int32_t enc_f32(float f)
{
    int32_t b;
    memcpy(&b, &f, 4);
    return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ inline float dec_f32(int32_t b)
{
    return __int_as_float(b >= 0 ? b : (b ^ 0x7fffffff));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// k_leafscan_f16: the bucket-major filter on an fp16 copy of the rows
// ---------------------------------------------------------------------------
// 64-row x 16-query blocks of scores on v_mfma_f32_16x16x32_f16; both operands are
// fp16 copies (rows padded to KS*32 dimensions, 64*KS bytes, never straddling a
// 128-byte line): half the HBM bytes of the fp32 rows.  The filter stays exact-safe because the rounding
// is MEASURED, not assumed: the conversion kernel stores |x - x~| per row (and per
// query), so by the triangle inequality
//     | |q~ - x~| - |q - x| | <= |q - q~| + |x - x~| =: eta,
// and a candidate is forwarded whenever its fp16 distance could still be within the
// threshold,   s - slack32 <= (sqrt(tau) + eta)^2 .
// What is forwarded is rescored in fp64 from the fp32 rows.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// The int8 form of the scan (I8): rows and queries whose values are all integers inside one window of 256 values are
// kept as int8 (value minus the window's centre; the distance does not change), 64 bytes per step of
// v_mfma_i32_16x16x64_i8.  Products and sums are exact, so the filter needs no error terms: the accumulator starts at
// -floor(|x'|^2 / 2) and delivers d_int = q'.x' - floor(|x'|^2 / 2), which is the exact d = q'.x' - |x'|^2 / 2 or half
// a unit above it; the test d_int >= ceil(G) therefore lets through every row the exact test would, and
// |q'|^2 - 2 d_int + 1 bounds the squared distance from above.  Half the bytes per row and per query row of the fp16
// form, same tiles, units, thresholds and forward lists.
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <bool I8> struct ScanTypes { typedef f32x4 acc4; typedef float val; };
template <> struct ScanTypes<true> { typedef i32x4 acc4; typedef int32_t val; };
static constexpr int32_t I8_ABSENT = -(1 << 29);       // accumulator start of a row the tile does not have
static constexpr int32_t I8_NEVER = 1 << 29;           // threshold of a query slot that holds no query
static constexpr float I8_G_FLOOR = -268435456.0f;     // thresholds are kept above -2^28: every present row passes, no absent one

// Work units of one batch.  A unit is one 64-row tile of a touched bucket against at most
// UNIT_QUERIES of the queries that probe the bucket: {first id slot, rows | queries << 8, first
// slot in leaf_queries[], the tile's largest |x~| and |x - x~| as two 16-bit halves}.  The tile's rows are loaded once per
// unit and scored against the queries 16 at a time; buckets probed by very many queries become
// several units per tile, so no unit is long and a static interleave over the persistent waves
// balances the chip.
static constexpr uint32_t UNIT_QUERIES = 256;     // (64: c4 +7 %; 128: +1.5 %; 512: as 256; c3 indifferent from 128 on)
__device__ inline uint32_t units_of_leaf(uint32_t cnt, uint32_t ntl) { return cnt ? ntl * ((cnt + UNIT_QUERIES - 1u) / UNIT_QUERIES) : 0u; }

// The two exclusive scans the batch needs over the buckets, in one set of launches: lq_off[] over the
// (query, bucket) incidence counts and unit_off[] over the per-bucket unit counts (bucket order, so the
// unit list is the same for the same batch on every run); totals[0..1] = the two sums.
__device__ __forceinline__ void scan2_inputs(const uint32_t *lq_cnt, const uint32_t *leaf_tile0, int64_t i, int64_t n, int64_t &a, int64_t &b)
{
    a = 0; b = 0;
    if (i < n) {
        const uint32_t c = lq_cnt[i];
        a = c;
        b = units_of_leaf(c, leaf_tile0[i + 1] - leaf_tile0[i]);
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan2_sums(const uint32_t *__restrict__ lq_cnt, const uint32_t *__restrict__ leaf_tile0,
                                                             int64_t n, int64_t *__restrict__ offa, int64_t *__restrict__ offb)
{
    __shared__ int64_t sh[4];
    const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILE;
    int64_t sa = 0, sb = 0;
    for (int j = 0; j < SCAN_PER; j++) {
        int64_t a, b;
        scan2_inputs(lq_cnt, leaf_tile0, t0 + j * SCAN_THREADS + threadIdx.x, n, a, b);
        sa += a; sb += b;
    }
    sa = block_sum256(sa, sh);
    __syncthreads();
    sb = block_sum256(sb, sh);
    if (threadIdx.x == 0) { offa[t0] = sa; offb[t0] = sb; }
}

__global__ __launch_bounds__(1024) void k_scan2_tiles(int64_t n, int64_t ntiles, int64_t *__restrict__ offa, int64_t *__restrict__ offb,
                                                      int64_t *__restrict__ totals, int guarded, int64_t cap_a, int64_t cap_b,
                                                      int32_t *__restrict__ flags)
{
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (ntiles + 1023) / 1024;
    const int64_t b = tid * per, e = (b + per < ntiles) ? b + per : ntiles;
    for (int which = 0; which < 2; which++) {
        int64_t *off = which ? offb : offa;
        int64_t s = 0;
        for (int64_t i = b; i < e; i++) s += off[i * SCAN_TILE];
        __syncthreads();
        part[tid] = s;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int64_t v = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        int64_t run = part[tid] - s;
        for (int64_t i = b; i < e; i++) { const int64_t v = off[i * SCAN_TILE]; off[i * SCAN_TILE] = run; run += v; }
        if (tid == 1023) {
            off[n] = part[1023];
            totals[which] = part[1023];
            // the kernels behind this one were launched for buffers of cap_a / cap_b entries
            if (guarded && (part[1023] > (which ? cap_b : cap_a) || flags[1] != 0)) flags[3] = 1;
        }
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan2_apply(const uint32_t *__restrict__ lq_cnt, const uint32_t *__restrict__ leaf_tile0,
                                                              int64_t n, int64_t *__restrict__ offa, int64_t *__restrict__ offb,
                                                              uint32_t *__restrict__ offa32)
{
    __shared__ int64_t wsa[4], wsb[4];
    const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILE;
    const int64_t basea = offa[t0], baseb = offb[t0];
    // thread t owns inputs t*SCAN_PER .. t*SCAN_PER+SCAN_PER-1 of the tile
    int64_t va[SCAN_PER], vb[SCAN_PER], sa = 0, sb = 0;
    const int64_t i0 = t0 + (int64_t)threadIdx.x * SCAN_PER;
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++) { scan2_inputs(lq_cnt, leaf_tile0, i0 + j, n, va[j], vb[j]); sa += va[j]; sb += vb[j]; }
    int64_t ia = sa, ib = sb;                         // inclusive scans over the wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t ua = __shfl_up(ia, o), ub = __shfl_up(ib, o);
        if (lane >= o) { ia += ua; ib += ub; }
    }
    if (lane == 63) { wsa[w] = ia; wsb[w] = ib; }
    __syncthreads();                                  // also orders the reads of off[t0] before the writes below
    int64_t ra = basea + ia - sa, rb = baseb + ib - sb;
    for (int u = 0; u < w; u++) { ra += wsa[u]; rb += wsb[u]; }
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++) {
        if (i0 + j < n) { offa[i0 + j] = ra; offb[i0 + j] = rb; offa32[i0 + j] = (uint32_t)ra; }
        ra += va[j]; rb += vb[j];
    }
}

void launch_bucket_scans(const uint32_t *lq_cnt, const uint32_t *leaf_tile0, int64_t nl, int64_t *lq_off, int64_t *unit_off,
                         int64_t *totals, bool guarded, int64_t cap_pairs, int64_t cap_units, int32_t *flags, uint32_t *lq_off32,
                         hipStream_t stream)
{
    const int64_t ntiles = (nl + SCAN_TILE - 1) / SCAN_TILE;
    if (ntiles > 0) hipLaunchKernelGGL(k_scan2_sums, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, lq_cnt, leaf_tile0, nl, lq_off, unit_off);
    hipLaunchKernelGGL(k_scan2_tiles, dim3(1), dim3(1024), 0, stream, nl, ntiles, lq_off, unit_off, totals, guarded ? 1 : 0,
                       cap_pairs, cap_units, flags);
    if (ntiles > 0) hipLaunchKernelGGL(k_scan2_apply, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, lq_cnt, leaf_tile0, nl, lq_off, unit_off, lq_off32);
}

// ---------------------------------------------------------------------------
// k_bin_lists: one workgroup per bin of buckets.  The bin's (bucket, query) pairs (k_probe_lanes, binned form) give,
// all in LDS: the buckets' incidence counts, their exclusive scan (list offsets, on top of the pairs of the bins
// before), the buckets' work-unit counts and their scan, and the place of every pair in its bucket's list.
// A compute unit retires about one lane of scattered LDS atomics or 4-byte stores per cycle, so a workgroup's time is
// its bin's pairs: the bins are cut to hold about equally many (k_bin_seed).
// ---------------------------------------------------------------------------
constexpr int BL_THREADS = 1024;
constexpr int BL_UNROLL = 8;                                 // pairs per thread in flight
constexpr int BL_STAGE = 64000;                              // list entries (16-bit query numbers) a workgroup stages in LDS
constexpr size_t BL_LDS = (size_t)BIN_MAX_LEAVES * 4 + (size_t)BL_STAGE * 2;
static_assert(BL_STAGE * 2 >= BIN_MAX_LEAVES * 4, "the staging area also holds one word per bucket");
__global__ __launch_bounds__(BL_THREADS) void k_bin_lists(BinListParams p)
{
    extern __shared__ uint32_t s_bl[];
    uint32_t *const s_cnt = s_bl;                            // incidences of each bucket, then its fill cursor
    uint32_t *const s_word = s_bl + BIN_MAX_LEAVES;          // the staging area as words (unit offsets of the buckets) ...
    uint16_t *const s_stage = (uint16_t *)s_word;            // ... and as the bin's list entries
    __shared__ unsigned long long s_red[BL_THREADS / 64][2];
    __shared__ uint32_t s_wave[BL_THREADS / 64][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    if (p.flags[3]) return;                                  // a bin's region was too small: the batch is run again
    // pairs of all bins, and of the bins before this one
    unsigned long long tot = 0, before = 0;
    for (int i = tid; i < p.nbins; i += BL_THREADS) {
        const unsigned long long c = p.bin_cursor[i];
        tot += c;
        if (i < b) before += c;
    }
    for (int o = 32; o > 0; o >>= 1) { tot += __shfl_xor(tot, o); before += __shfl_xor(before, o); }
    if (lane == 0) { s_red[wave][0] = tot; s_red[wave][1] = before; }
    __syncthreads();
    tot = 0; before = 0;
    for (int w = 0; w < BL_THREADS / 64; w++) { tot += s_red[w][0]; before += s_red[w][1]; }
    if (tot > (unsigned long long)p.cap_pairs) {             // leaf_queries[] has no room (every workgroup sees the same)
        if (b == 0 && tid == 0) p.flags[3] = 1;
        return;
    }
    const uint32_t leaf0 = p.bin_lo[b];
    const uint32_t nleaf = min(p.bin_lo[b + 1] - leaf0, (uint32_t)BIN_MAX_LEAVES);
    const uint32_t n = p.bin_cursor[b];
    const uint2 *src = p.pairs + p.bin_off[b];
    for (uint32_t i = tid; i < (uint32_t)BIN_MAX_LEAVES; i += BL_THREADS) s_cnt[i] = 0;
    __syncthreads();
    // (several pairs per thread in flight: a workgroup has little else to hide the loads behind)
    for (uint32_t i0 = tid; i0 < n; i0 += BL_UNROLL * BL_THREADS) {
        uint32_t lf[BL_UNROLL];
#pragma unroll
        for (int u = 0; u < BL_UNROLL; u++) lf[u] = i0 + u * BL_THREADS < n ? src[i0 + u * BL_THREADS].x - leaf0 : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < BL_UNROLL; u++)
            if (lf[u] < nleaf) atomicAdd(&s_cnt[lf[u]], 1u);
    }
    __syncthreads();
    // exclusive scans over the bin's buckets: incidences and work units; thread t owns BIN_MAX_LEAVES / BL_THREADS buckets
    constexpr uint32_t per = BIN_MAX_LEAVES / BL_THREADS;
    const uint32_t l0 = tid * per;
    uint32_t cn[per], un[per];
    uint32_t sa = 0, su = 0;
#pragma unroll
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t l = l0 + j;
        cn[j] = 0; un[j] = 0;
        if (l < nleaf) {
            cn[j] = s_cnt[l];
            un[j] = units_of_leaf(cn[j], p.leaf_tile0[leaf0 + l + 1] - p.leaf_tile0[leaf0 + l]);
        }
        sa += cn[j]; su += un[j];
    }
    uint32_t ia = sa, iu = su;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ua = __shfl_up(ia, o), uu = __shfl_up(iu, o);
        if (lane >= o) { ia += ua; iu += uu; }
    }
    if (lane == 63) { s_wave[wave][0] = ia; s_wave[wave][1] = iu; }
    __syncthreads();
    uint32_t ra = ia - sa, ru = iu - su, all_u = 0;
    for (int w = 0; w < BL_THREADS / 64; w++) {
        if (w < wave) { ra += s_wave[w][0]; ru += s_wave[w][1]; }
        all_u += s_wave[w][1];
    }
#pragma unroll
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t l = l0 + j;
        if (l < nleaf) {
            s_cnt[l] = ra;                                   // (from here on: where the bucket's next list entry goes)
            s_word[l] = ru;
        }
        ra += cn[j]; ru += un[j];
    }
    if (tid == 0) {
        p.bin_units[b] = (int64_t)all_u;
        if (b == p.nbins - 1) p.lq_off[p.nl] = (int64_t)tot;
    }
    __syncthreads();
    for (uint32_t l = tid; l < nleaf; l += BL_THREADS) p.unit_off[leaf0 + l] = (int64_t)s_word[l];
    __syncthreads();
    // The lists.  A compute unit retires a scattered 4-byte store every other cycle, so the entries are put in place in
    // LDS and written out in order, whole lines at a time; a bin that outgrew the staging area (or query numbers past 16
    // bits) is placed in global memory directly.
    int32_t *out = p.leaf_queries + before;
    const bool staged = p.stage && n <= (uint32_t)BL_STAGE && p.nq <= 65536;
    for (uint32_t i0 = tid; i0 < n; i0 += BL_UNROLL * BL_THREADS) {
        uint2 e[BL_UNROLL];
#pragma unroll
        for (int u = 0; u < BL_UNROLL; u++) e[u] = i0 + u * BL_THREADS < n ? src[i0 + u * BL_THREADS] : make_uint2(0xffffffffu, 0u);
#pragma unroll
        for (int u = 0; u < BL_UNROLL; u++) {
            const uint32_t l = e[u].x - leaf0;
            if (e[u].x != 0xffffffffu && l < nleaf) {
                const uint32_t at = atomicAdd(&s_cnt[l], 1u);
                if (staged) s_stage[at] = (uint16_t)e[u].y;
                else out[at] = (int32_t)e[u].y;
            }
        }
    }
    __syncthreads();
    if (staged)
        for (uint32_t i = tid; i < n; i += BL_THREADS) out[i] = (int32_t)s_stage[i];
    // (every cursor now stands at its bucket's end = the next bucket's start)
    for (uint32_t l = tid; l < nleaf; l += BL_THREADS) {
        const uint32_t lo = l ? s_cnt[l - 1] : 0u;
        p.lq_cnt[leaf0 + l] = s_cnt[l] - lo;
        p.lq_off[leaf0 + l] = (int64_t)(before + lo);
    }
}

void launch_bin_lists(const BinListParams &p, hipStream_t stream)
{
    if (p.nbins > 0) hipLaunchKernelGGL(k_bin_lists, dim3((unsigned)p.nbins), dim3(BL_THREADS), BL_LDS, stream, p);
}

__global__ __launch_bounds__(1024) void k_bin_prefix(const uint32_t *__restrict__ bin_cursor, const int64_t *__restrict__ bin_units, int nbins,
                                                     int64_t cap_units, int64_t *__restrict__ bin_unit_base, int64_t *__restrict__ totals,
                                                     int32_t *__restrict__ flags)
{
    __shared__ long long part[1024];
    __shared__ long long pairs[1024];
    const int tid = threadIdx.x;
    if (flags[3]) return;
    const long long v = tid < nbins ? bin_units[tid] : 0;
    part[tid] = v;
    pairs[tid] = tid < nbins ? (long long)bin_cursor[tid] : 0;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long a = tid >= o ? part[tid - o] : 0, c = tid >= o ? pairs[tid - o] : 0;
        __syncthreads();
        part[tid] += a; pairs[tid] += c;
        __syncthreads();
    }
    if (tid < nbins) bin_unit_base[tid] = part[tid] - v;
    if (tid == 1023) {
        totals[0] = pairs[1023];
        totals[1] = part[1023];
        if (part[1023] > cap_units) flags[3] = 1;
    }
}

void launch_bin_prefix(const uint32_t *bin_cursor, const int64_t *bin_units, int32_t nbins, int64_t cap_units, int64_t *bin_unit_base,
                       int64_t *totals, int32_t *flags, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bin_prefix, dim3(1), dim3(1024), 0, stream, bin_cursor, bin_units, (int)nbins, cap_units, bin_unit_base, totals, flags);
}

// the incidences of every granule of buckets of a batch (counted into lq_cnt): a wave per granule
__global__ __launch_bounds__(256) void k_bin_granules(const uint32_t *__restrict__ lq_cnt, int64_t nl, int gran_shift, int ngran,
                                                      uint32_t *__restrict__ gran_cnt)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= ngran) return;
    const int64_t l0 = (int64_t)g << gran_shift, l1 = min(nl, l0 + ((int64_t)1 << gran_shift));
    uint32_t c = 0;
    for (int64_t l = l0 + lane; l < l1; l += 64) c += lq_cnt[l];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) gran_cnt[g] = c;
}

// One workgroup: the cut of the granules into bins.  With e(g) = the pairs of the granules before g, granule g gets the
// key e(g) / target, a run of granules with one key is cut every BIN_MAX_LEAVES buckets, and every piece is a bin: a bin
// holds at most target pairs and one granule's more (a granule larger than the target has a key to itself).  The target:
// the batch's incidences / 224 (a bin for nearly every compute unit), but no more than half of what k_bin_lists stages in
// LDS (a later batch may bring twice as many) - unless that made more than about 700 bins.  Three scans over the granules
// (pairs, start of the key's run, pieces), each thread owning eight consecutive granules, and one over the bins.
__global__ __launch_bounds__(1024) void k_bin_seed(const uint32_t *__restrict__ gran_cnt, int64_t nl, int gran_shift, int ngran,
                                                   uint16_t *__restrict__ bin_map, uint32_t *__restrict__ bin_lo,
                                                   uint32_t *__restrict__ bin_cap, uint32_t *__restrict__ bin_off,
                                                   unsigned long long *__restrict__ info)
{
    constexpr int PER = BIN_MAX_GRANULES / 1024;
    static_assert(BIN_MAX <= 1024, "one bin per thread in the last scan");
    __shared__ unsigned long long s_w[16];
    __shared__ uint32_t s_pairs[BIN_MAX], s_lo[BIN_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // exclusive scan over the workgroup's threads, sum or maximum; `all` = over every thread
    auto scan = [&](unsigned long long v, bool take_max, unsigned long long &all) {
        unsigned long long in = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long u = __shfl_up(in, o);
            if (lane >= o) in = take_max ? max(in, u) : in + u;
        }
        __syncthreads();                                     // (s_w may still be read from the scan before)
        if (lane == 63) s_w[wave] = in;
        __syncthreads();
        unsigned long long before = 0;
        all = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) before = take_max ? max(before, s_w[w]) : before + s_w[w];
            all = take_max ? max(all, s_w[w]) : all + s_w[w];
        }
        const unsigned long long prev = __shfl_up(in, 1);    // inclusive value of the lane before
        if (lane == 0) return before;
        return take_max ? max(before, prev) : before + prev;
    };
    for (int i = tid; i < BIN_MAX; i += 1024) { s_pairs[i] = 0; s_lo[i] = 0; }
    const int g0 = tid * PER;
    uint32_t c[PER];
    unsigned long long mine = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { c[j] = g0 + j < ngran ? gran_cnt[g0 + j] : 0u; mine += c[j]; }
    unsigned long long total;
    unsigned long long e = scan(mine, false, total);         // pairs before this thread's first granule
    unsigned long long target = max(total / 224ull, 1024ull);
    if (target > (unsigned long long)BL_STAGE / 2) target = max((unsigned long long)BL_STAGE / 2, total / 700ull);
    const int max_gran = BIN_MAX_LEAVES >> gran_shift;
    // where a key's run starts
    unsigned long long key[PER];
    bool first[PER];
    unsigned long long prev_key = 0;
    if (g0 > 0 && g0 <= ngran) prev_key = (e - gran_cnt[g0 - 1]) / target;
    unsigned long long last_start = 0;                       // (granule number + 1 of the last run start among this thread's)
#pragma unroll
    for (int j = 0; j < PER; j++) {
        key[j] = e / target;
        first[j] = g0 + j < ngran && (g0 + j == 0 || key[j] != prev_key);
        if (first[j]) last_start = (unsigned long long)(g0 + j) + 1ull;
        prev_key = key[j];
        e += c[j];
    }
    unsigned long long unused;
    unsigned long long run0 = scan(last_start, true, unused); // the last run start before this thread's granules (+ 1)
    // pieces of at most max_gran granules
    bool start[PER];
    unsigned long long nstart = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        if (first[j]) run0 = (unsigned long long)(g0 + j) + 1ull;
        start[j] = g0 + j < ngran && (first[j] || ((unsigned long long)(g0 + j) + 1ull - run0) % (unsigned long long)max_gran == 0);
        nstart += start[j] ? 1ull : 0ull;
    }
    unsigned long long nbins;
    unsigned long long bin = scan(nstart, false, nbins);     // bins that start before this thread's granules
    const bool fits = nbins > 0 && nbins <= (unsigned long long)BIN_MAX;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        if (g0 + j >= ngran) break;
        if (start[j]) {
            if (fits) s_lo[bin] = (uint32_t)(g0 + j) << gran_shift;
            bin++;
        }
        // (bin is now this granule's bin + 1)
        bin_map[g0 + j] = (uint16_t)(fits ? bin - 1 : 0);
        if (fits && c[j]) atomicAdd(&s_pairs[bin - 1], c[j]);
    }
    __syncthreads();
    const int nb = fits ? (int)nbins : 0;
    const unsigned long long cap = tid < nb ? 2ull * s_pairs[tid] + 1024ull : 0ull;
    unsigned long long total_cap;
    const unsigned long long off = scan(cap, false, total_cap);
    if (tid < nb) {
        bin_cap[tid] = (uint32_t)min(cap, 0xffffffffull);
        bin_off[tid] = (uint32_t)min(off, 0xffffffffull);
        bin_lo[tid] = s_lo[tid];
    }
    if (tid == 0) {
        if (nb > 0) bin_lo[nb] = (uint32_t)nl;
        info[0] = total_cap;
        info[1] = (unsigned long long)nb;
    }
}

void launch_bin_seed(const uint32_t *lq_cnt, int64_t nl, int32_t gran_shift, int32_t ngran, uint32_t *gran_cnt, uint16_t *bin_map,
                     uint32_t *bin_lo, uint32_t *bin_cap, uint32_t *bin_off, unsigned long long *info, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bin_granules, dim3((unsigned)((ngran + 3) / 4)), dim3(256), 0, stream, lq_cnt, nl, (int)gran_shift, (int)ngran, gran_cnt);
    hipLaunchKernelGGL(k_bin_seed, dim3(1), dim3(1024), 0, stream, gran_cnt, nl, (int)gran_shift, (int)ngran, bin_map, bin_lo, bin_cap, bin_off, info);
}

// one thread per tile (a long bucket has thousands of them)
__global__ __launch_bounds__(256) void k_units(const uint32_t *__restrict__ lq_cnt, const int64_t *__restrict__ lq_off,
                                               const uint32_t *__restrict__ leaf_tile0, const uint4 *__restrict__ tiles,
                                               int64_t ntiles, const int64_t *__restrict__ unit_off, uint4 *__restrict__ units,
                                               const int32_t *__restrict__ abort, const int64_t *__restrict__ bin_unit_base,
                                               const uint16_t *__restrict__ bin_map, int gran_shift)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles || (abort && *abort)) return;
    const uint4 td = tiles[t];
    const uint32_t l = td.x, cnt = lq_cnt[l];
    if (cnt == 0) return;
    const uint32_t groups = (cnt + UNIT_QUERIES - 1u) / UNIT_QUERIES;
    int64_t base = unit_off[l] + (int64_t)((uint32_t)t - leaf_tile0[l]) * groups;
    if (bin_unit_base) base += bin_unit_base[bin_map[l >> gran_shift]];
    const uint32_t lqb = (uint32_t)lq_off[l];
    for (uint32_t c = 0; c < cnt; c += UNIT_QUERIES)
        units[base++] = make_uint4(td.y, td.z | (min(UNIT_QUERIES, cnt - c) << 8), lqb + c, td.w);
}

// a non-negative float rounded UP to 8 significant bits, as the upper half of a word
__device__ inline uint32_t ceil16_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    if (u & 0xffffu) u = (u | 0xffffu) + 1u;
    return u & 0xffff0000u;
}

// 16-byte tails (8 fp16 values, see leafscan_f16_filter_dim): inner product in fp32 (exact products)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tail_dot(const uint4 &a, const uint4 &b)
{
    float c = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a.x), __builtin_bit_cast(f16x2, b.x), 0.f, false);
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a.y), __builtin_bit_cast(f16x2, b.y), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a.z), __builtin_bit_cast(f16x2, b.z), c, false);
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a.w), __builtin_bit_cast(f16x2, b.w), c, false);
}

// tiles[t].w = two upper bounds over the tile's resident rows, both rounded up to 8 significant bits:
// the largest norm of the fp16 copies (upper half) and the largest rounding distance (lower half).
// slot_norms[] = the squared norms again, in bucket order beside ids[]: the scan reads a tile's 64
// norms as one 256-byte run instead of 64 scattered 4-byte reads that each pull a whole line.
// bounds[0..2] (optional; non-negative floats, raised with integer atomicMax on their bit patterns): the largest of the
// tiles' two bounds and the largest squared norm of a row's tail - what a reader of the forward lists that does not
// know an entry's tile may assume of it.
__global__ __launch_bounds__(256) void k_tile_maxnorm(uint4 *__restrict__ tiles, int64_t ntiles, const int32_t *__restrict__ ids,
                                                      const float *__restrict__ norms, const float *__restrict__ errs,
                                                      int64_t row_begin, int64_t row_end, float *__restrict__ slot_norms,
                                                      const uint4 *__restrict__ tails, uint4 *__restrict__ slot_tails,
                                                      uint32_t *__restrict__ bounds)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float xr = 0.f, cx = 0.f, tm = 0.f;
    if (t < ntiles) {
        uint4 td = tiles[t];
        float m = 0.f, e = 0.f;
        for (uint32_t i = 0; i < td.z; i++) {
            const int64_t id = ids[td.y + i];
            float nn = 0.f;
            uint4 tl = make_uint4(0u, 0u, 0u, 0u);
            if (id >= row_begin && id < row_end) {
                nn = norms[id - row_begin]; m = fmaxf(m, nn); e = fmaxf(e, errs[id - row_begin]);
                if (tails) { tl = tails[id - row_begin]; tm = fmaxf(tm, tail_dot(tl, tl)); }
            }
            slot_norms[td.y + i] = nn;
            if (slot_tails) slot_tails[td.y + i] = tl;
        }
        td.w = ceil16_bits(sqrtf(m) * 1.00001f) | (ceil16_bits(e) >> 16);
        tiles[t] = td;
        xr = __uint_as_float(td.w & 0xffff0000u);
        cx = __uint_as_float(td.w << 16);
    }
    if (bounds) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            xr = fmaxf(xr, __shfl_xor(xr, o)); cx = fmaxf(cx, __shfl_xor(cx, o)); tm = fmaxf(tm, __shfl_xor(tm, o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMax(bounds, __float_as_uint(xr));
            atomicMax(bounds + 1, __float_as_uint(cx));
            atomicMax(bounds + 2, __float_as_uint(tm));
        }
    }
}

void launch_units(const uint32_t *lq_cnt, const int64_t *lq_off, const uint32_t *leaf_tile0, const uint4 *tiles, int64_t ntiles,
                  const int64_t *unit_off, uint4 *units, const int32_t *abort, hipStream_t stream, const int64_t *bin_unit_base,
                  const uint16_t *bin_map, int32_t gran_shift)
{
    if (ntiles > 0)
        hipLaunchKernelGGL(k_units, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, stream, lq_cnt, lq_off, leaf_tile0, tiles, ntiles,
                           unit_off, units, abort, bin_unit_base, bin_map, (int)gran_shift);
}
void launch_tile_maxnorm(uint4 *tiles, int64_t ntiles, const int32_t *ids, const float *norms, const float *errs, int64_t row_begin,
                         int64_t row_end, float *slot_norms, hipStream_t stream, const uint4 *tails, uint4 *slot_tails, uint32_t *bounds)
{
    if (ntiles > 0) hipLaunchKernelGGL(k_tile_maxnorm, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, stream, tiles, ntiles, ids, norms, errs, row_begin, row_end, slot_norms, tails, slot_tails, bounds);
}

// Bounds instrumentation of the scan's loaders (builds made with -DDPF_DEBUG_BOUNDS only): an index outside what the
// host says the arrays hold is counted and replaced by 0, so that a faulty index shows as a count, not as a memory fault.
#ifdef DPF_DEBUG_BOUNDS
template <typename I>
__device__ __forceinline__ I dbg_bound(const LeafScanParams &p, I idx, int64_t limit)
{
    if ((unsigned long long)(long long)idx < (unsigned long long)limit) return idx;
    atomicAdd(p.dbg_violations, 1ull);
    return (I)0;
}
#define DPF_BOUND(p, idx, limit) dbg_bound(p, idx, limit)
#else
#define DPF_BOUND(p, idx, limit) (idx)
#endif

// Ablation switches of the scan (timing experiments that give wrong answers on purpose) exist only in
// builds made with -DDPF_ABLATE=<bits> (1: loads and products only, 2: every query row from one hot spot,
// 4: no in-tile tightening); the shipped kernel has none.
#ifdef DPF_ABLATE
#define DPF_ABLATION(p, bit) (((DPF_ABLATE) & (bit)) != 0)
#else
#define DPF_ABLATION(p, bit) false
#endif

// Rounding allowance of the filter, per unit of |q~|^2 + |x~|^2: fp32 accumulation of K products on
// top of the row's share of the threshold (the accumulator starts at -J), the fp32 sums behind the
// two norms, and the roundings of the threshold arithmetic.
__device__ __forceinline__ float leafscan_c32(int stride) { return (5.0f * (float)(stride + 8) + 24.0f) * 5.9604645e-8f; }

// the largest of a lane's scores for accumulator register r over the tile's 16-row blocks
template <int RB>
__device__ __forceinline__ float best_of(const f32x4 (&acc)[RB], int r)
{
    if constexpr (RB == 4) return fmaxf(fmaxf(acc[0][r], acc[1][r]), fmaxf(acc[2][r], acc[3][r]));
    float m = acc[0][r];
#pragma unroll
    for (int rb = 1; rb < RB; rb++) m = fmaxf(m, acc[rb][r]);
    return m;
}
template <int RB>
__device__ __forceinline__ int32_t best_of(const i32x4 (&acc)[RB], int r)
{
    int32_t m = acc[0][r];
#pragma unroll
    for (int rb = 1; rb < RB; rb++) m = max(m, acc[rb][r]);
    return m;
}


// Coalesced operand loads (COAL).  The matrix cores want lane (g, i16) to hold bytes 16 g .. 16 g + 15 of row i16's
// 64-byte step, so a plain operand load has neighbouring lanes on 16 different rows: 64 cache look-ups per instruction.
// Loaded the other way round - lane quad j reads the 64 consecutive bytes of row pi(j) = 4 (j % 4) + j / 4, one look-up
// per quad - the same registers are a lane permutation away from operand shape: lane (g, m, y) takes what lane (y, m, g)
// loaded (the two 2-bit ends of the lane number change places), four ds_bpermute per 16-byte register, no LDS memory.
#ifndef LS8_COAL
#define LS8_COAL 1
#endif
#ifndef LS16_COAL
#define LS16_COAL 0
#endif
#ifndef LS16_SLICE_AHEAD
#define LS16_SLICE_AHEAD 0
#endif
__device__ __forceinline__ f16x8 operand_from_quads(const f16x8 &v, int src4)
{
    const i32x4 w = __builtin_bit_cast(i32x4, v);
    i32x4 o;
    o[0] = __builtin_amdgcn_ds_bpermute(src4, w[0]);
    o[1] = __builtin_amdgcn_ds_bpermute(src4, w[1]);
    o[2] = __builtin_amdgcn_ds_bpermute(src4, w[2]);
    o[3] = __builtin_amdgcn_ds_bpermute(src4, w[3]);
    return __builtin_bit_cast(f16x8, o);
}

// The queries of one unit against the tile a wave holds in registers, 16 at a time: thresholds,
// products on the matrix cores, tests, forwarding.  s_* are this wave's LDS areas: forwarded (query,
// id, bound) triples parked until 64 are there; per query slot the threshold G, and for the rare
// paths {trigger level G2, bits of the query index, l2: |q - q~| / dot: the slot's slack, |q~|^2}
// and the query id.
//
// Per query the scan reads one 16-byte record qc = {l2: A = |q~|^2 (1 - c32) / 2, dot: upper bound of
// |q~|; |q - q~|; |q~|^2; bits of the query id} and its threshold word tau:
//   l2:  tau holds R >= sqrt(k-th best squared distance) + |q - q~|.  With eta = |q - q~| + |x - x~|,
//        | |q~ - x~| - |q - x| | <= eta, so a row must be forwarded when |q~ - x~| <= R + |x - x~|max =: hi:
//        q~.x~ - |x~|^2 (1 - c32) / 2 >= A - hi^2 / 2 =: G.  The accumulator starts at -J(row) =
//        -|x~|^2 (1 - c32) / 2, so the matrix cores deliver d = q~.x~ - J and the test is d >= G.
//   dot: tau holds the k-th best product; |q.x - q~.x~| <= |q - q~||x~| + |q||x - x~|, forward when
//        q~.x~ + that slack >= tau.
// G2 is the level at which a row is roughly inside 0.985 of the k-th best distance; it only
// triggers the exact selection below, so it needs no rigour.
//
// TAIL (l2 only): the rows and query rows are cut at KS * 32 dimensions and xt[rb] / q_tail hold the up to 8
// dimensions behind the cut.  The matrix cores' d then belongs to the cut vectors, whose distance bounds the whole
// one from below: the test d >= G lets through every row the whole-row test would.  For the scores that pass it the
// tail's share t2 = |q_t - x_t|^2 = |q_t|^2 + |x_t|^2 - 2 q_t.x_t (fp32 from exact fp16 products; its rounding,
// at most a few fp32 ulps of |q_t|^2 + |x_t|^2, is covered by the slack 2e-6 (|q_t|^2 + |x_t|^2)) is put back:
// d - t2_lo / 2 is the d of the whole fp16 rows (rounded towards keeping the row), s_hi + t2_hi their squared
// distance from above.  Norms in the thresholds are the cut vectors', rounding distances the whole rows'.
template <int KS, int METRIC, int RB, bool TAIL, bool I8, bool COAL>
__device__ __forceinline__ void leafscan_chunks(const LeafScanParams &p, const uint16_t *__restrict__ Q16, const f16x8 (&xb)[RB][KS],
                                                const typename ScanTypes<I8>::val (&nJ)[RB], const int32_t (&idj)[RB], uint32_t rows, uint32_t nq,
                                                uint32_t lq0, float xrm, float cxm, int32_t qsc, int32_t qsn, int32_t *s_fq,
                                                int32_t *s_fid, float *s_fv, float4 *s_slot, float *s_G, int32_t *s_qid, int &nbuf,
                                                const uint4 *s_xt, uint4 *s_qt, float *s_qt2, float *s_sel, uint4 *s_q)
{
    static_assert(!TAIL || METRIC == 1, "tails belong to the l2 filter");
    static_assert(RB == 4 || RB == 1, "64-row tiles, or 16-row tiles for rows of more than 768 dimensions");
    static_assert(!I8 || (METRIC == 1 && RB == 4 && !TAIL && KS <= 4), "the int8 scan: l2, 64-row tiles, rows of up to 256 bytes");
    static_assert(!COAL || KS <= 8, "coalesced operand loads: whole query rows in registers");
    typedef typename ScanTypes<I8>::acc4 acc4;
    typedef typename ScanTypes<I8>::val val;
    // Rows of more than 8 steps: a 64-row tile still pays (one wave per SIMD holds it in the 512 registers
    // of its lanes; sharing a tile's rows among 4x the queries outweighs the lost occupancy: 1.4-1.5x
    // faster than 16-row tiles at 512 and 768 dimensions); beyond 24 steps only 16 rows fit (RB == 1).
    // Their query rows come in slices of QS steps, as long as the registers allow without spilling: a wave alone on its
    // SIMD waits out every slice's round trip (768 dimensions, slices of 4 / 6 / 8 steps: 5.72 / 5.27 / 5.46 ms per
    // launch on 1.25 M rows - the last one spills; 512 dimensions hold the whole query row: 3.63 -> 3.49 ms).
    constexpr int QS = KS > 8 ? (RB == 4 ? 4 : 8) : KS;
    static_assert(KS % QS == 0, "whole slices");
    // Rows of more than 16 steps: the chunk's 16 query rows are fetched whole into the wave's own LDS block by
    // global_load_lds (no registers; lane quad m fetches the 64 bytes of step t of the query in operand row m, so
    // instruction t fills the 1 KiB block of step t and lane (m, g) later reads slot 4 m + g of it: conflict-free) and
    // the slices come from there: one memory round trip per chunk instead of one per slice.
    constexpr bool QLDS = !I8 && RB == 4 && KS > 8;
    const int lane = threadIdx.x & 63;
    const int i16 = lane & 15, g = lane >> 4;
    const uint32_t jq = 4u * (uint32_t)(i16 & 3) + (uint32_t)(i16 >> 2);   // chunk position of this lane's MFMA row
    constexpr int STRIDE = KS * 32;                       // fp16 elements per padded row
    const float c32 = I8 ? 0.f : leafscan_c32(STRIDE);
    auto flush = [&](int n) {
        if (lane < n) {
            const int32_t fq = DPF_BOUND(p, s_fq[lane], p.dbg_nq);
            const int at = atomicAdd(&p.emit_cnt[fq], 1);
            if (at < p.emit_cap) {
                p.emit[(size_t)fq * p.emit_cap + at] = s_fid[lane];
                p.emit_v[(size_t)fq * p.emit_cap + at] = s_fv[lane];
            }
        }
    };

    // (QLDS) the 16 query rows named by `qs` (lane i16 holds operand row i16's query) on their way into the wave's LDS block
    auto stage_queries = [&](int32_t qs) {
        const int32_t ql = __shfl(qs, lane >> 2);
        const uint16_t *qr = Q16 + DPF_BOUND(p, (size_t)(ql >= 0 ? ql : 0), p.dbg_nq) * STRIDE + 8 * (lane & 3);
#pragma unroll
        for (int t = 0; t < KS; t++)
            __builtin_amdgcn_global_load_lds(qr + 32 * t, (__attribute__((address_space(3))) void *)(s_q + 64 * t), 16, 0, 0);
    };
    for (uint32_t c = 0; c < nq; c += 16u) {
        f16x8 qa[QS];
        const size_t qat = DPF_BOUND(p, (size_t)(qsc >= 0 ? qsc : 0), p.dbg_nq);
        const uint16_t *qrow = Q16 + (DPF_ABLATION(p, 2) ? (qat & 15) : qat) * STRIDE + 8 * g;   // debug 2: all query rows from one hot spot (timing only)
        if constexpr (QLDS) {
            if (c == 0) stage_queries(qsc);               // (later chunks: requested while the chunk before them was finishing)
        } else if constexpr (COAL) {
            // quad j reads the query at chunk position j; operand row i16 then holds position pi(i16) = jq
            const int lj = lane >> 2;
            const int32_t ql = __shfl(qsc, 4 * (lj & 3) + (lj >> 2));
            const uint16_t *qr = Q16 + DPF_BOUND(p, (size_t)(ql >= 0 ? ql : 0), p.dbg_nq) * STRIDE + 8 * (lane & 3);
            const int src4 = (((lane & 3) << 4) | (lane & 12) | (lane >> 4)) << 2;
            f16x8 raw[QS];
#pragma unroll
            for (int t = 0; t < QS; t++) raw[t] = *(const f16x8 *)(qr + 32 * t);
#pragma unroll
            for (int t = 0; t < QS; t++) qa[t] = operand_from_quads(raw[t], src4);
        } else {
#pragma unroll
            for (int t = 0; t < QS; t++) qa[t] = *(const f16x8 *)(qrow + 32 * t);
        }
        const float m_tau = dec_f32(__hip_atomic_load(&p.tau[qat], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const float4 qc = p.qc[qat];
        uint4 qtv = make_uint4(0u, 0u, 0u, 0u);
        if (TAIL && g == 0) qtv = p.q_tail[qat];
        const int32_t qcur = qsc;
        // query indices two chunks ahead
        qsc = qsn;
        qsn = c + 32u + jq < nq ? p.leaf_queries[DPF_BOUND(p, lq0 + c + 32u + jq, p.dbg_pairs)] : -1;

        // per-slot thresholds; lane i16 of every 16-lane group holds query slot i16
        {
            float G = HUGE_VALF, G2 = HUGE_VALF, aux = qc.y;
            if (qcur >= 0) {
                if (METRIC == 1) {
                    const float hi = m_tau + cxm;
                    G = fmaxf(qc.x - 0.5f * hi * hi * 1.000001f, -3.0e38f);
                    const float w = 0.985f * m_tau - 1.985f * qc.y - cxm;
                    if (w > 0.f) G2 = fmaxf(qc.x - 0.5f * w * w, -3.0e38f);
                } else {
                    aux = (qc.y * xrm + (qc.x + qc.y) * cxm + c32 * qc.x * xrm) * 1.00001f + 1e-30f;
                    G = fmaxf(m_tau - aux, -3.0e38f);
                }
            }
            if (g == 0) {
                if constexpr (I8) s_G[i16] = __int_as_float(qcur >= 0 ? (int32_t)ceilf(fmaxf(G, I8_G_FLOOR)) : I8_NEVER);
                else s_G[i16] = G;
                s_slot[i16] = make_float4(G2, __int_as_float(qcur), aux, qc.z);
                s_qid[i16] = __float_as_int(qc.w);
                if (TAIL) { s_qt[i16] = qtv; s_qt2[i16] = tail_dot(qtv, qtv); }
            }
        }
        acc4 acc[RB];
        if constexpr (I8) {
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                acc[rb] = (i32x4){nJ[rb], nJ[rb], nJ[rb], nJ[rb]};
                if (rb * 16u >= rows && rb > 0) continue;
#pragma unroll
                for (int t = 0; t < KS; t++)
                    acc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, qa[t]), __builtin_bit_cast(i32x4, xb[rb][t]), acc[rb], 0, 0, 0);
            }
        } else if constexpr (KS == QS) {
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                acc[rb] = (f32x4){nJ[rb], nJ[rb], nJ[rb], nJ[rb]};
                if (rb * 16u >= rows && rb > 0) continue;
#pragma unroll
                for (int t = 0; t < KS; t++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[t], xb[rb][t], acc[rb], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; rb++) acc[rb] = (f32x4){nJ[rb], nJ[rb], nJ[rb], nJ[rb]};
#if LS16_SLICE_AHEAD
            // the next slice of the query rows is requested before this slice's products (one wave per SIMD: nothing
            // else hides the round trip of a slice)
            f16x8 qn[QS];
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += QS) {
                if (s0 + QS < KS) {
#pragma unroll
                    for (int t = 0; t < QS; t++) qn[t] = *(const f16x8 *)(qrow + 32 * (s0 + QS + t));
                }
#pragma unroll
                for (int rb = 0; rb < RB; rb++) {
                    if (rb * 16u >= rows && rb > 0) continue;
#pragma unroll
                    for (int t = 0; t < QS; t++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[t], xb[rb][s0 + t], acc[rb], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < QS; t++) qa[t] = qn[t];
            }
#else
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += QS) {
                if constexpr (QLDS) {
                    if (s0 == 0) {
                        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the chunk's query rows have landed in LDS
                        asm volatile("" ::: "memory");
                    }
#pragma unroll
                    for (int t = 0; t < QS; t++) qa[t] = __builtin_bit_cast(f16x8, s_q[64 * (s0 + t) + 4 * i16 + g]);
                } else if (s0 > 0) {
#pragma unroll
                    for (int t = 0; t < QS; t++) qa[t] = *(const f16x8 *)(qrow + 32 * (s0 + t));
                }
#pragma unroll
                for (int rb = 0; rb < RB; rb++) {
                    if (rb * 16u >= rows && rb > 0) continue;
#pragma unroll
                    for (int t = 0; t < QS; t++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[t], xb[rb][s0 + t], acc[rb], 0, 0, 0);
                }
            }
            if constexpr (QLDS) {
                // the block has been read out: the next chunk's rows are requested now, beside this chunk's last products and tests
                if (c + 16u < nq) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the last slice has left LDS
                    stage_queries(qsc);
                }
            }
#endif
        }
        // lane l holds d = q~(MFMA row 4g + r) . x~(rb*16 + i16) - J in acc[rb][r]
        if (DPF_ABLATION(p, 1)) {                            // ablation runs: loads and products only
            float t = 0.f;
#pragma unroll
            for (int rb = 0; rb < RB; rb++) t += (float)acc[rb][rb];
            if (t == 12345.678f) nbuf = 1;
            continue;
        }
        // the common case first: nothing of this lane's 16 scores reaches its query's threshold
        const float4 Gs = *(const float4 *)(s_G + 4 * g);     // the thresholds of this lane's four query slots
        if constexpr (I8) {
            const int any = (int)(best_of<RB>(acc, 0) >= __float_as_int(Gs.x)) | (int)(best_of<RB>(acc, 1) >= __float_as_int(Gs.y)) |
                            (int)(best_of<RB>(acc, 2) >= __float_as_int(Gs.z)) | (int)(best_of<RB>(acc, 3) >= __float_as_int(Gs.w));
            if (__ballot(any != 0) == 0) continue;
        } else {
            const float e0 = best_of<RB>(acc, 0) - Gs.x;
            const float e1 = best_of<RB>(acc, 1) - Gs.y;
            const float e2 = best_of<RB>(acc, 2) - Gs.z;
            const float e3 = best_of<RB>(acc, 3) - Gs.w;
            if (__ballot(fmaxf(fmaxf(e0, e1), fmaxf(e2, e3)) >= 0.f) == 0) continue;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int slot = 4 * g + r;
            const float Gf = r == 0 ? Gs.x : r == 1 ? Gs.y : r == 2 ? Gs.z : Gs.w;
            val Gr;
            if constexpr (I8) Gr = __float_as_int(Gf); else Gr = Gf;
            if (__ballot(best_of<RB>(acc, r) >= Gr) == 0) continue;
            const float4 sc = s_slot[slot];           // {G2, query index, |q - q~| or slack, |q~|^2}
            const int32_t qq = __float_as_int(sc.y), qidq = s_qid[slot];
            // the tail's share of the squared distance of row block rb to this slot's query, from below and from above
            uint4 qt = make_uint4(0u, 0u, 0u, 0u);
            float qt2 = 0.f;
            if (TAIL) { qt = s_qt[slot]; qt2 = s_qt2[slot]; }
            auto tail2 = [&](int rb, float &lo, float &hi) {
                lo = 0.f; hi = 0.f;
                if (TAIL) {
                    const uint4 x = s_xt[rb * 16 + i16];
                    const float ab = qt2 + tail_dot(x, x);
                    const float t2 = fmaxf(fmaf(-2.0f, tail_dot(qt, x), ab), 0.f), slack = 2e-6f * ab;
                    lo = fmaxf(t2 - slack, 0.f);
                    hi = t2 + slack;
                }
            };
            int tight = 0;                            // this lane's rows inside the trigger level
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                // first stage: the matrix cores' score of the (cut) rows; second stage, for those that pass: the tail put back
                const bool first = acc[rb][r] >= Gr;
                if (__ballot(first) == 0) continue;
                float t2lo, t2hi;
                tail2(rb, t2lo, t2hi);
                const float d2 = fmaf(-0.5f, t2lo, (float)acc[rb][r]);     // (int8: the integer itself, exact in fp32)
                if (METRIC == 1) tight += first && d2 >= sc.x ? 1 : 0;
                const bool fwd = first && (I8 || d2 >= Gr) && !self_excluded(idj[rb], qidq, p.exclude_self);
                const unsigned long long m = __ballot(fwd);
                if (m) {
                    if (fwd) {
                        const int at = nbuf + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        s_fq[at] = qq;
                        s_fid[at] = idj[rb];
                        // l2: |q~ - x~|^2 <= |q~|^2 (1 + c32) + 2 c32 |x~|^2 - 2 d =: s_hi (d = q~.x~ - J, J = |x~|^2 (1 - c32) / 2,
                        // so 2 c32 |x~|^2 <= 4.001 c32 J),
                        // sqrt(exact distance) <= sqrt(s_hi) + |x - x~|max + |q - q~|, the query's share is added by
                        // the reader; dot: exact product >= q~.x~ - slack
                        if constexpr (I8) {
                            // d >= d_int - 1/2, so |q' - x'|^2 = |q'|^2 - 2 d <= |q'|^2 - 2 d_int + 1
                            const float s_hi = fmaf(-2.0f, (float)acc[rb][r], sc.w) + 1.0f;
                            s_fv[at] = sqrtf(fmaxf(s_hi, 0.f) * 1.000001f) * 1.000001f + cxm;
                        } else if (METRIC == 1) {
                            const float s_hi = fmaf(-2.0f, acc[rb][r], sc.w * (1.0f + c32) - 4.001f * c32 * nJ[rb]) + t2hi;
                            s_fv[at] = sqrtf(fmaxf(s_hi, 0.f) * 1.000001f) * 1.000001f + cxm;
                        } else {
                            s_fv[at] = acc[rb][r] - sc.z * 1.000001f;
                        }
                    }
                    nbuf += __popcll(m);
                    if (nbuf >= 64) {
                        flush(64);
                        const int rest = nbuf - 64;
                        int32_t tq2 = 0, ti2 = 0;
                        float tv2 = 0.f;
                        if (lane < rest) { tq2 = s_fq[64 + lane]; ti2 = s_fid[64 + lane]; tv2 = s_fv[64 + lane]; }
                        if (lane < rest) { s_fq[lane] = tq2; s_fid[lane] = ti2; s_fv[lane] = tv2; }
                        nbuf = rest;
                    }
                }
            }
            if (METRIC == 1 && __ballot(tight != 0)) {
                // rows of the 16-lane group (one query) inside the trigger level
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) tight += __shfl_xor(tight, o);
            }
            if (METRIC == 1 && !DPF_ABLATION(p, 4) && __ballot(tight >= p.k) != 0) {
                // Roughly k or more rows of this tile sit inside 0.985 of the k-th best distance (in some 16-lane group:
                // the wave goes through this together, groups below k take no part).  With
                //   v = sqrt(s_hi) + |x - x~|max >= sqrt(exact distance) - |q - q~|
                // the k-th smallest v among those rows bounds the query's k-th best distance from above (the k-th
                // smallest of any k or more of its rows does), and R = that + 2 |q - q~| is a new radius.
                const bool grp_on = tight >= p.k;
                // (lane-derived values of this rare path are made here, from a copy of the lane number the compiler cannot
                // trace: hoisted out of the chunk loop they would cost the whole kernel three registers)
                int l2 = lane;
                asm volatile("" : "+v"(l2));
                const int g2 = l2 >> 4, i2 = l2 & 15;
                float *const sel = s_sel + 64 * g2;
                // the group's rows inside the trigger level, packed into its 64 slots of s_sel in (row block, lane) order
                int m_g = 0;
#pragma unroll
                for (int rb = 0; rb < RB; rb++) {
                    float t2lo, t2hi;
                    tail2(rb, t2lo, t2hi);
                    const float d2 = fmaf(-0.5f, t2lo, (float)acc[rb][r]);
                    const bool mem = grp_on && idj[rb] >= 0 && acc[rb][r] >= Gr && d2 >= sc.x && !self_excluded(idj[rb], qidq, p.exclude_self);
                    float s_hi;
                    if constexpr (I8) s_hi = fmaf(-2.0f, (float)acc[rb][r], sc.w) + 1.0f;
                    else s_hi = fmaf(-2.0f, acc[rb][r], sc.w * (1.0f + c32) - 4.001f * c32 * nJ[rb]) + t2hi;
                    const uint32_t bits = (uint32_t)(__ballot(mem) >> (16 * g2)) & 0xffffu;
                    if (mem) sel[m_g + __popc(bits & ((1u << i2) - 1u))] = s_hi;     // (v grows with s_hi: ranked as it is)
                    m_g += __popc(bits);
                }
                // every packed value finds its rank among the group's: 16 values per round, one LDS broadcast per comparison
                for (int r0 = 0; __ballot(r0 < m_g) != 0; r0 += 16) {
                    const int my = r0 + i2;
                    const bool have = my < m_g;
                    const float mine = have ? sel[my] : HUGE_VALF;
                    int rank = 0;
                    for (int i = 0; __ballot(i < m_g) != 0; i++) {
                        const float o = i < m_g ? sel[i] : HUGE_VALF;
                        rank += (o < mine || (o == mine && i < my)) ? 1 : 0;
                    }
                    if (have && rank == p.k - 1) {
                        const float v = sqrtf(fmaxf(mine, 0.f) * 1.000001f) * 1.000001f + cxm;
                        atomicMin(&p.tau[DPF_BOUND(p, qq, p.dbg_nq)], enc_f32((v + 2.0f * sc.z) * 1.000002f));
                    }
                }
            }
        }
    }
}

// what is still parked when a wave runs out of units
__device__ __forceinline__ void leafscan_flush_rest(const LeafScanParams &p, const int32_t *s_fq, const int32_t *s_fid, const float *s_fv,
                                                    int nbuf)
{
    const int lane = threadIdx.x & 63;
    if (lane < nbuf) {
        const int32_t fq = DPF_BOUND(p, s_fq[lane], p.dbg_nq);
        const int at = atomicAdd(&p.emit_cnt[fq], 1);
        if (at < p.emit_cap) {
            p.emit[(size_t)fq * p.emit_cap + at] = s_fid[lane];
            p.emit_v[(size_t)fq * p.emit_cap + at] = s_fv[lane];
        }
    }
}

// Persistent waves drawing units from ticket counters.  A unit needs two dependent levels
// of loads (descriptor -> ids / query indices -> rows / query rows): descriptors run two units
// ahead and ids one, query indices one 16-query chunk ahead, so a wave waits once per unit (rows +
// first queries) and once per further chunk; the other waves of the SIMD fill those waits.
// RB*16 rows per tile: 64 measured best on gfx950 (32-row tiles double the per-unit work and
// were 1.9x slower in spite of the higher occupancy).
#ifndef LS8_WAVES
#define LS8_WAVES 4
#endif
constexpr int ls16_waves(int ks, int rb, bool i8 = false) { return i8 && ks <= 2 ? LS8_WAVES : ks > 8 && rb == 4 ? 1 : ks >= 8 ? 2 : LS16_WAVES4; }

template <int KS, int METRIC, int RB, bool TAIL, bool I8, bool COAL>
__device__ __forceinline__ void leafscan_body(const LeafScanParams &p, const uint16_t *__restrict__ Q16, const uint16_t *__restrict__ B16)
{
    if (p.gate && (*p.gate != 0) != (p.gate_want != 0)) return;     // the other form of the scan answers this batch
    __shared__ uint4 s_qt[4][TAIL ? 16 : 1];              // the chunk's query tails and their squared norms
    __shared__ float s_qt2[4][TAIL ? 16 : 1];
    __shared__ uint4 s_xt[4][TAIL ? 64 : 1];              // the tile's row tails
    __shared__ int32_t s_fq[4][128], s_fid[4][128];
    __shared__ float s_fv[4][128];
    __shared__ float4 s_slot[4][16];
    __shared__ __attribute__((aligned(16))) float s_G[4][16];
    __shared__ int32_t s_qid[4][16];
    extern __shared__ uint4 s_dyn[];                      // rows of more than 16 steps: 16 whole query rows per wave
    __shared__ float s_sel[4][METRIC == 1 ? 256 : 1];       // in-tile tightening: the rows of each 16-lane group inside the trigger level
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Units are handed out dynamically, so that a wave that drew long units takes fewer: NC ticket counters,
    // counter c serves the units u = unit_begin + t * NC + c (t = 0, 1, ...) to the workgroups with
    // blockIdx % NC == c.  A wave's first three tickets are fixed (its place among the counter's waves, plus
    // one and two rounds), later ones come from atomicAdd, fetched one unit ahead of the descriptor they are
    // needed for.  Tickets only grow, so a wave stops at its first unit past the end.
    uint32_t ubegin = (uint32_t)p.unit_begin, nunits = (uint32_t)p.unit_end;
    if (p.totals) {                                       // the slice comes from the device-side unit count
        if (*p.abort) return;
        const int64_t all = p.totals[1];
        ubegin = (uint32_t)(all * p.permille_lo / 1000);
        nunits = (uint32_t)(all * p.permille_hi / 1000);
    }
    const uint32_t nc = (uint32_t)p.num_counters, cidx = blockIdx.x % nc;
    const uint32_t place = (blockIdx.x / nc) * 4u + (uint32_t)wave;
    const uint32_t places = ((gridDim.x - cidx + nc - 1u) / nc) * 4u;          // waves served by this counter
    unsigned int *ctr = p.tickets + cidx * 32u;
    auto unit_of = [&](uint32_t t) {
        const unsigned long long uu = (unsigned long long)ubegin + (unsigned long long)t * nc + cidx;
        return uu < (unsigned long long)nunits ? (uint32_t)uu : nunits;
    };
    uint32_t u = unit_of(place), u_next = unit_of(place + places), u_after = unit_of(place + 2u * places);
    if (u >= nunits) return;
    uint32_t pend = 0;
    if (lane == 0) pend = atomicAdd(ctr, 1u);
    const int i16 = lane & 15, g = lane >> 4;
    // Query j of a 16-query chunk sits in MFMA row 4*(j%4) + j/4, so that the accumulator register
    // r of every lane belongs to queries 4r..4r+3 and a chunk of n queries needs only its first
    // ceil(n/4) registers tested; jq = the chunk position this lane's row holds.
    const uint32_t jq = 4u * (uint32_t)(i16 & 3) + (uint32_t)(i16 >> 2);
    constexpr int STRIDE = KS * 32;                       // fp16 elements per padded row
    const int32_t row_lo = (int32_t)p.row_begin, row_hi = (int32_t)p.row_end;

    int nbuf = 0;
    auto desc = [&](uint32_t uu) { return uu < nunits ? p.units[uu] : make_uint4(0u, 0u, 0u, 0u); };
    auto small = [&](const uint4 &U, int32_t (&id)[RB], int32_t &qs) {
        const uint32_t rows = U.y & 0xffu, nq = U.y >> 8;
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            const uint32_t idx = rb * 16 + i16;
            int32_t v = idx < rows ? p.ids[DPF_BOUND(p, U.x + idx, p.dbg_slots)] : -1;
            if (!(v >= row_lo && v < row_hi)) v = -1;
            id[rb] = v;
        }
        qs = jq < nq ? p.leaf_queries[DPF_BOUND(p, U.z + jq, p.dbg_pairs)] : -1;
    };

    uint4 U0 = desc(u), U1 = desc(u_next);
    int32_t id0[RB], qs0;
    small(U0, id0, qs0);

    for (; u < nunits;) {
        // this unit's rows
        f16x8 xb[RB][KS];
        typename ScanTypes<I8>::val nJ[RB];               // minus the row's share of the threshold, -|x~|^2 (1 - c32) / 2;
                                                          // -inf for an absent row, which then never reaches any threshold
        const uint32_t rows = U0.y & 0xffu;               // 16-row blocks past the tile's end are skipped throughout
        const uint32_t tile_slot0 = U0.x;
        if (TAIL) {
            // the tile's tails go straight to LDS (no registers): 16 lanes x 16 bytes per 16-row block, requested before
            // the rows, so that the wait for the rows (loads complete in order) covers them
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                if (rb * 16u >= rows && rb > 0) continue;
                if (g == 0 && rb * 16u + (uint32_t)i16 < rows)
                    __builtin_amdgcn_global_load_lds(p.slot_tail + DPF_BOUND(p, tile_slot0 + rb * 16u + (uint32_t)i16, p.dbg_slots),
                                                     (__attribute__((address_space(3))) void *)(s_xt[wave] + rb * 16), 16, 0, 0);
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            if (rb * 16u >= rows && rb > 0) continue;
            if constexpr (COAL) {
                // quad j reads row pi(j) of the block whole; the permutation leaves row i16 in operand column i16
                const int lj = lane >> 2;
                const int32_t idl = __shfl(id0[rb], 4 * (lj & 3) + (lj >> 2));
                const uint16_t *row = B16 + DPF_BOUND(p, (size_t)(idl >= 0 ? idl - row_lo : 0), p.dbg_rows) * STRIDE + 8 * (lane & 3);
#pragma unroll
                for (int t = 0; t < KS; t++) xb[rb][t] = *(const f16x8 *)(row + 32 * t);
            } else {
            // by id, or (bucket-order copy) by the slot the id sits in; lanes past the tile's end re-read its last row
            const size_t at = p.rows_in_slot_order ? DPF_BOUND(p, (size_t)(tile_slot0 + min(rb * 16u + (uint32_t)i16, rows - 1u)), p.dbg_slots)
                                                   : DPF_BOUND(p, (size_t)(id0[rb] >= 0 ? id0[rb] - row_lo : 0), p.dbg_rows);
            const uint16_t *row = B16 + at * STRIDE + 8 * g;
#pragma unroll
            for (int t = 0; t < KS; t++) xb[rb][t] = *(const f16x8 *)(row + 32 * t);
            }
            if constexpr (I8) nJ[rb] = rb * 16u + (uint32_t)i16 < rows ? p.slot_nj8[DPF_BOUND(p, tile_slot0 + rb * 16u + (uint32_t)i16, p.dbg_slots)] : 0;
            else nJ[rb] = rb * 16u + (uint32_t)i16 < rows ? p.slot_norm[DPF_BOUND(p, tile_slot0 + rb * 16u + (uint32_t)i16, p.dbg_slots)] : 0.f;
        }
        if constexpr (COAL) {
            const int src4 = (((lane & 3) << 4) | (lane & 12) | (lane >> 4)) << 2;
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                if (rb * 16u >= rows && rb > 0) continue;
#pragma unroll
                for (int t = 0; t < KS; t++) xb[rb][t] = operand_from_quads(xb[rb][t], src4);
            }
        }
        int32_t idj[RB];
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            idj[rb] = id0[rb];
            const bool there = id0[rb] >= 0 && (rb == 0 || rb * 16u < rows);
            if constexpr (I8) nJ[rb] = there ? nJ[rb] : I8_ABSENT;
            else nJ[rb] = there ? (METRIC == 1 ? -0.5f * nJ[rb] * (1.0f - leafscan_c32(STRIDE)) : 0.f) : -HUGE_VALF;
        }
        const uint32_t nq = U0.y >> 8, lq0 = U0.z;
        const float xrm = __uint_as_float(U0.w & 0xffff0000u);    // upper bound of the largest |x~| of the tile
        const float cxm = __uint_as_float(U0.w << 16);            // upper bound of the largest |x - x~|
        // the next unit's ids and first query indices, the descriptor after it
        // the second chunk's query indices ride along with the rows
        const int32_t qsc = qs0, qsn = 16u + jq < nq ? p.leaf_queries[DPF_BOUND(p, lq0 + 16u + jq, p.dbg_pairs)] : -1;
        U0 = U1;
        U1 = desc(u_after);
        u = u_next;
        u_next = u_after;
        u_after = unit_of(3u * places + (uint32_t)__builtin_amdgcn_readfirstlane((int)pend));
        if (lane == 0) pend = atomicAdd(ctr, 1u);
        small(U0, id0, qs0);
        leafscan_chunks<KS, METRIC, RB, TAIL, I8, COAL>(p, Q16, xb, nJ, idj, rows, nq, lq0, xrm, cxm, qsc, qsn, s_fq[wave], s_fid[wave], s_fv[wave],
                                              s_slot[wave], s_G[wave], s_qid[wave], nbuf, s_xt[wave], s_qt[wave], s_qt2[wave], s_sel[wave],
                                              s_dyn + (size_t)wave * (KS * 64));
    }
    leafscan_flush_rest(p, s_fq[wave], s_fid[wave], s_fv[wave], nbuf);
}

template <int KS, int METRIC, int RB, bool TAIL>
__global__ __launch_bounds__(256, ls16_waves(KS, RB)) void k_leafscan_f16(LeafScanParams p, const uint16_t *__restrict__ Q16,
                                                                          const uint16_t *__restrict__ B16)
{
    leafscan_body<KS, METRIC, RB, TAIL, false, (LS16_COAL != 0) && KS <= 4 && RB == 4>(p, Q16, B16);
}
// the int8 form: rows and query rows of KS * 64 bytes
template <int KS>
__global__ __launch_bounds__(256, ls16_waves(KS, 4, true)) void k_leafscan_i8(LeafScanParams p, const uint16_t *__restrict__ Q8,
                                                                              const uint16_t *__restrict__ B8)
{
    leafscan_body<KS, 1, 4, false, true, LS8_COAL != 0>(p, Q8, B8);
}

// dynamic LDS of the scan: the staged query rows of the forms with more than 16 steps (4 waves x 16 rows x KS x 64 bytes)
constexpr size_t leafscan_dyn_lds(int ks, int rb) { return rb == 4 && ks > 8 ? (size_t)4 * ks * 1024 : 0; }

template <int METRIC, int RB>
static void launch_leafscan_f16_m(const LeafScanParams &p, hipStream_t stream)
{
    const int64_t resident = (int64_t)p.num_cus * ls16_waves(p.ks, RB, p.i8 != 0);   // workgroups the chip holds at once
    const int64_t n = p.totals ? resident * 4 : p.unit_end - p.unit_begin;
    if (n <= 0) return;
    dim3 grid((unsigned)std::min<int64_t>((n + 3) / 4, resident)), block(256);
    LeafScanParams q = p;
    q.num_counters = (int32_t)std::min<int64_t>(p.num_counters, grid.x);     // every counter needs a workgroup
#define DPF_LB(KS)                                                                                                       \
    do {                                                                                                                 \
        if constexpr (METRIC == 1) {                                                                                     \
            if (q.slot_tail) { hipLaunchKernelGGL((k_leafscan_f16<KS, METRIC, RB, true>), grid, block, leafscan_dyn_lds(KS, RB), stream, q, q.Q16, q.base16); break; } \
        }                                                                                                                \
        hipLaunchKernelGGL((k_leafscan_f16<KS, METRIC, RB, false>), grid, block, leafscan_dyn_lds(KS, RB), stream, q, q.Q16, q.base16);         \
    } while (0)
    if constexpr (RB == 4 && METRIC == 1) {
        if (p.i8) {
#define DPF_LB8(KS) hipLaunchKernelGGL((k_leafscan_i8<KS>), grid, block, 0, stream, q, q.Q16, q.base16)
            switch (p.ks) {
            case 1: DPF_LB8(1); break;
            case 2: DPF_LB8(2); break;
            case 3: DPF_LB8(3); break;
            default: DPF_LB8(4); break;
            }
#undef DPF_LB8
            return;
        }
    }
    if constexpr (RB == 4) {
        switch (p.ks) {
        case 1: DPF_LB(1); break;
        case 2: DPF_LB(2); break;
        case 3: DPF_LB(3); break;
        case 4: DPF_LB(4); break;
        case 8: DPF_LB(8); break;
        case 16: DPF_LB(16); break;
        default: DPF_LB(24); break;
        }
    } else {
        DPF_LB(32);
    }
#undef DPF_LB
}

__global__ __launch_bounds__(256) void k_rows_to_slot_order(const int32_t *__restrict__ ids, int64_t nslots, const uint4 *__restrict__ base16,
                                                            int pieces, int64_t row_begin, int64_t row_end, uint4 *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nslots * pieces) return;
    const int64_t s = i / pieces;
    const int c = (int)(i - s * pieces);
    const int64_t id = ids[s];
    dst[i] = id >= row_begin && id < row_end ? base16[(id - row_begin) * pieces + c] : make_uint4(0u, 0u, 0u, 0u);
}

void launch_rows_to_slot_order(const int32_t *ids, int64_t nslots, const uint16_t *base16, int row_elems, int64_t row_begin,
                               int64_t row_end, uint16_t *dst, hipStream_t stream)
{
    const int pieces = row_elems / 8;
    const int64_t total = nslots * pieces;
    if (total > 0)
        hipLaunchKernelGGL(k_rows_to_slot_order, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ids, nslots,
                           (const uint4 *)base16, pieces, row_begin, row_end, (uint4 *)dst);
}

// k_refine: between two passes of the scan, tighten each query's threshold from what the first
// pass forwarded.  The k-th best of the pessimistic bounds of k DISTINCT rows bounds the query's
// k-th best exact score; one wave per query keeps the k best (bound, id) sorted across its lanes.
template <int METRIC>
__global__ __launch_bounds__(256) void k_refine(const int32_t *__restrict__ emit, const float *__restrict__ emit_v,
                                                const int32_t *__restrict__ emit_cnt, int32_t cap, int k, int64_t nq,
                                                const float *__restrict__ q16_err, int32_t *__restrict__ tau)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int n = min(emit_cnt[q], cap);
    if (n < k) return;
    float ls = METRIC == 1 ? HUGE_VALF : -HUGE_VALF;
    int32_t lid = -1;
    for (int base = 0; base < n; base += 64) {
        const bool live = base + lane < n;
        const float v = live ? emit_v[(size_t)q * cap + base + lane] : 0.f;
        const int32_t id = live ? emit[(size_t)q * cap + base + lane] : -1;
        const float kth = __shfl(ls, k - 1);
        unsigned long long m = __ballot(live && (METRIC == 1 ? v < kth : v > kth));
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
            const int32_t sid = __builtin_amdgcn_readlane(id, src);
            // one entry per row: a row forwarded from two buckets keeps its first bound
            if (__ballot(lane < k && lid == sid)) continue;
            const bool shift = lane < k && (METRIC == 1 ? s < ls : s > ls);
            const unsigned long long mask = __ballot(shift);
            if (mask == 0) continue;
            const int pos = __ffsll((long long)mask) - 1;
            const float ps = __shfl_up(ls, 1);
            const int32_t pid = __shfl_up(lid, 1);
            if (lane == pos) { ls = s; lid = sid; }
            else if (lane > pos && lane < k) { ls = ps; lid = pid; }
        }
    }
    if (lane == k - 1 && lid >= 0) {
        if (METRIC == 1) {
            // ls >= sqrt(exact distance) - |q - q~| for the k-th best of k distinct rows; the radius
            // the scan works with is sqrt(k-th best distance) + |q - q~|
            atomicMin(&tau[q], enc_f32((ls + 2.0f * q16_err[q]) * 1.000002f));
        } else {
            atomicMax(&tau[q], enc_f32(ls));
        }
    }
}

void launch_refine(int metric, const int32_t *emit, const float *emit_v, const int32_t *emit_cnt, int32_t cap, int k, int64_t nq,
                   const float *q16_err, int32_t *tau, hipStream_t stream)
{
    if (nq <= 0) return;
    dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (metric == 0) hipLaunchKernelGGL((k_refine<0>), grid, block, 0, stream, emit, emit_v, emit_cnt, cap, k, nq, q16_err, tau);
    else hipLaunchKernelGGL((k_refine<1>), grid, block, 0, stream, emit, emit_v, emit_cnt, cap, k, nq, q16_err, tau);
}

// fp16 steps (of 32 dimensions) a padded row takes: 1, 2, 3, 4, 8 (64-row tiles), 16, 24 or 32 (16-row tiles);
// -1 when the row is too long
int leafscan_f16_steps(int dim)
{
    const int ks = (dim + 31) / 32;
    if (ks <= 1) return 1;
    if (ks <= 2) return 2;
    if (ks <= 3) return 3;
    if (ks <= 4) return 4;
    if (ks <= 8) return 8;
    if (ks <= 16) return 16;
    if (ks <= 24) return 24;
    return ks <= 32 ? 32 : -1;
}

// Dimensions the fp16 rows of the scan keep (l2 metric).  The scan's rate is set by the 64-byte requests its rows
// cost, and a row of 100 dimensions padded to 128 costs four of them for little more than three requests' worth of
// data.  When at most 8 dimensions (and at most a sixteenth of the row) stick out over a row length the scan has a
// kernel for, the rows are cut there - 100 -> 96 (192-byte rows), 136 -> 128, 260 -> 256, 770 -> 768 - and the
// dimensions behind the cut travel as a 16-byte tail per row.  The squared distance over the leading dimensions
// bounds the whole one from below, so the matrix cores' test on the cut rows lets every row through that the test
// on whole rows would; for the few scores that pass it the tail's share |q_t - x_t|^2 is added on the vector ALU,
// which gives the whole-row fp16 distance again: the same filter, in two stages.
int leafscan_f16_filter_dim(int dim, int metric)
{
    if (metric != 1) return dim;
    const int t = dim / 32, rest = dim - 32 * t;
    if (t < 1 || rest == 0 || rest > 8 || rest * 16 > dim) return dim;
    const bool has_kernel = t == 1 || t == 2 || t == 3 || t == 4 || t == 8 || t == 16 || t == 24 || t == 32;
    return has_kernel ? 32 * t : dim;
}

// First threshold and the per-query record the scan reads (see leafscan_chunks), written by the lane that finishes a
// query row's conversion.  The exact k-th best score of the query's seed sample, moved to the safe side by the
// conversion to fp32 (l2: up, dot: down); l2 stores the radius sqrt(k-th best squared distance) + |q - q~|.  A
// query with fewer than k seeds starts without a threshold.
__device__ __forceinline__ void write_tau_value(const TauParams &t, int64_t q, float v, float qn, float qe, int stride)
{
    const float c32 = leafscan_c32(stride);
    t.tau[q] = enc_f32(v);
    t.qc[q] = make_float4(t.metric == 1 ? 0.5f * qn * (1.0f - c32) : sqrtf(qn) * 1.00001f, qe, qn,
                          __int_as_float(t.qids ? t.qids[q] : -1));
}
__device__ __forceinline__ void write_tau(const TauParams &t, int64_t q, float qn, float qe, int stride)
{
    // (no seed sample: every query starts without a threshold)
    const int32_t kth_id = t.seed_ids ? t.seed_ids[q * t.k + t.k - 1] : -1;
    const double kth = t.seed_ids ? t.seed_scores[q * t.k + t.k - 1] : 0.0;
    float v;
    if (t.metric == 1) v = kth_id < 0 ? HUGE_VALF : sqrtf(fmaxf((float)kth, 0.f) * 1.000001f) * 1.000001f + qe;
    else v = kth_id < 0 ? -HUGE_VALF : (float)kth - fabsf((float)kth) * 1e-6f - 1e-30f;
    write_tau_value(t, q, v, qn, qe, stride);
}

// fp32 rows (stride src_stride floats, `dim` meaningful) -> fp16 rows of dst_stride elements (a
// multiple of 32), round to nearest even, zero padded; norms[r] = sum of the squares of the ROUNDED
// elements of row r, accumulated in fp32; errs[r] = an upper bound of |row - rounded row| (the
// element differences are exact in fp32).  One lane converts 8 consecutive elements.
// nearest fp16 of x, clamped to the finite range, would-be subnormals flushed to zero; err2 gathers (x - fp16)^2
__device__ __forceinline__ uint16_t to_f16_measured(float x, float &sq, float &err2)
{
    _Float16 h = (_Float16)fminf(fmaxf(x, -65504.0f), 65504.0f);
    if (fabsf((float)h) < 6.103515625e-5f) h = (_Float16)0.0f;
    const float f = (float)h;
    sq = fmaf(f, f, sq);
    const float df = x - f;
    err2 = fmaf(df, df, err2);
    return __builtin_bit_cast(uint16_t, h);
}

// the dimensions [dim, dim_all) of row r (at most 8) as one 16-byte tail, zero padded; their rounding distance joins err2
__device__ __forceinline__ void tail_to_f16(const float *__restrict__ row, int dim, int dim_all, uint4 *__restrict__ out, float &err2,
                                            const float *__restrict__ mu, float &shifted2)
{
    uint16_t o[8];
    float unused = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        o[e] = 0;
        if (dim + e < dim_all) {
            const float x = mu ? row[dim + e] - mu[dim + e] : row[dim + e];
            shifted2 = fmaf(x, x, shifted2);
            o[e] = to_f16_measured(x, unused, err2);
        }
    }
    *out = make_uint4(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16), o[4] | ((uint32_t)o[5] << 16), o[6] | ((uint32_t)o[7] << 16));
}

// mu (optional): a centre subtracted from every row before the conversion (l2 handles: distances do not change, the
// rounding distances shrink with the values).  The subtraction is one fp32 operation per element; its own rounding, at
// most 2^-24 of the shifted value, joins the measured distance as 1.2e-7 |shifted row|.
template <bool TAU>
__global__ void k_rows_to_f16(const float *__restrict__ src, int64_t src_stride, int dim, int64_t n,
                               uint16_t *__restrict__ dst, int dst_stride, float *__restrict__ norms, float *__restrict__ errs,
                               TauParams tp, int dim_all, uint4 *__restrict__ tails, const float *__restrict__ mu)
{
    const int per_row = dst_stride >> 3;                  // lanes per row: a power of two up to 64
    const int64_t total = n * per_row;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < total;
    const int64_t r = live ? i / per_row : 0;
    const int c0 = (int)(i - r * per_row) * 8;
    uint16_t o[8];
    float ss = 0.f, ee = 0.f, xx = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        o[e] = 0;
        if (live && c0 + e < dim) {
            float x = src[r * src_stride + c0 + e];
            if (mu) x -= mu[c0 + e];
            xx = fmaf(x, x, xx);
            // nearest fp16, clamped to the finite range; what would be subnormal becomes 0 so that
            // the matrix cores see exactly the values the norms and distances are taken from
            _Float16 h = (_Float16)fminf(fmaxf(x, -65504.0f), 65504.0f);
            if (fabsf((float)h) < 6.103515625e-5f) h = (_Float16)0.0f;
            o[e] = __builtin_bit_cast(uint16_t, h);
            const float f = (float)h;
            ss = fmaf(f, f, ss);
            const float df = x - f;
            ee = fmaf(df, df, ee);
        }
    }
    if (live) {
        uint4 w;
        w.x = o[0] | ((uint32_t)o[1] << 16); w.y = o[2] | ((uint32_t)o[3] << 16);
        w.z = o[4] | ((uint32_t)o[5] << 16); w.w = o[6] | ((uint32_t)o[7] << 16);
        *(uint4 *)(dst + i * 8) = w;
        if (tails && c0 == 0) tail_to_f16(src + r * src_stride, dim, dim_all, tails + r, ee, mu, xx);
    }
    for (int m = 1; m < per_row; m <<= 1) { ss += __shfl_xor(ss, m); ee += __shfl_xor(ee, m); xx += __shfl_xor(xx, m); }   // rows never straddle a wave
    if (live && c0 == 0) {
        const float er = sqrtf(ee * 1.0001f) * 1.00001f + (mu ? 1.2e-7f * sqrtf(xx) : 0.f);
        norms[r] = ss; errs[r] = er;
        if (TAU) write_tau(tp, r, ss, er, dst_stride);
    }
}
// The same for rows of more than 512 elements (or a stride that is no power of two): one wave per row,
// a lane converts 8 consecutive elements of every 512.
template <bool TAU>
__global__ __launch_bounds__(256) void k_rows_to_f16_long(const float *__restrict__ src, int64_t src_stride, int dim, int64_t n,
                                                          uint16_t *__restrict__ dst, int dst_stride, float *__restrict__ norms,
                                                          float *__restrict__ errs, TauParams tp, int dim_all, uint4 *__restrict__ tails,
                                                          const float *__restrict__ mu)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;                                   // whole waves leave together
    float ss = 0.f, ee = 0.f, xx = 0.f;
    if (tails && lane == 0) tail_to_f16(src + r * src_stride, dim, dim_all, tails + r, ee, mu, xx);
    for (int c0 = lane * 8; c0 < dst_stride; c0 += 512) {
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[e] = 0;
            if (c0 + e < dim) {
                float x = src[r * src_stride + c0 + e];
                if (mu) x -= mu[c0 + e];
                xx = fmaf(x, x, xx);
                _Float16 h = (_Float16)fminf(fmaxf(x, -65504.0f), 65504.0f);
                if (fabsf((float)h) < 6.103515625e-5f) h = (_Float16)0.0f;
                o[e] = __builtin_bit_cast(uint16_t, h);
                const float f = (float)h;
                ss = fmaf(f, f, ss);
                const float df = x - f;
                ee = fmaf(df, df, ee);
            }
        }
        uint4 w;
        w.x = o[0] | ((uint32_t)o[1] << 16); w.y = o[2] | ((uint32_t)o[3] << 16);
        w.z = o[4] | ((uint32_t)o[5] << 16); w.w = o[6] | ((uint32_t)o[7] << 16);
        *(uint4 *)(dst + r * dst_stride + c0) = w;
    }
    for (int m = 1; m < 64; m <<= 1) { ss += __shfl_xor(ss, m); ee += __shfl_xor(ee, m); xx += __shfl_xor(xx, m); }
    if (lane == 0) {
        const float er = sqrtf(ee * 1.0001f) * 1.00001f + (mu ? 1.2e-7f * sqrtf(xx) : 0.f);
        norms[r] = ss; errs[r] = er;
        if (TAU) write_tau(tp, r, ss, er, dst_stride);
    }
}
// column sums of the rows (the centre the fp16 copies of an l2 handle are taken around): a workgroup adds 256 rows,
// thread c the columns c, c + 256, ..., then one atomic per column and workgroup
__global__ __launch_bounds__(256) void k_col_sums(const float *__restrict__ src, int64_t src_stride, int dim, int64_t n, float *__restrict__ sums)
{
    const int64_t r0 = (int64_t)blockIdx.x * 256, r1 = r0 + 256 < n ? r0 + 256 : n;
    for (int c = threadIdx.x; c < dim; c += 256) {
        float a = 0.f;
        for (int64_t r = r0; r < r1; r++) a += src[r * src_stride + c];
        atomicAdd(&sums[c], a);
    }
}
void launch_col_sums(const float *src, int64_t src_stride, int dim, int64_t n, float *sums, hipStream_t stream)
{
    if (n > 0) hipLaunchKernelGGL(k_col_sums, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, src_stride, dim, n, sums);
}

// ---------------------------------------------------------------------------
// k_seed_radius: first thresholds of an l2 batch from the fp16 copies
// ---------------------------------------------------------------------------
// The scan only needs an UPPER bound of each query's k-th best distance to start from, and the fp16 copies give
// one at half the bytes of the fp32 rows (or less): with S = |q~ - x~|^2 (fp32 sum of squared differences, all terms
// non-negative: relative error at most (terms + 16) 2^-24), sqrt(exact distance) <= sqrt(S) + |x - x~| + |q - q~|.
// One wave per query, 4 lanes per seed row (a lane reads the 16-byte piece 4 t + c of every 32-dimension step t: the
// four lanes of a row read 64 consecutive bytes), 32 rows in flight; the wave keeps the k smallest
// u = sqrt(S) + |x - x~| of distinct ids sorted across its lanes; radius R = k-th smallest u + 2 |q - q~|
// (what the scan's threshold word holds: R >= sqrt(k-th best squared distance) + |q - q~|).  Fewer than k seeds: no
// threshold.  Also writes the per-query record of the scan (as the TAU form of the row conversion does).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float sqdiff8(const uint4 &a, const uint4 &b, float acc)
{
    const f16x8_t x = __builtin_bit_cast(f16x8_t, a), y = __builtin_bit_cast(f16x8_t, b);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float d = (float)x[e] - (float)y[e];
        acc = fmaf(d, d, acc);
    }
    return acc;
}
__device__ __forceinline__ void topk_insert_f(float s, int32_t id, float &ls, int32_t &lid, int K, int lane)
{
    const bool shift = (lane < K) && (s < ls || (s == ls && id < lid));
    const unsigned long long mask = __ballot(shift);
    if (mask == 0) return;
    if (__ballot((lane < K) && lid == id)) return;        // the same row reached through two buckets: once
    const int pos = __ffsll((long long)mask) - 1;
    const float ps = __shfl_up(ls, 1);
    const int32_t pid = __shfl_up(lid, 1);
    if (lane == pos) { ls = s; lid = id; }
    else if (lane > pos && lane < K) { ls = ps; lid = pid; }
}

struct SeedRadiusParams {
    const int32_t *cand;      // [nq][cand_stride] seed ids
    const int32_t *cnt;       // [nq]
    int64_t cand_stride;
    const uint16_t *Q16, *B16;   // fp16 rows of ks * 32 elements
    const float *q_norm, *q_err; // [nq] squared norms and rounding distances of the query copies
    const float *b_err;          // [rows] rounding distances of the row copies
    const uint4 *q_tail, *b_tail; // tails of cut rows (nullptr: whole rows)
    int64_t row_begin, row_end;
    int32_t ks;
    TauParams tau;
};

template <int KSQ>   // > 0: ks == KSQ, the query row stays in registers; 0: any ks, query pieces re-read (they stay in L1)
__global__ __launch_bounds__(256) void k_seed_radius(SeedRadiusParams p, int64_t nq)
{
#ifndef SEED_UR
#define SEED_UR 2
#endif
    constexpr int UR = SEED_UR;
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int grp = lane >> 2, gl = lane & 3;
    const int K = p.tau.k;
    const int ks = KSQ > 0 ? KSQ : p.ks;
    const uint4 *qrow = (const uint4 *)(p.Q16 + (size_t)q * ks * 32) + gl;
    uint4 qv[KSQ > 0 ? KSQ : 1];
    if (KSQ > 0) {
#pragma unroll
        for (int t = 0; t < KSQ; t++) qv[t] = qrow[4 * t];
    }
    uint4 qt = make_uint4(0u, 0u, 0u, 0u);
    if (p.q_tail) qt = p.q_tail[q];
    float ls = HUGE_VALF;
    int32_t lid = 0x7fffffff;
    int32_t cnt = p.cnt[q];
    if (cnt > p.cand_stride) cnt = (int32_t)p.cand_stride;
    const int32_t *cand = p.cand + q * p.cand_stride;
    // S <= computed (1 + (terms + 16) 2^-24): 32 ks elements per lane and the tail, three additions across and inside lanes
    const float up = 1.0f + (float)(32 * ks + 8 + 16) * 6.3e-8f;

    int32_t idn[UR];
#pragma unroll
    for (int u = 0; u < UR; u++) idn[u] = u * 16 + grp < cnt ? cand[u * 16 + grp] : -1;
    for (int32_t b0 = 0; b0 < cnt; b0 += 16 * UR) {
        int32_t id[UR];
        float acc[UR], xe[UR];
#pragma unroll
        for (int u = 0; u < UR; u++) {
            id[u] = idn[u];
            if (!(id[u] >= p.row_begin && id[u] < p.row_end)) id[u] = -1;
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const int32_t idx = b0 + 16 * UR + u * 16 + grp;
            idn[u] = idx < cnt ? cand[idx] : -1;
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const size_t at = (size_t)(id[u] >= 0 ? id[u] - p.row_begin : 0);
            const uint4 *row = (const uint4 *)(p.B16 + at * ks * 32) + gl;
            float a = 0.f;
            if (KSQ > 0) {
                uint4 x[KSQ > 0 ? KSQ : 1];
#pragma unroll
                for (int t = 0; t < KSQ; t++) x[t] = row[4 * t];
#pragma unroll
                for (int t = 0; t < KSQ; t++) a = sqdiff8(qv[t], x[t], a);
            } else {
                for (int t = 0; t < ks; t += 4) {
                    uint4 x[4], y[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) { x[i] = t + i < ks ? row[4 * (t + i)] : make_uint4(0u, 0u, 0u, 0u); y[i] = t + i < ks ? qrow[4 * (t + i)] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
                    for (int i = 0; i < 4; i++) a = sqdiff8(y[i], x[i], a);
                }
            }
            if (p.b_tail && gl == 0) a = sqdiff8(qt, p.b_tail[at], a);
            xe[u] = p.b_err[at];
            acc[u] = a;
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            float a = acc[u];
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            const float uu = sqrtf(a * up) * 1.000001f + xe[u];
            const float kth = __shfl(ls, K - 1);
            const int32_t kth_id = __shfl(lid, K - 1);
            const bool cand_ok = gl == 0 && id[u] >= 0 && (uu < kth || (uu == kth && id[u] < kth_id));
            unsigned long long m = __ballot(cand_ok);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                topk_insert_f(__shfl(uu, src), __shfl(id[u], src), ls, lid, K, lane);
            }
        }
    }
    const float kth = __shfl(ls, K - 1);
    const int32_t kth_id = __shfl(lid, K - 1);
    if (lane == 0) {
        const float qe = p.q_err[q];
        const float v = kth_id == 0x7fffffff ? HUGE_VALF : (kth + 2.0f * qe) * 1.000002f;
        write_tau_value(p.tau, q, v, p.q_norm[q], qe, ks * 32);
    }
}

void launch_seed_radius(const int32_t *cand, const int32_t *cnt, int64_t cand_stride, const uint16_t *Q16, const uint16_t *B16, int ks,
                        const float *q_norm, const float *q_err, const float *b_err, const uint4 *q_tail, const uint4 *b_tail,
                        int64_t row_begin, int64_t row_end, const TauParams &tau, int64_t nq, hipStream_t stream)
{
    if (nq <= 0) return;
    SeedRadiusParams p;
    p.cand = cand; p.cnt = cnt; p.cand_stride = cand_stride; p.Q16 = Q16; p.B16 = B16; p.q_norm = q_norm; p.q_err = q_err;
    p.b_err = b_err; p.q_tail = b_tail ? q_tail : nullptr; p.b_tail = q_tail ? b_tail : nullptr;
    p.row_begin = row_begin; p.row_end = row_end; p.ks = ks; p.tau = tau;
    const dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    switch (ks) {
    case 1: hipLaunchKernelGGL((k_seed_radius<1>), grid, block, 0, stream, p, nq); break;
    case 2: hipLaunchKernelGGL((k_seed_radius<2>), grid, block, 0, stream, p, nq); break;
    case 3: hipLaunchKernelGGL((k_seed_radius<3>), grid, block, 0, stream, p, nq); break;
    case 4: hipLaunchKernelGGL((k_seed_radius<4>), grid, block, 0, stream, p, nq); break;
    case 8: hipLaunchKernelGGL((k_seed_radius<8>), grid, block, 0, stream, p, nq); break;
    default: hipLaunchKernelGGL((k_seed_radius<0>), grid, block, 0, stream, p, nq); break;
    }
}

void launch_rows_to_f16(const float *src, int64_t src_stride, int dim, int64_t n, uint16_t *dst, int dst_stride,
                         float *norms, float *errs, hipStream_t stream, const TauParams *tau, int dim_all, uint4 *tails, const float *mu)
{
    if (n <= 0) return;
    if (!tails || dim_all <= dim) { tails = nullptr; dim_all = dim; }
    const int per_row = dst_stride / 8;
    const TauParams tp = tau ? *tau : TauParams();
    if (per_row > 64 || (per_row & (per_row - 1)) != 0) {
        const dim3 grid((unsigned)((n + 3) / 4)), block(256);
        if (tau) hipLaunchKernelGGL((k_rows_to_f16_long<true>), grid, block, 0, stream, src, src_stride, dim, n, dst, dst_stride, norms, errs, tp, dim_all, tails, mu);
        else hipLaunchKernelGGL((k_rows_to_f16_long<false>), grid, block, 0, stream, src, src_stride, dim, n, dst, dst_stride, norms, errs, tp, dim_all, tails, mu);
        return;
    }
    const int64_t total = n * (dst_stride / 8);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (tau) hipLaunchKernelGGL((k_rows_to_f16<true>), grid, block, 0, stream, src, src_stride, dim, n, dst, dst_stride, norms, errs, tp, dim_all, tails, mu);
    else hipLaunchKernelGGL((k_rows_to_f16<false>), grid, block, 0, stream, src, src_stride, dim, n, dst, dst_stride, norms, errs, tp, dim_all, tails, mu);
}

// ---------------------------------------------------------------------------
// int8 copies for the exact integer form of the scan
// ---------------------------------------------------------------------------
// smallest and largest value of the rows and whether any value is no integer (NaN counts as one)
__global__ __launch_bounds__(256) void k_i8_range(const float *__restrict__ src, int64_t src_stride, int dim, int64_t n,
                                                  int32_t *__restrict__ out)
{
    float lo = HUGE_VALF, hi = -HUGE_VALF;
    int nonint = 0;
    const int64_t total = n * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim;
        const float x = src[r * src_stride + (i - r * dim)];
        lo = fminf(lo, x); hi = fmaxf(hi, x);
        nonint |= !(x == rintf(x)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o));
        nonint |= __shfl_xor(nonint, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(out, enc_f32(lo));
        atomicMax(out + 1, enc_f32(hi));
        if (nonint) atomicOr(out + 2, 1);
    }
}
void launch_i8_range(const float *src, int64_t src_stride, int dim, int64_t n, int32_t *out, hipStream_t stream)
{
    if (n <= 0) return;
    const int64_t total = n * dim;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_i8_range, dim3(blocks), dim3(256), 0, stream, src, src_stride, dim, n, out);
}

// 16 lanes per row, a lane converts 16 consecutive elements (rows of up to 256 bytes)
template <bool QUERY>
__global__ __launch_bounds__(256) void k_rows_to_i8(const float *__restrict__ src, int64_t src_stride, int dim, int64_t n, float offset,
                                                    int8_t *__restrict__ dst, int dst_stride, int32_t *__restrict__ norm2,
                                                    float4 *__restrict__ qc, const int32_t *__restrict__ qids, int32_t *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = i >> 4;
    const int c0 = (int)(i & 15) * 16;
    const bool live = r < n && c0 < dst_stride;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    int32_t sum = 0, wrong = 0;
    if (live) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            int32_t v = 0;
            if (c0 + e < dim) {
                const float f = src[r * src_stride + c0 + e] - offset;
                if (!(f == rintf(f)) || f < -128.0f || f > 127.0f) wrong = 1;
                else v = (int32_t)f;
            }
            sum += v * v;
            w[e >> 2] |= (uint32_t)(v & 0xff) << (8 * (e & 3));
        }
        *(uint4 *)(dst + r * dst_stride + c0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { sum += __shfl_xor(sum, o); wrong |= __shfl_xor(wrong, o); }
    if (r < n && c0 == 0) {
        if (wrong) atomicOr(bad, 1);
        if (QUERY) qc[r] = make_float4(0.5f * (float)sum, 0.f, (float)sum, __int_as_float(qids ? qids[r] : -1));
        else norm2[r] = sum;
    }
}
void launch_rows_to_i8(const float *src, int64_t src_stride, int dim, int64_t n, float offset, int8_t *dst, int dst_stride,
                       int32_t *norm2, float4 *qc, const int32_t *qids, int32_t *bad, hipStream_t stream)
{
    if (n <= 0) return;
    const dim3 grid((unsigned)((n * 16 + 255) / 256)), block(256);
    if (qc) hipLaunchKernelGGL((k_rows_to_i8<true>), grid, block, 0, stream, src, src_stride, dim, n, offset, dst, dst_stride, norm2, qc, qids, bad);
    else hipLaunchKernelGGL((k_rows_to_i8<false>), grid, block, 0, stream, src, src_stride, dim, n, offset, dst, dst_stride, norm2, qc, qids, bad);
}

__global__ __launch_bounds__(256) void k_slot_nj8(const int32_t *__restrict__ ids, int64_t nslots, const int32_t *__restrict__ norm2,
                                                  int64_t row_begin, int64_t row_end, int32_t *__restrict__ slot_nj8)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nslots) return;
    const int64_t id = ids[s];
    slot_nj8[s] = id >= row_begin && id < row_end ? -(norm2[id - row_begin] >> 1) : I8_ABSENT;
}
void launch_slot_nj8(const int32_t *ids, int64_t nslots, const int32_t *norm2, int64_t row_begin, int64_t row_end,
                     int32_t *slot_nj8, hipStream_t stream)
{
    if (nslots > 0) hipLaunchKernelGGL(k_slot_nj8, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, stream, ids, nslots, norm2, row_begin, row_end, slot_nj8);
}

void launch_leafscan(int metric, const LeafScanParams &p, hipStream_t stream)
{
    if (leafscan_tile_rows(p.ks) == 16) {
        if (metric == 0) launch_leafscan_f16_m<0, 1>(p, stream);
        else launch_leafscan_f16_m<1, 1>(p, stream);
        return;
    }
    if (metric == 0) launch_leafscan_f16_m<0, 4>(p, stream);
    else launch_leafscan_f16_m<1, 4>(p, stream);
}

// rows per tile of the scan for rows of `ks` fp16 steps
int leafscan_tile_rows(int ks) { return ks > 24 ? 16 : 64; }

// flag[0] = number of queries whose forward list overflowed, their indices in list[]; flag[2] = the longest forward
// list of the batch.  One workgroup (the batch's counts are a few tens of KB); its last act is to copy the 40-byte
// record at flag[] (four flag words, two scan totals, the int8 gate word) into pinned host memory, where the host reads it after its wait
// on the stream: no separate device-to-host copy.
__global__ __launch_bounds__(1024) void k_overflow_list(const int32_t *__restrict__ emit_cnt, int64_t nq, int32_t cap, int32_t *flag,
                                                        int32_t *__restrict__ list, int32_t *__restrict__ host_record)
{
    int32_t m = 0;
    for (int64_t q = threadIdx.x; q < nq; q += 1024) {
        const int32_t c = emit_cnt[q];
        if (c > cap) list[atomicAdd(flag, 1)] = (int32_t)q;
        m = max(m, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(flag + 2, m);
    __threadfence();
    __syncthreads();
    if (threadIdx.x < 10) {
        host_record[threadIdx.x] = __hip_atomic_load(flag + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
    }
}

void launch_overflow_list(const int32_t *emit_cnt, int64_t nq, int32_t cap, int32_t *flag, int32_t *list, int32_t *host_record,
                          hipStream_t stream)
{
    hipLaunchKernelGGL(k_overflow_list, dim3(1), dim3(1024), 0, stream, emit_cnt, nq, cap, flag, list, host_record);
}

// The queries named in list[] copied out as a batch of their own (16-byte pieces of the padded rows,
// and the query ids when there are any) ...
__global__ void k_gather_queries(const int32_t *__restrict__ list, int64_t n, const uint4 *__restrict__ Q, int64_t pieces,
                                 const int32_t *__restrict__ qids, uint4 *__restrict__ subQ, int32_t *__restrict__ sub_qids)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * pieces) return;
    const int64_t r = i / pieces, c = i - r * pieces;
    const int64_t q = list[r];
    subQ[i] = Q[q * pieces + c];
    if (c == 0 && qids) sub_qids[r] = qids[q];
}
// ... and their results put back in place
__global__ void k_scatter_results(const int32_t *__restrict__ list, int64_t n, int k, const int32_t *__restrict__ sub_ids,
                                  const double *__restrict__ sub_scores, int32_t *__restrict__ ids, double *__restrict__ scores)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * k) return;
    const int64_t r = i / k, e = i - r * k;
    const int64_t q = list[r];
    ids[q * k + e] = sub_ids[i];
    scores[q * k + e] = sub_scores[i];
}
void launch_gather_queries(const int32_t *list, int64_t n, const void *Q, int64_t row_bytes, const int32_t *qids, void *subQ,
                           int32_t *sub_qids, hipStream_t stream)
{
    const int64_t total = n * (row_bytes / 16);
    if (total > 0) hipLaunchKernelGGL(k_gather_queries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, list, n, (const uint4 *)Q,
                                      row_bytes / 16, qids, (uint4 *)subQ, sub_qids);
}
void launch_scatter_results(const int32_t *list, int64_t n, int k, const int32_t *sub_ids, const double *sub_scores, int32_t *ids,
                            double *scores, hipStream_t stream)
{
    const int64_t total = n * k;
    if (total > 0) hipLaunchKernelGGL(k_scatter_results, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, list, n, k, sub_ids,
                                      sub_scores, ids, scores);
}

// k_unique_lists: the scan forwards a row once per tree it was found in (near neighbours sit in
// many trees); before the exact rerank gathers 400-byte rows for them, each query's forward list is
// reduced to its distinct ids.  One workgroup per query: the ids go into an open-addressing table
// in LDS (twice the list length, a power of two), then the occupied entries are written back over
// the list (order does not matter: the rerank orders by score, then id).
//
// With `pr` (the forward lists after the scan) an entry is also left out when the query's FINAL threshold has passed
// it by: it was forwarded under a looser threshold, and by the filter's guarantee - which holds for every valid
// threshold, the final one included - a row the final threshold would not forward cannot be among the best k.  The
// list keeps one number per entry, the scan's pessimistic bound v; what the scan tested is recovered from it on the
// safe side with what holds for every tile of the handle (PruneParams):
//   l2 fp16: v = sqrt(s_hi) + cxm(tile), s_hi = |q~|^2 (1 + c32) + 2.0005 c32 (1 - c32) |x~|^2 - 2 d (+ t2_hi), and the
//            test was d >= A - (R + cxm)^2 / 2, i.e. R >= sqrt(2 (A - d)) - cxm with 2 (A - d) = s_hi - delta,
//            delta <= 2 c32 |q~|^2 + 2.0005 c32 max|x~|^2 + 4e-6 (|q_t|^2 + max|x_t|^2);
//   l2 int8: the same with delta = 1 (s_hi = |q'|^2 - 2 d_int + 1) and no rounding distances;
//   dot:     v = q~.x~ - slack, forwarded while q~.x~ + slack >= tau, slack <= what the largest tile bounds give.
__global__ __launch_bounds__(256) void k_unique_lists(int32_t *__restrict__ emit, const int32_t *__restrict__ emit_cnt, int32_t cap,
                                                      int32_t *__restrict__ unique_cnt, PruneParams pr)
{
    extern __shared__ int32_t tab[];
    __shared__ int32_t s_out;
    const int64_t q = blockIdx.x;
    const int n = min(emit_cnt[q], cap);
    int32_t *list = emit + (size_t)q * cap;
    if (n <= 1) { if (threadIdx.x == 0) unique_cnt[q] = n; return; }
    // per query: keep an entry iff v <= v_keep (l2) / v >= v_keep (dot)
    float v_keep = pr.metric == 1 ? HUGE_VALF : -HUGE_VALF;
    if (pr.emit_v) {
        const float t_final = dec_f32(pr.tau[q]);
        const bool i8 = pr.gate && *pr.gate == 0;
        if (pr.metric == 1) {
            // forwarded at the final radius iff sqrt(s_hi - delta) - cxm <= R, and sqrt(s_hi) >= (v - cx_max) (1 - 3e-6):
            // keep iff (v - cx_max)(1 - 3e-6) <= sqrt((R + cx_max)^2 + delta), every rounding given to the keeping side
            const float4 c = i8 ? pr.qc8[q] : pr.qc[q];
            float delta;
            if (i8) delta = 1.0f + 2e-6f * c.z;
            else {
                float qt2 = 0.f;
                if (pr.q_tail) { const uint4 t = pr.q_tail[q]; qt2 = tail_dot(t, t); }
                delta = (2.0f * pr.c32 * c.z + 2.0005f * pr.c32 * pr.xr_max * pr.xr_max + 4e-6f * (qt2 + pr.xt_max)) * 1.001f + 3e-6f * (c.z + pr.xr_max * pr.xr_max);
            }
            const float cx = pr.cx_max;
            if (t_final < 3.0e37f) {
                const float r = (t_final + cx) * 1.00001f;
                v_keep = sqrtf(r * r + delta) * 1.00001f + cx * 1.00001f + 1e-30f;
            }
        } else {
            // forwarded at the final product iff q~.x~ + slack >= tau; q~.x~ = v + slack (1.000001): keep iff v + 2.00001 slack_max >= tau
            const float4 c = pr.qc[q];
            const float slack = (c.y * pr.xr_max + (c.x + c.y) * pr.cx_max + pr.c32 * c.x * pr.xr_max) * 1.0001f + 1e-30f;
            if (t_final > -3.0e37f) v_keep = t_final - 2.0001f * slack - fabsf(t_final) * 1e-5f - 1e-30f;
        }
    }
    int T = 2;
    while (T < 2 * n) T <<= 1;
    for (int i = threadIdx.x; i < T; i += 256) tab[i] = -1;
    if (threadIdx.x == 0) s_out = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int32_t id = list[i];                       // ids are row numbers: never -1
        if (pr.emit_v) {
            const float v = pr.emit_v[(size_t)q * cap + i];
            if (pr.metric == 1 ? v > v_keep : v < v_keep) continue;
        }
        uint32_t h = ((uint32_t)id * 2654435761u) & (uint32_t)(T - 1);
        for (;;) {
            const int32_t old = atomicCAS(&tab[h], -1, id);
            if (old == -1 || old == id) break;
            h = (h + 1) & (uint32_t)(T - 1);
        }
    }
    __syncthreads();                                      // the list has been read; it is overwritten below
    const int lane = threadIdx.x & 63;
    for (int i0 = (threadIdx.x & ~63); i0 < T; i0 += 256) {
        const int32_t id = tab[i0 + lane];                // T is a multiple of 64 or smaller than the first stride
        const bool have = i0 + lane < T && id != -1;
        const unsigned long long m = __ballot(have);
        if (m == 0) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_out, (int)__popcll(m));
        base = __shfl(base, 0);
        if (have) list[base + __popcll(m & ((1ull << lane) - 1ull))] = id;
    }
    __syncthreads();
    if (threadIdx.x == 0) unique_cnt[q] = s_out;
}

// LDS bytes the table of a list of `longest` ids needs
size_t unique_lists_lds(int32_t longest)
{
    size_t T = 2;
    while (T < 2 * (size_t)std::max(longest, 1)) T <<= 1;
    return T * 4;
}

// longest = the longest list of the batch (at most cap): it sizes the table, and with it how many
// workgroups a CU holds
void launch_unique_lists(int32_t *emit, const int32_t *emit_cnt, int32_t cap, int32_t longest, int32_t *unique_cnt, int64_t nq,
                         hipStream_t stream, const PruneParams *prune)
{
    if (nq <= 0) return;
    const size_t T = unique_lists_lds(std::min(longest, cap)) / 4;
    const PruneParams pr = prune ? *prune : PruneParams();
    hipLaunchKernelGGL(k_unique_lists, dim3((unsigned)nq), dim3(256), (size_t)T * 4, stream, emit, emit_cnt, cap, unique_cnt, pr);
}
float leafscan_c32_of(int stride) { return (5.0f * (float)(stride + 8) + 24.0f) * 5.9604645e-8f; }

// ---------------------------------------------------------------------------
// k_merge_topk: one thread per query merges `parts` sorted lists of k entries
// ---------------------------------------------------------------------------
template <int METRIC>
__global__ void k_merge_topk(const int32_t *__restrict__ in_ids, const double *__restrict__ in_scores,
                             int parts, int64_t ids_stride, int64_t scores_stride, int64_t nq, int k,
                             int32_t *__restrict__ out_ids, double *__restrict__ out_scores)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int head[64];
    for (int s = 0; s < parts; s++) head[s] = 0;
    for (int e = 0; e < k; e++) {
        int best = -1;
        double bs = 0; int32_t bi = 0;
        for (int s = 0; s < parts; s++) {
            if (head[s] >= k) continue;
            const int64_t at = q * k + head[s];
            const int32_t id = in_ids[s * ids_stride + at];
            if (id < 0) { head[s] = k; continue; }
            const double sc = in_scores[s * scores_stride + at];
            if (best < 0 || better<METRIC>(sc, id, bs, bi)) { best = s; bs = sc; bi = id; }
        }
        if (best < 0) { out_ids[q * k + e] = -1; out_scores[q * k + e] = nan(""); continue; }
        out_ids[q * k + e] = bi;
        out_scores[q * k + e] = bs;
        // lists of table shards can hold the same row (same exact score): it sits at the head of
        // every list that has it, and is taken once
        for (int s = 0; s < parts; s++)
            if (head[s] < k && in_ids[s * ids_stride + q * k + head[s]] == bi) head[s]++;
    }
}

// list s of the ids starts ids_stride elements behind list s-1, of the scores scores_stride elements
void launch_merge_topk(int metric, const int32_t *in_ids, const double *in_scores, int parts, int64_t ids_stride,
                       int64_t scores_stride, int64_t nq, int k, int32_t *out_ids, double *out_scores, hipStream_t stream)
{
    if (nq <= 0) return;
    unsigned grid = (unsigned)((nq + 127) / 128);
    if (metric == 0)
        hipLaunchKernelGGL((k_merge_topk<0>), dim3(grid), dim3(128), 0, stream, in_ids, in_scores, parts, ids_stride, scores_stride, nq, k, out_ids, out_scores);
    else
        hipLaunchKernelGGL((k_merge_topk<1>), dim3(grid), dim3(128), 0, stream, in_ids, in_scores, parts, ids_stride, scores_stride, nq, k, out_ids, out_scores);
}

// Dynamic LDS above 64 KiB has to be opted into per kernel.
int configure_kernels()
{
    const int max_lds = 160 * 1024;
    hipError_t e;
    e = hipFuncSetAttribute((const void *)k_hash<float, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void *)k_hash<double, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void *)k_hash<float, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void *)k_hash<double, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void *)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return (int)e;
    {
        const void *long_rows[] = {(const void *)k_leafscan_f16<24, 0, 4, false>, (const void *)k_leafscan_f16<24, 1, 4, false>,
                                   (const void *)k_leafscan_f16<24, 1, 4, true>, (const void *)k_leafscan_f16<16, 0, 4, false>,
                                   (const void *)k_leafscan_f16<16, 1, 4, false>, (const void *)k_leafscan_f16<16, 1, 4, true>};
        for (const void *f : long_rows) {
            e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)leafscan_dyn_lds(24, 4));
            if (e != hipSuccess) return (int)e;
        }
    }
    e = hipFuncSetAttribute((const void *)k_unique_lists, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void *)k_bin_lists, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BL_LDS);
    if (e != hipSuccess) return (int)e;
    return 0;
}

// copies rows [n][dim] of T into the padded base store
template <typename T>
__global__ void k_copy_rows(const T *__restrict__ src, int64_t n, int dim, T *__restrict__ dst, int64_t dst_stride)
{
    const int64_t total = n * dst_stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dst_stride;
        const int c = (int)(i - r * dst_stride);
        dst[i] = c < dim ? src[r * dim + c] : (T)0;
    }
}

void launch_copy_rows(int dtype, const void *src, int64_t n, int dim, void *dst, int64_t dst_stride, hipStream_t stream)
{
    if (n <= 0) return;
    int64_t total = n * dst_stride;
    unsigned grid = (unsigned)((total + 255) / 256 < 65535 * 4 ? (total + 255) / 256 : 65535 * 4);
    if (dtype == 0) hipLaunchKernelGGL((k_copy_rows<float>), dim3(grid), dim3(256), 0, stream, (const float *)src, n, dim, (float *)dst, dst_stride);
    else hipLaunchKernelGGL((k_copy_rows<double>), dim3(grid), dim3(256), 0, stream, (const double *)src, n, dim, (double *)dst, dst_stride);
}

}  // namespace dpf
