// Device-side construction of the forest snapshot from the hashes of all ids.
//
// The reference grows every table by inserting ids one at a time
// (RandomDrawTreeMap.java:1558-1584, putInner 1662-1790): descend the directory
// nodes by NBITS-wide slots of the hash; a bucket that already holds
// BUCKET_OVERFLOW ids and sits at level >= 1 is replaced by ONE directory a
// level down when the next id arrives (RandomDrawTreeMap.java:1719-1768), the
// split does not cascade.  With ids inserted in ascending order
// (DensevectorRDFInit.newFastFit, DensevectorRDFInit.scala:127-151) the final
// shape is a function of the (hash, id) multiset:
//
//   a bucket at level v holding n ids, c0 of which were already there when its
//   parent directory was created, stays a bucket iff v == 0, n <= B or n == c0;
//   otherwise it became a directory when its (max(c0, B) + 1)-th id arrived,
//   and that id is the "present at creation" threshold of its children.
//
// Here the rule is applied to all tables at once, one trie level per round:
// the (table, sub-index, segment, trie bits) keys are kept sorted by the bits
// consumed so far with ids ascending inside every segment (a stable radix sort
// per round refines the order by one more slot), so the order statistics the
// rule needs -- "the m-th smallest id of the segment", "how many ids <= t" --
// are one array read and one binary search.  A segment that stays a bucket is
// frozen: its remaining key bits are cleared, so later rounds leave it alone.
// Segment start positions never move once the segment exists, which lets
// per-position arrays carry the bookkeeping (directories created at a
// position, bucket flag, parent of the bucket).
//
// Numbering equals a depth-first walk in key order (directories: pre-order),
// the same as the host builder in dpf_forest.cpp, so both produce identical
// arrays.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "dpf_build.hpp"

namespace dpf {

namespace {

constexpr int32_t FROZEN = -2;          // "thr" value of ids whose bucket is final
constexpr uint32_t ROOT = 0xffffffffu;  // parent marker of root directories

struct Geo {
    int32_t L, nbits, dns, max_level, lowbits, bucket_bits, seg, num_parts, B;
};

struct Rec {            // one directory node
    uint32_t start;     // first position of its segment
    uint32_t rank;      // directories created earlier at the same position (its ancestors there)
    uint32_t pstart;    // parent directory: its position (ROOT: none) ...
    uint32_t pinfo;     // ... and (rank << 16 | slot); for roots: (table * P + part) * SEG + segment
};

#define BCHECK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            err = std::string("device forest build: ") + #call + ": " + hipGetErrorString(e_); \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)

__global__ void k_build_keys(const int32_t *__restrict__ hashes, const int32_t *__restrict__ parts, int64_t total, Geo g,
                             uint64_t *__restrict__ key, uint64_t *__restrict__ val)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int64_t i = e / g.L;
    const int32_t t = (int32_t)(e - i * g.L);
    const uint32_t h = (uint32_t)hashes[e];
    const uint64_t p = (uint64_t)(uint32_t)parts[e];
    const uint64_t sg = h >> g.bucket_bits;
    const uint64_t lowmask = g.lowbits >= 32 ? 0xffffffffull : ((1ull << g.lowbits) - 1);
    const uint64_t grp = ((uint64_t)t * g.num_parts + p) * (uint64_t)g.seg + sg;
    key[e] = (grp << g.lowbits) | ((uint64_t)h & lowmask);
    val[e] = ((uint64_t)0xffffffffu << 32) | (uint64_t)(uint32_t)i;   // thr = -1: under a root directory
}

__global__ void k_head_pos(const uint64_t *__restrict__ key, int64_t total, int shift, uint32_t *__restrict__ hp)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const bool head = e == 0 || (key[e] >> shift) != (key[e - 1] >> shift);
    hp[e] = head ? (uint32_t)e : 0u;
}

// first position in [lo, hi) whose key >> shift exceeds pre
__device__ inline int64_t upper_of(const uint64_t *key, int64_t lo, int64_t hi, int shift, uint64_t pre)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((key[mid] >> shift) <= pre) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first position in [lo, hi) whose key >> shift is at least pre
__device__ inline int64_t lower_of(const uint64_t *key, int64_t lo, int64_t hi, int shift, uint64_t pre)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((key[mid] >> shift) < pre) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct RoundOut {
    int32_t *dec;             // per position: decision of the segment starting there (FROZEN or the children's threshold)
    const uint32_t *cnt_dn;   // directories created so far per position
    uint32_t *leaf_flag, *leaf_pa, *leaf_pb;
    Rec *recs;
    unsigned long long *counters;   // [0] records, [1] misflag events, [2] max bucket
    int64_t rec_cap;
};

// One thread per id; only the first id of a live segment works.  level = the trie level whose
// slot bits the segment's PARENT directory consumes (max_level + 1: the segment is a root group).
__global__ void k_decide(const uint64_t *__restrict__ key, const uint64_t *__restrict__ val, int64_t total, Geo g, int level,
                         RoundOut o)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int shift = level * g.nbits;
    const uint64_t pre = key[e] >> shift;
    if (e > 0 && (key[e - 1] >> shift) == pre) return;
    const int32_t thr = (int32_t)(val[e] >> 32);
    if (thr == FROZEN) return;
    const int64_t end = upper_of(key, e + 1, total, shift, pre);
    const int64_t n = end - e;
    if (level == g.max_level + 1) {
        // a root group: always a directory (RandomDrawTreeMap.java:1680-1690 creates it on first use)
        o.dec[e] = -1;
        const unsigned long long r = atomicAdd(&o.counters[0], 1ull);
        if ((int64_t)r < o.rec_cap) o.recs[r] = Rec{(uint32_t)e, 0u, ROOT, (uint32_t)pre};
        return;
    }
    // ids <= thr were present when the parent directory was created
    int64_t lo = e, hi = end;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(int32_t)(uint32_t)val[mid] <= (int64_t)thr) lo = mid + 1; else hi = mid;
    }
    const int64_t c0 = lo - e;
    const bool has_thr = c0 > 0 && (int32_t)(uint32_t)val[e + c0 - 1] == thr;
    // RandomDrawTreeMap.java:1733-1734: the id that triggered the parent's split is stored with a
    // directory tag, repaired only when a redistributed id shares its slot
    if (has_thr && c0 == 1) atomicAdd(&o.counters[1], 1ull);
    const int pshift = shift + g.nbits;
    const int64_t pstart = lower_of(key, 0, e, pshift, key[e] >> pshift);
    const uint32_t prank = o.cnt_dn[pstart] - 1u;
    const uint32_t slot = (uint32_t)(pre & (uint64_t)(g.dns - 1));
    if (level == 0 || n <= g.B || n == c0) {
        o.dec[e] = FROZEN;
        o.leaf_flag[e] = 1u;
        o.leaf_pa[e] = (uint32_t)pstart;
        o.leaf_pb[e] = (prank << 16) | slot;
        atomicMax(&o.counters[2], (unsigned long long)n);
        return;
    }
    const int64_t m = (c0 > g.B ? c0 : (int64_t)g.B) + 1;
    o.dec[e] = (int32_t)(uint32_t)val[e + m - 1];
    const unsigned long long r = atomicAdd(&o.counters[0], 1ull);
    if ((int64_t)r < o.rec_cap) o.recs[r] = Rec{(uint32_t)e, o.cnt_dn[e], (uint32_t)pstart, (prank << 16) | slot};
}

__global__ void k_count_new(const Rec *__restrict__ recs, int64_t begin, int64_t end, uint32_t *__restrict__ cnt_dn)
{
    const int64_t r = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= end) return;
    cnt_dn[recs[r].start] += 1u;     // one new directory per position per round
}

__global__ void k_apply(uint64_t *__restrict__ key, uint64_t *__restrict__ val, int64_t total, int shift,
                        const uint32_t *__restrict__ seg_start, const int32_t *__restrict__ dec)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const uint64_t v = val[e];
    if ((int32_t)(v >> 32) == FROZEN) return;
    const int32_t d = dec[seg_start[e]];
    if (d == FROZEN) key[e] &= ~((1ull << shift) - 1ull);
    val[e] = ((uint64_t)(uint32_t)d << 32) | (v & 0xffffffffull);
}

__global__ void k_place_dnodes(const Rec *__restrict__ recs, int64_t nrec, const uint32_t *__restrict__ dn_base, int dns,
                               uint32_t *__restrict__ roots, uint32_t *__restrict__ dnodes)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const Rec c = recs[r];
    const uint32_t idx = 1u + dn_base[c.start] + c.rank;
    if (c.pstart == ROOT) { roots[c.pinfo] = idx; return; }
    const uint32_t pidx = 1u + dn_base[c.pstart] + (c.pinfo >> 16);
    dnodes[(size_t)pidx * dns + (c.pinfo & 0xffffu)] = idx << 1;
}

__global__ void k_place_leaves(const uint64_t *__restrict__ val, int64_t total, const uint32_t *__restrict__ leaf_flag,
                               const uint32_t *__restrict__ leaf_idx, const uint32_t *__restrict__ leaf_pa,
                               const uint32_t *__restrict__ leaf_pb, const uint32_t *__restrict__ dn_base, int dns,
                               uint32_t *__restrict__ dnodes, uint32_t *__restrict__ leaf_off, int32_t *__restrict__ ids,
                               uint32_t num_leaves)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    ids[e] = (int32_t)(uint32_t)val[e];
    if (e == 0) leaf_off[num_leaves] = (uint32_t)total;
    if (!leaf_flag[e]) return;
    const uint32_t l = leaf_idx[e];
    leaf_off[l] = (uint32_t)e;
    const uint32_t pb = leaf_pb[e];
    const uint32_t pidx = 1u + dn_base[leaf_pa[e]] + (pb >> 16);
    dnodes[(size_t)pidx * dns + (pb & 0xffffu)] = (l << 1) | 1u;
}

__global__ void k_part_hist(const int32_t *__restrict__ parts, int64_t total, int L, int P, unsigned long long *__restrict__ out)
{
    extern __shared__ uint32_t s_hist[];
    const int bins = L * P;
    for (int i = threadIdx.x; i < bins; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e % L);
        atomicAdd(&s_hist[t * P + parts[e]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&out[i], (unsigned long long)s_hist[i]);
}

// ---- row shards and scan tiles ---------------------------------------------------------------

__global__ void k_keep_flags(const int32_t *__restrict__ ids, int64_t total, int64_t lo, int64_t hi, uint32_t *__restrict__ keep)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    keep[e] = (ids[e] >= lo && ids[e] < hi) ? 1u : 0u;
}

__global__ void k_keep_move(const int32_t *__restrict__ ids, int64_t total, const uint32_t *__restrict__ keep,
                            const uint32_t *__restrict__ pos, int32_t *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    if (keep[e]) out[pos[e]] = ids[e];
}

__global__ void k_keep_offsets(uint32_t *__restrict__ leaf_off, int64_t nl, int64_t total, const uint32_t *__restrict__ keep,
                               const uint32_t *__restrict__ pos)
{
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l > nl) return;
    const uint32_t o = leaf_off[l];
    leaf_off[l] = (int64_t)o < total ? pos[o] : (total > 0 ? pos[total - 1] + keep[total - 1] : 0u);
}

__global__ void k_leaf_tiles(const uint32_t *__restrict__ leaf_off, int64_t nl, int rows_per_tile, uint32_t *__restrict__ ntile)
{
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nl) return;
    ntile[l] = (leaf_off[l + 1] - leaf_off[l] + rows_per_tile - 1) / rows_per_tile;
}

__global__ void k_fill_tiles(const uint32_t *__restrict__ leaf_off, int64_t nl, int rows_per_tile, const uint32_t *__restrict__ ntile,
                             uint32_t *__restrict__ leaf_tile0, uint4 *__restrict__ tiles)
{
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nl) return;
    const uint32_t off = leaf_off[l], len = leaf_off[l + 1] - off;
    uint32_t t = leaf_tile0[l];
    if (l == nl - 1) leaf_tile0[nl] = t + ntile[l];
    for (uint32_t r = 0; r < len; r += rows_per_tile, t++)
        tiles[t] = make_uint4((uint32_t)l, off + r, min((uint32_t)rows_per_tile, len - r), 0u);
}

inline unsigned blocks_for(int64_t n, int tpb) { return (unsigned)((n + tpb - 1) / tpb); }

struct Tmp {            // frees every scratch allocation on scope exit
    std::vector<void *> ptrs;
    ~Tmp() { for (void *p : ptrs) if (p) (void)hipFree(p); }
    template <typename T> hipError_t get(T **out, size_t count)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 256));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T *)p;
        return e;
    }
    void drop(void *p)
    {
        for (auto &q : ptrs) if (q == p) { (void)hipFree(q); q = nullptr; }
    }
};

}  // namespace

void DeviceForest::release()
{
    if (roots) (void)hipFree(roots);
    if (dnodes) (void)hipFree(dnodes);
    if (leaf_off) (void)hipFree(leaf_off);
    if (ids) (void)hipFree(ids);
    roots = dnodes = leaf_off = nullptr;
    ids = nullptr;
}

int build_forest_device(const ForestGeometry &fg, const int32_t *d_hashes, const int32_t *d_parts, int64_t n, DeviceForest &out,
                        std::string &err)
{
    const int L = fg.num_tables, DNS = fg.dir_node_size;
    const int64_t total = n * (int64_t)L;
    const size_t PS = (size_t)fg.num_parts * fg.seg;
    if (total >= (int64_t)0x7fffffff) { err = "device forest build: more than 2^31 (id, table) pairs"; return -1; }
    Geo g{L, fg.nbits, DNS, fg.max_level, (fg.max_level + 1) * fg.nbits, fg.bucket_bits, fg.seg, fg.num_parts, fg.bucket_overflow};
    int grp_bits = 1;
    while (((uint64_t)1 << grp_bits) < (uint64_t)L * PS) grp_bits++;
    const int total_bits = g.lowbits + grp_bits;
    if (total_bits > 64) { err = "device forest build: key does not fit 64 bits"; return -1; }
    const int TPB = 256;
    hipStream_t stream = nullptr;
    Tmp tmp;

    out.release();
    out.num_dnodes = out.num_leaves = out.max_leaf = out.misflag_events = 0;
    out.num_ids = total;
    out.part_counts.assign((size_t)L * fg.num_parts, 0);
    BCHECK(hipMalloc((void **)&out.roots, std::max<size_t>(PS * L * 4, 16)));
    BCHECK(hipMemsetAsync(out.roots, 0, PS * L * 4, stream));

    if (total == 0) {
        BCHECK(hipMalloc((void **)&out.dnodes, (size_t)DNS * 4));
        BCHECK(hipMemsetAsync(out.dnodes, 0, (size_t)DNS * 4, stream));
        BCHECK(hipMalloc((void **)&out.leaf_off, 16));
        BCHECK(hipMemsetAsync(out.leaf_off, 0, 16, stream));
        BCHECK(hipMalloc((void **)&out.ids, 16));
        BCHECK(hipStreamSynchronize(stream));
        return 0;
    }

    unsigned long long *d_counters = nullptr, *d_hist = nullptr;
    const int bins = L * fg.num_parts;
    BCHECK(tmp.get(&d_counters, 4));
    BCHECK(tmp.get(&d_hist, (size_t)bins));
    BCHECK(hipMemsetAsync(d_counters, 0, 4 * sizeof(unsigned long long), stream));
    BCHECK(hipMemsetAsync(d_hist, 0, (size_t)bins * sizeof(unsigned long long), stream));

    uint64_t *key[2] = {nullptr, nullptr}, *val[2] = {nullptr, nullptr};
    uint32_t *hp = nullptr, *seg_start = nullptr, *cnt_dn = nullptr, *leaf_flag = nullptr, *leaf_pa = nullptr, *leaf_pb = nullptr;
    int32_t *dec = nullptr;
    Rec *recs = nullptr;
    int64_t rec_cap = 0, nrec = 0;
    unsigned long long h_counters[4] = {0, 0, 0, 0};
    for (int i = 0; i < 2; i++) { BCHECK(tmp.get(&key[i], (size_t)total)); BCHECK(tmp.get(&val[i], (size_t)total)); }
    BCHECK(tmp.get(&hp, (size_t)total));
    BCHECK(tmp.get(&seg_start, (size_t)total));
    BCHECK(tmp.get(&cnt_dn, (size_t)total));
    BCHECK(tmp.get(&leaf_flag, (size_t)total));
    BCHECK(tmp.get(&leaf_pa, (size_t)total));
    BCHECK(tmp.get(&leaf_pb, (size_t)total));
    BCHECK(tmp.get(&dec, (size_t)total));
    BCHECK(hipMemsetAsync(cnt_dn, 0, (size_t)total * 4, stream));
    BCHECK(hipMemsetAsync(leaf_flag, 0, (size_t)total * 4, stream));

    k_part_hist<<<1024, TPB, (size_t)bins * 4, stream>>>(d_parts, total, L, fg.num_parts, d_hist);
    k_build_keys<<<blocks_for(total, TPB), TPB, 0, stream>>>(d_hashes, d_parts, total, g, key[0], val[0]);

    rocprim::double_buffer<uint64_t> kb(key[0], key[1]), vb(val[0], val[1]);
    size_t sort_bytes = 0, scan_bytes = 0, scan2 = 0;
    BCHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, kb, vb, (size_t)total, 0u, (unsigned)total_bits, stream));
    BCHECK(rocprim::inclusive_scan(nullptr, scan_bytes, hp, seg_start, (size_t)total, rocprim::maximum<uint32_t>(), stream));
    BCHECK(rocprim::exclusive_scan(nullptr, scan2, cnt_dn, hp, 0u, (size_t)total, rocprim::plus<uint32_t>(), stream));
    char *d_tmp = nullptr;
    const size_t tmp_bytes = std::max(sort_bytes, std::max(scan_bytes, scan2));
    BCHECK(tmp.get(&d_tmp, tmp_bytes));

    for (int level = g.max_level + 1; level >= 0; level--) {
        const int shift = level * g.nbits;
        // records this round can add: one per root group, or one per segment of more than B ids
        const int64_t room = level == g.max_level + 1 ? std::min<int64_t>(total, (int64_t)L * (int64_t)PS)
                                                      : total / ((int64_t)g.B + 1) + 1;
        if (nrec + room > rec_cap) {
            Rec *grown = nullptr;
            const int64_t cap = nrec + room;
            BCHECK(tmp.get(&grown, (size_t)cap));
            if (nrec > 0) BCHECK(hipMemcpyAsync(grown, recs, (size_t)nrec * sizeof(Rec), hipMemcpyDeviceToDevice, stream));
            BCHECK(hipStreamSynchronize(stream));
            if (recs) tmp.drop(recs);
            recs = grown;
            rec_cap = cap;
        }
        size_t sb = tmp_bytes;
        BCHECK(rocprim::radix_sort_pairs(d_tmp, sb, kb, vb, (size_t)total, (unsigned)shift, (unsigned)total_bits, stream));
        uint64_t *K = kb.current(), *V = vb.current();
        k_head_pos<<<blocks_for(total, TPB), TPB, 0, stream>>>(K, total, shift, hp);
        sb = tmp_bytes;
        BCHECK(rocprim::inclusive_scan(d_tmp, sb, hp, seg_start, (size_t)total, rocprim::maximum<uint32_t>(), stream));
        RoundOut ro{dec, cnt_dn, leaf_flag, leaf_pa, leaf_pb, recs, d_counters, rec_cap};
        k_decide<<<blocks_for(total, TPB), TPB, 0, stream>>>(K, V, total, g, level, ro);
        BCHECK(hipMemcpyAsync(h_counters, d_counters, sizeof(h_counters), hipMemcpyDeviceToHost, stream));
        BCHECK(hipStreamSynchronize(stream));
        const int64_t now = (int64_t)h_counters[0];
        if (now > rec_cap) { err = "device forest build: directory record buffer too small"; return -2; }
        if (now > nrec) k_count_new<<<blocks_for(now - nrec, TPB), TPB, 0, stream>>>(recs, nrec, now, cnt_dn);
        k_apply<<<blocks_for(total, TPB), TPB, 0, stream>>>(K, V, total, shift, seg_start, dec);
        const bool more = now > nrec;
        nrec = now;
        if (!more) break;        // no directory was created: every id sits in a final bucket
    }
    BCHECK(hipGetLastError());

    // numbering: directories in pre-order = by (position, rank); buckets by position
    if ((uint64_t)(nrec + 1) * (uint64_t)DNS >= 0x7fffffffull) { err = "device forest build: directory arena exceeds 32-bit indexing"; return -1; }
    uint32_t *dn_base = hp, *leaf_idx = seg_start;
    size_t sb = tmp_bytes;
    BCHECK(rocprim::exclusive_scan(d_tmp, sb, cnt_dn, dn_base, 0u, (size_t)total, rocprim::plus<uint32_t>(), stream));
    sb = tmp_bytes;
    BCHECK(rocprim::exclusive_scan(d_tmp, sb, leaf_flag, leaf_idx, 0u, (size_t)total, rocprim::plus<uint32_t>(), stream));
    uint32_t last_idx = 0, last_flag = 0;
    BCHECK(hipMemcpyAsync(&last_idx, leaf_idx + (total - 1), 4, hipMemcpyDeviceToHost, stream));
    BCHECK(hipMemcpyAsync(&last_flag, leaf_flag + (total - 1), 4, hipMemcpyDeviceToHost, stream));
    BCHECK(hipMemcpyAsync(h_counters, d_counters, sizeof(h_counters), hipMemcpyDeviceToHost, stream));
    std::vector<unsigned long long> h_hist((size_t)bins);
    BCHECK(hipMemcpyAsync(h_hist.data(), d_hist, (size_t)bins * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    BCHECK(hipStreamSynchronize(stream));
    const uint32_t num_leaves = last_idx + last_flag;
    out.num_dnodes = nrec;
    out.num_leaves = num_leaves;
    out.misflag_events = (int64_t)h_counters[1];
    out.max_leaf = (int64_t)h_counters[2];
    for (int i = 0; i < bins; i++) out.part_counts[(size_t)i] = (int64_t)h_hist[(size_t)i];
    BCHECK(hipMalloc((void **)&out.dnodes, (size_t)(nrec + 1) * DNS * 4));
    BCHECK(hipMemsetAsync(out.dnodes, 0, (size_t)(nrec + 1) * DNS * 4, stream));
    BCHECK(hipMalloc((void **)&out.ids, (size_t)total * 4));
    BCHECK(hipMalloc((void **)&out.leaf_off, ((size_t)num_leaves + 1) * 4));
    k_place_dnodes<<<blocks_for(nrec, TPB), TPB, 0, stream>>>(recs, nrec, dn_base, DNS, out.roots, out.dnodes);
    k_place_leaves<<<blocks_for(total, TPB), TPB, 0, stream>>>(vb.current(), total, leaf_flag, leaf_idx, leaf_pa, leaf_pb, dn_base, DNS,
                                                              out.dnodes, out.leaf_off, out.ids, num_leaves);
    BCHECK(hipGetLastError());
    BCHECK(hipStreamSynchronize(stream));
    return 0;
}

// Keeps only the ids of [lo, hi) in every bucket (a row-sharded handle replicates the trie, not the bucket
// contents).  ids and leaf_off are rewritten in place; returns the number of ids kept.
int filter_forest_rows(DeviceForest &f, int64_t lo, int64_t hi, std::string &err)
{
    const int64_t total = f.num_ids, nl = f.num_leaves;
    if (total == 0) return 0;
    const int TPB = 256;
    Tmp tmp;
    uint32_t *keep = nullptr, *pos = nullptr;
    int32_t *kept = nullptr;
    char *d_tmp = nullptr;
    BCHECK(tmp.get(&keep, (size_t)total));
    BCHECK(tmp.get(&pos, (size_t)total));
    size_t bytes = 0;
    BCHECK(rocprim::exclusive_scan(nullptr, bytes, keep, pos, 0u, (size_t)total, rocprim::plus<uint32_t>(), nullptr));
    BCHECK(tmp.get(&d_tmp, bytes));
    k_keep_flags<<<blocks_for(total, TPB), TPB>>>(f.ids, total, lo, hi, keep);
    BCHECK(rocprim::exclusive_scan(d_tmp, bytes, keep, pos, 0u, (size_t)total, rocprim::plus<uint32_t>(), nullptr));
    uint32_t last_pos = 0, last_keep = 0;
    BCHECK(hipMemcpy(&last_pos, pos + (total - 1), 4, hipMemcpyDeviceToHost));
    BCHECK(hipMemcpy(&last_keep, keep + (total - 1), 4, hipMemcpyDeviceToHost));
    const int64_t nkept = (int64_t)last_pos + last_keep;
    BCHECK(hipMalloc((void **)&kept, std::max<size_t>((size_t)nkept * 4, 16)));
    k_keep_move<<<blocks_for(total, TPB), TPB>>>(f.ids, total, keep, pos, kept);
    k_keep_offsets<<<blocks_for(nl + 1, TPB), TPB>>>(f.leaf_off, nl, total, keep, pos);
    BCHECK(hipGetLastError());
    BCHECK(hipDeviceSynchronize());
    (void)hipFree(f.ids);
    f.ids = kept;
    f.num_ids = nkept;
    return 0;
}

// Cuts every bucket into tiles of rows_per_tile ids: tiles[] = {bucket, first id slot, rows, 0},
// leaf_tile0[bucket] = its first tile (num_leaves + 1 entries).  Both arrays are hipMalloc'ed here.
int make_forest_tiles(const DeviceForest &f, int rows_per_tile, uint32_t **leaf_tile0, uint4 **tiles, int64_t *ntiles, std::string &err)
{
    const int64_t nl = f.num_leaves;
    const int TPB = 256;
    *leaf_tile0 = nullptr;
    *tiles = nullptr;
    *ntiles = 0;
    BCHECK(hipMalloc((void **)leaf_tile0, ((size_t)nl + 1) * 4 + 16));
    if (nl == 0) {
        BCHECK(hipMemset(*leaf_tile0, 0, 16));
        BCHECK(hipMalloc((void **)tiles, 16));
        return 0;
    }
    Tmp tmp;
    uint32_t *ntile = nullptr;
    char *d_tmp = nullptr;
    BCHECK(tmp.get(&ntile, (size_t)nl));
    size_t bytes = 0;
    BCHECK(rocprim::exclusive_scan(nullptr, bytes, ntile, *leaf_tile0, 0u, (size_t)nl, rocprim::plus<uint32_t>(), nullptr));
    BCHECK(tmp.get(&d_tmp, bytes));
    k_leaf_tiles<<<blocks_for(nl, TPB), TPB>>>(f.leaf_off, nl, rows_per_tile, ntile);
    BCHECK(rocprim::exclusive_scan(d_tmp, bytes, ntile, *leaf_tile0, 0u, (size_t)nl, rocprim::plus<uint32_t>(), nullptr));
    uint32_t a = 0, b = 0;
    BCHECK(hipMemcpy(&a, *leaf_tile0 + (nl - 1), 4, hipMemcpyDeviceToHost));
    BCHECK(hipMemcpy(&b, ntile + (nl - 1), 4, hipMemcpyDeviceToHost));
    *ntiles = (int64_t)a + b;
    BCHECK(hipMalloc((void **)tiles, std::max<size_t>((size_t)*ntiles * 16, 16)));
    k_fill_tiles<<<blocks_for(nl, TPB), TPB>>>(f.leaf_off, nl, rows_per_tile, ntile, *leaf_tile0, *tiles);
    BCHECK(hipGetLastError());
    BCHECK(hipDeviceSynchronize());
    return 0;
}

}  // namespace dpf
