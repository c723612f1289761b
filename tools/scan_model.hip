// Model of the bucket-major scan's memory / latency structure, without its bookkeeping: every unit
// = 48 random rows (3 blocks of 16) of a 1M x 256 B fp16 table held as MFMA B operands + 2 chunks
// of 16 random queries out of 10k (256 B rows + a 16 B record + a 4 B threshold each) as A operands,
// 12 MFMAs per chunk, one max/compare over the accumulators.  Variants differ in how the queries
// reach the matrix cores.
//   0: as the production kernel: query loads at the head of each chunk
//   1: rows only (no chunk work)
//   2: chunks only (no row loads)
//   3: both chunks' query rows requested together with the unit's rows (32 more VGPRs)
//   4: query rows of both chunks by LDS-DMA at the head of the unit, A operands read from LDS
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/scan_model tools/scan_model.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    const uint16_t *table;     // [nrows][128]
    const uint16_t *Q;         // [nq][128]
    const float4 *qc;          // [nq]
    const float *tau;          // [nq]
    const int32_t *ids;        // [nunits][48]
    const int32_t *qidx;       // [nunits][32]
    uint32_t nunits;
    uint32_t *sink;
};

template <int V, int WAVES, bool NT = false>
__global__ __launch_bounds__(256, WAVES) void k_model(Args a)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_q[V == 4 ? 4 : 1][V == 4 ? 32 * 128 : 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const uint32_t W = gridDim.x * 4u;
    uint32_t hits = 0;
    for (uint32_t u = blockIdx.x * 4u + wave; u < a.nunits; u += W) {
        f16x8 xb[3][4];
        int32_t q0 = a.qidx[(size_t)u * 32 + i16], q1 = a.qidx[(size_t)u * 32 + 16 + i16];
        if (V != 2) {
#pragma unroll
            for (int rb = 0; rb < 3; rb++) {
                const size_t row = (size_t)a.ids[(size_t)u * 48 + rb * 16 + i16];
                const uint16_t *p = a.table + row * 128 + 8 * g;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    xb[rb][t] = NT ? __builtin_nontemporal_load((const f16x8 *)(p + 32 * t)) : *(const f16x8 *)(p + 32 * t);
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < 3; rb++)
#pragma unroll
                for (int t = 0; t < 4; t++) xb[rb][t] = (f16x8){(_Float16)(float)u, 0, 0, 0, 0, 0, 0, 0};
        }
        if (V == 1) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 s = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int rb = 0; rb < 3; rb++)
#pragma unroll
                for (int t = 0; t < 4; t++) s ^= __builtin_bit_cast(u32x4, xb[rb][t]);
            if ((s.x ^ s.y ^ s.z ^ s.w) == 0x12345u) hits++;
            continue;
        }
        f16x8 qa[2][4];
        if (V == 3) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint16_t *p = a.Q + (size_t)(c ? q1 : q0) * 128 + 8 * g;
#pragma unroll
                for (int t = 0; t < 4; t++) qa[c][t] = *(const f16x8 *)(p + 32 * t);
            }
        }
        if (V == 4) {
            // 32 query rows x 256 B = 8 KiB: 8 DMA instructions of 64 lanes x 16 B; instruction j brings
            // rows 4j .. 4j+3 (lane -> row 4j + lane/16, piece lane%16); LDS image: row r at r*256 B
            const int32_t mine = lane < 16 ? q0 : q1;      // lanes 0-15: chunk 0's queries, 16-31: chunk 1's (by i16)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 4 * j + (lane >> 4);          // query slot 0..31
                const int32_t qi = __shfl(mine, r);
                const uint16_t *src = a.Q + (size_t)qi * 128 + 8 * (lane & 15);
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(s_q[wave] + j * 512), 16, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int32_t qi = c ? q1 : q0;
            if (V == 0 || V == 2) {
                const uint16_t *p = a.Q + (size_t)qi * 128 + 8 * g;
#pragma unroll
                for (int t = 0; t < 4; t++) qa[c][t] = *(const f16x8 *)(p + 32 * t);
            }
            if (V == 4) {
                // A operand of slot i16: row (16c + i16), pieces 4t + g
#pragma unroll
                for (int t = 0; t < 4; t++) qa[c][t] = *(const f16x8 *)(s_q[wave] + (16 * c + i16) * 128 + (4 * t + g) * 8);
            }
            const float tau = a.tau[qi];
            const float4 qc = a.qc[qi];
            const float G = qc.x - 0.5f * (tau + qc.y) * (tau + qc.y);
            f32x4 acc[3];
#pragma unroll
            for (int rb = 0; rb < 3; rb++) {
                acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; t++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[c][t], xb[rb][t], acc[rb], 0, 0, 0);
            }
            float m = -1e30f;
#pragma unroll
            for (int rb = 0; rb < 3; rb++) m = fmaxf(m, fmaxf(fmaxf(acc[rb][0], acc[rb][1]), fmaxf(acc[rb][2], acc[rb][3])));
            if (__ballot(m >= G + 1e30f)) hits++;
        }
    }
    if (hits == 0x7fffffffu) a.sink[0] = 1;
}


// variant 5: the next unit's rows are requested before the current unit's chunks run (two tile
// register sets, 2 waves per SIMD)
__device__ __forceinline__ void load_tile(const Args &a, uint32_t u, int i16, int g, f16x8 (&xb)[3][4])
{
#pragma unroll
    for (int rb = 0; rb < 3; rb++) {
        const size_t row = (size_t)a.ids[(size_t)u * 48 + rb * 16 + i16];
        const uint16_t *p = a.table + row * 128 + 8 * g;
#pragma unroll
        for (int t = 0; t < 4; t++) xb[rb][t] = *(const f16x8 *)(p + 32 * t);
    }
}
__device__ __forceinline__ uint32_t run_chunks(const Args &a, uint32_t u, int i16, int g, const f16x8 (&xb)[3][4])
{
    uint32_t hits = 0;
    const int32_t q0 = a.qidx[(size_t)u * 32 + i16], q1 = a.qidx[(size_t)u * 32 + 16 + i16];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int32_t qi = c ? q1 : q0;
        f16x8 qa[4];
        const uint16_t *p = a.Q + (size_t)qi * 128 + 8 * g;
#pragma unroll
        for (int t = 0; t < 4; t++) qa[t] = *(const f16x8 *)(p + 32 * t);
        const float tau = a.tau[qi];
        const float4 qc = a.qc[qi];
        const float G = qc.x - 0.5f * (tau + qc.y) * (tau + qc.y);
        f32x4 acc[3];
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
            acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; t++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[t], xb[rb][t], acc[rb], 0, 0, 0);
        }
        float m = -1e30f;
#pragma unroll
        for (int rb = 0; rb < 3; rb++) m = fmaxf(m, fmaxf(fmaxf(acc[rb][0], acc[rb][1]), fmaxf(acc[rb][2], acc[rb][3])));
        if (__ballot(m >= G + 1e30f)) hits++;
    }
    return hits;
}
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_model5(Args a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const uint32_t W = gridDim.x * 4u;
    uint32_t hits = 0;
    uint32_t u = blockIdx.x * 4u + wave;
    if (u >= a.nunits) return;
    f16x8 xa[3][4], xb[3][4];
    load_tile(a, u, i16, g, xa);
    for (; u < a.nunits; u += 2 * W) {
        if (u + W < a.nunits) load_tile(a, u + W, i16, g, xb);
        hits += run_chunks(a, u, i16, g, xa);
        if (u + W >= a.nunits) break;
        if (u + 2 * W < a.nunits) load_tile(a, u + 2 * W, i16, g, xa);
        hits += run_chunks(a, u + W, i16, g, xb);
    }
    if (hits == 0x7fffffffu) a.sink[0] = 1;
}

// variants 9/10: rows fetched as whole rows (one instruction = 4 rows x 256 B; 10: non-temporal) and
// turned into MFMA B operands through a swizzled LDS image
template <bool NT>
__global__ __launch_bounds__(256, 3) void k_model9(Args a)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ u32x4 s_t[4][48 * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const uint32_t W = gridDim.x * 4u;
    uint32_t hits = 0;
    for (uint32_t u = blockIdx.x * 4u + wave; u < a.nunits; u += W) {
        const int32_t mine = lane < 48 ? a.ids[(size_t)u * 48 + lane] : 0;
        u32x4 raw[12];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const size_t row = (size_t)__shfl(mine, 4 * j + (lane >> 4));
            const u32x4 *p = (const u32x4 *)(a.table + row * 128) + (lane & 15);
            raw[j] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const int row = 4 * j + (lane >> 4);
            s_t[wave][row * 16 + ((lane & 15) ^ (row & 15))] = raw[j];
        }
        __builtin_amdgcn_wave_barrier();
        f16x8 xb[3][4];
#pragma unroll
        for (int rb = 0; rb < 3; rb++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int row = rb * 16 + i16;
                xb[rb][t] = __builtin_bit_cast(f16x8, s_t[wave][row * 16 + ((4 * t + g) ^ (row & 15))]);
            }
        __builtin_amdgcn_wave_barrier();
        hits += run_chunks(a, u, i16, g, xb);
    }
    if (hits == 0x7fffffffu) a.sink[0] = 1;
}

// variant 11: both chunks' query rows are requested BEFORE the next unit's rows, so that the in-order
// load counter lets the wave wait for its queries while the next tile is still on its way
struct Qs { f16x8 qa[2][4]; float tau[2]; float4 qc[2]; };
__device__ __forceinline__ void load_queries(const Args &a, uint32_t u, int i16, int g, Qs &q)
{
    const int32_t q0 = a.qidx[(size_t)u * 32 + i16], q1 = a.qidx[(size_t)u * 32 + 16 + i16];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int32_t qi = c ? q1 : q0;
        const uint16_t *p = a.Q + (size_t)qi * 128 + 8 * g;
#pragma unroll
        for (int t = 0; t < 4; t++) q.qa[c][t] = *(const f16x8 *)(p + 32 * t);
        q.tau[c] = a.tau[qi];
        q.qc[c] = a.qc[qi];
    }
}
__device__ __forceinline__ uint32_t chunks_of(const Qs &q, const f16x8 (&xb)[3][4])
{
    uint32_t hits = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const float G = q.qc[c].x - 0.5f * (q.tau[c] + q.qc[c].y) * (q.tau[c] + q.qc[c].y);
        f32x4 acc[3];
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
            acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; t++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q.qa[c][t], xb[rb][t], acc[rb], 0, 0, 0);
        }
        float m = -1e30f;
#pragma unroll
        for (int rb = 0; rb < 3; rb++) m = fmaxf(m, fmaxf(fmaxf(acc[rb][0], acc[rb][1]), fmaxf(acc[rb][2], acc[rb][3])));
        if (__ballot(m >= G + 1e30f)) hits++;
    }
    return hits;
}
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_model11(Args a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const uint32_t W = gridDim.x * 4u;
    uint32_t hits = 0;
    uint32_t u = blockIdx.x * 4u + wave;
    if (u >= a.nunits) return;
    f16x8 xa[3][4], xb[3][4];
    Qs q;
    load_tile(a, u, i16, g, xa);
    for (; u < a.nunits; u += 2 * W) {
        load_queries(a, u, i16, g, q);
        if (u + W < a.nunits) load_tile(a, u + W, i16, g, xb);
        hits += chunks_of(q, xa);
        if (u + W >= a.nunits) break;
        load_queries(a, u + W, i16, g, q);
        if (u + 2 * W < a.nunits) load_tile(a, u + 2 * W, i16, g, xa);
        hits += chunks_of(q, xb);
    }
    if (hits == 0x7fffffffu) a.sink[0] = 1;
}

int main(int argc, char **argv)
{
    const uint32_t nrows = 1u << 20, nq = 10000, nunits = argc > 1 ? (uint32_t)atol(argv[1]) : 800000u;
    std::vector<int32_t> ids((size_t)nunits * 48), qidx((size_t)nunits * 32);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); };
    for (auto &v : ids) v = (int32_t)(rnd() & (nrows - 1));
    // queries of a unit come from a small neighbourhood, as buckets of one region share their queries
    for (uint32_t u = 0; u < nunits; u++) {
        const uint32_t base = rnd() % nq;
        for (int j = 0; j < 32; j++) qidx[(size_t)u * 32 + j] = (int32_t)((base + rnd() % 512) % nq);
    }
    Args a;
    uint16_t *d_table, *d_Q;
    float4 *d_qc;
    float *d_tau;
    int32_t *d_ids, *d_qidx;
    CK(hipMalloc((void **)&d_table, (size_t)nrows * 256));
    CK(hipMemset(d_table, 0, (size_t)nrows * 256));
    CK(hipMalloc((void **)&d_Q, (size_t)nq * 256));
    CK(hipMemset(d_Q, 0, (size_t)nq * 256));
    CK(hipMalloc((void **)&d_qc, (size_t)nq * 16));
    CK(hipMemset(d_qc, 0, (size_t)nq * 16));
    CK(hipMalloc((void **)&d_tau, (size_t)nq * 4));
    CK(hipMemset(d_tau, 0, (size_t)nq * 4));
    CK(hipMalloc((void **)&d_ids, ids.size() * 4));
    CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_qidx, qidx.size() * 4));
    CK(hipMemcpy(d_qidx, qidx.data(), qidx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&a.sink, 4));
    a.table = d_table; a.Q = d_Q; a.qc = d_qc; a.tau = d_tau; a.ids = d_ids; a.qidx = d_qidx; a.nunits = nunits;
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int v = 0; v < 13; v++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            const int mult = argc > 2 ? atoi(argv[2]) : 3;
            const dim3 grid(cus * ((v == 5 || v == 11) ? 2 : mult)), block(256);
            CK(hipEventRecord(e0));
            switch (v) {
            case 0: hipLaunchKernelGGL((k_model<0, 1>), grid, block, 0, 0, a); break;
            case 1: hipLaunchKernelGGL((k_model<1, 1>), grid, block, 0, 0, a); break;
            case 2: hipLaunchKernelGGL((k_model<2, 1>), grid, block, 0, 0, a); break;
            case 3: hipLaunchKernelGGL((k_model<3, 3>), grid, block, 0, 0, a); break;
            case 4: hipLaunchKernelGGL((k_model<4, 3>), grid, block, 0, 0, a); break;
            case 5: hipLaunchKernelGGL((k_model5<2>), grid, block, 0, 0, a); break;
            case 6: hipLaunchKernelGGL((k_model5<3>), grid, block, 0, 0, a); break;
            case 7: hipLaunchKernelGGL((k_model<0, 1, true>), grid, block, 0, 0, a); break;
            case 8: hipLaunchKernelGGL((k_model<1, 1, true>), grid, block, 0, 0, a); break;
            case 9: hipLaunchKernelGGL((k_model9<false>), grid, block, 0, 0, a); break;
            case 10: hipLaunchKernelGGL((k_model9<true>), grid, block, 0, 0, a); break;
            case 11: hipLaunchKernelGGL((k_model11<2>), grid, block, 0, 0, a); break;
            default: hipLaunchKernelGGL((k_model11<3>), grid, block, 0, 0, a); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("variant %d: %u units: %.3f ms\n", v, nunits, best);
    }
    return 0;
}
