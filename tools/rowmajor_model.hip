// Model of a ROW-major scan, to size it against the bucket-major one: every 64-row tile of a
// 1M x 256 B fp16 table is read ONCE (contiguous rows) and scored on the matrix cores against a list of S
// queries out of 10k (the union of the queries that probe any bucket of any of its rows), 16 at a time:
// query rows (256 B, from L2), a 16 B record and a 4 B threshold per query, 16 MFMAs and one max/compare
// per chunk -- the common path of the production kernel, none of its forwarding.
//   variant 0: one chunk at a time        variant 1: two chunks in flight
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/rowmajor_model tools/rowmajor_model.hip
// Run:   rowmajor_model [S = 1200] [workgroups per CU = 3]
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
    const int32_t *qlist;      // [ntiles][S]
    uint32_t ntiles, S;
    uint32_t *sink;
};

__device__ __forceinline__ bool chunk(const Args &a, int32_t qi, int g, const f16x8 (&xb)[4][4])
{
    f16x8 qa[4];
    const uint16_t *p = a.Q + (size_t)qi * 128 + 8 * g;
#pragma unroll
    for (int t = 0; t < 4; t++) qa[t] = *(const f16x8 *)(p + 32 * t);
    const float tau = a.tau[qi];
    const float4 qc = a.qc[qi];
    const float G = qc.x - 0.5f * (tau + qc.y) * (tau + qc.y);
    f32x4 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
        acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; t++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[t], xb[rb][t], acc[rb], 0, 0, 0);
    }
    float m = -1e30f;
#pragma unroll
    for (int rb = 0; rb < 4; rb++) m = fmaxf(m, fmaxf(fmaxf(acc[rb][0], acc[rb][1]), fmaxf(acc[rb][2], acc[rb][3])));
    return __ballot(m >= G + 1e30f) != 0;
}

template <int V, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_rowmajor(Args a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const uint32_t W = gridDim.x * 4u;
    uint32_t hits = 0;
    for (uint32_t t = blockIdx.x * 4u + wave; t < a.ntiles; t += W) {
        f16x8 xb[4][4];
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            const uint16_t *p = a.table + ((size_t)t * 64 + rb * 16 + i16) * 128 + 8 * g;
#pragma unroll
            for (int k = 0; k < 4; k++) xb[rb][k] = *(const f16x8 *)(p + 32 * k);
        }
        const int32_t *ql = a.qlist + (size_t)t * a.S;
        if (V == 0) {
            int32_t qn = ql[i16];
            for (uint32_t c = 0; c < a.S; c += 16) {
                const int32_t qi = qn;
                if (c + 16 < a.S) qn = ql[c + 16 + i16];
                hits += chunk(a, qi, g, xb) ? 1u : 0u;
            }
        } else {
            int32_t qn0 = ql[i16], qn1 = ql[16 + i16];
            for (uint32_t c = 0; c < a.S; c += 32) {
                const int32_t q0 = qn0, q1 = qn1;
                if (c + 32 < a.S) { qn0 = ql[c + 32 + i16]; qn1 = ql[c + 48 + i16]; }
                const bool h0 = chunk(a, q0, g, xb), h1 = chunk(a, q1, g, xb);
                hits += (h0 ? 1u : 0u) + (h1 ? 1u : 0u);
            }
        }
    }
    if (hits == 0x7fffffffu) a.sink[0] = 1;
}

int main(int argc, char **argv)
{
    const uint32_t nrows = 1u << 20, nq = 10000, ntiles = nrows / 64;
    const uint32_t S = (argc > 1 ? (uint32_t)atol(argv[1]) : 1200u) / 32 * 32;
    const int mult = argc > 2 ? atoi(argv[2]) : 3;
    std::vector<int32_t> qlist((size_t)ntiles * S);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); };
    // a tile's queries: an ascending subset of a window of 4S query numbers (similar rows share their queries)
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint32_t base = rnd() % nq;
        uint32_t at = 0;
        for (uint32_t j = 0; j < S; j++) { at += 1 + rnd() % 7; qlist[(size_t)t * S + j] = (int32_t)((base + at) % nq); }
    }
    Args a;
    uint16_t *d_table, *d_Q;
    float4 *d_qc;
    float *d_tau;
    int32_t *d_ql;
    CK(hipMalloc((void **)&d_table, (size_t)nrows * 256));
    CK(hipMemset(d_table, 0, (size_t)nrows * 256));
    CK(hipMalloc((void **)&d_Q, (size_t)nq * 256));
    CK(hipMemset(d_Q, 0, (size_t)nq * 256));
    CK(hipMalloc((void **)&d_qc, (size_t)nq * 16));
    CK(hipMemset(d_qc, 0, (size_t)nq * 16));
    CK(hipMalloc((void **)&d_tau, (size_t)nq * 4));
    CK(hipMemset(d_tau, 0, (size_t)nq * 4));
    CK(hipMalloc((void **)&d_ql, qlist.size() * 4));
    CK(hipMemcpy(d_ql, qlist.data(), qlist.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&a.sink, 4));
    a.table = d_table; a.Q = d_Q; a.qc = d_qc; a.tau = d_tau; a.qlist = d_ql; a.ntiles = ntiles; a.S = S;
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int v = 0; v < 2; v++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            const dim3 grid(cus * mult), block(256);
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL((k_rowmajor<0, 1>), grid, block, 0, 0, a);
            else hipLaunchKernelGGL((k_rowmajor<1, 1>), grid, block, 0, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double pairs = (double)ntiles * 64.0 * S;
        printf("variant %d: %u tiles x %u queries (%.3g pairs, %.2f GB of query rows from L2): %.3f ms, %.1f TFLOP/s\n", v, ntiles, S,
               pairs, (double)ntiles * S * 256 / 1e9, best, pairs * 256.0 / best / 1e9);
    }
    return 0;
}
