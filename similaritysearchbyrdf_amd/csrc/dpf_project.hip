// Matrix-core LSH projection for gfx950: the dot products of a batch of vectors with the whole function
// pool as one fp64 contraction X[n x dim] . fnsT[dim x F] on v_mfma_f64_16x16x4_f64, with LDS-staged
// tiles of both operands, followed by the exact kernel on the few results the contraction cannot decide.
//
// Replaces, for large batches, the per-function loops of
//   src/main/scala/mclab/lsh/hashFamilies/AngleHashFamily.scala:187-195   (sign of each dot product)
//   src/main/scala/mclab/lsh/hashFamilies/PStableHashFamily.scala:122-143 (((dot + b) / w).toInt)
//   src/main/scala/mclab/lsh/vector/SimilarityCalculator.scala:40-49      (the dot product itself)
//
// Why the hashes are still bit-identical to the JVM's.  The reference sums a_i * x_i left to right in
// fp64 with separate multiply and add; the matrix cores fuse and reorder.  For ANY order of summation and
// any mix of fused and unfused operations, the computed sum differs from the exact real sum S by at most
// gamma * sum|a_i x_i| with gamma = (dim + 1) u / (1 - (dim + 1) u), u = 2^-53 (one rounding per product at
// most, at most dim additions on every term).  So the reference's sum and the contraction's differ by at most
// 2 gamma sum|a_i x_i| <= 2 gamma |a| |x| (Cauchy-Schwarz).  The kernel uses four times that as its guard
// band (GUARD below):
//   angle:   the sign bit is taken from the contraction when |s| > band; then the reference's sum has the
//            same sign.  Otherwise (this includes s == 0, where sign(0) = 0 matters, and NaN) the row is
//            undecided.
//   pStable: y = (s + b) / w; the truncated integer is taken when y is further from every integer than the
//            propagated band (two more roundings) and far from the int32 saturation limits; otherwise the
//            row is undecided.
// Undecided rows (a handful per billion results on real data; every all-zero row) are listed and hashed
// again, whole, by the exact left-to-right kernel (k_hash); every other row is assembled from the stored
// sign bits / bucket indices by k_assemble.  tests/test_gpu_parity.py compares both paths with the oracle,
// including inputs built to sit on the decision boundaries.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "dpf_device.hpp"
#include "dpf_hash_common.hpp"

namespace dpf {

namespace {

using f64x4 = __attribute__((ext_vector_type(4))) double;

constexpr int BM = 128;          // vectors per workgroup tile
constexpr int BN = 64;           // pool functions per workgroup tile
constexpr int KC = 32;           // dimensions per LDS stage
constexpr int XS = KC + 2;       // LDS strides in doubles.  Vector tile [row][k], stride 34: the 16 rows x 2 dimensions one half-wave
                                 // reads as an MFMA operand fall on 32 distinct bank pairs (4 row + 2 k), and a
                                 // half-wave's 32 consecutive dimensions of one row are stored without conflict.
constexpr int BS = BN + 16;      // Function tile [k][function], stride 80: the two dimensions of a half-wave's operand
                                 // load sit 32 bank pairs apart.

__device__ __forceinline__ double guard_factor(int dim) { return 8.0 * (double)(dim + 2) * 1.1102230246251565e-16 * 1.001; }

// |x| of every row, fp64 (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void k_row_norm(const T *__restrict__ X, int64_t x_stride, int64_t n, int dim, double *__restrict__ out,
                                                  uint32_t *__restrict__ row_flag)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const T *x = X + row * x_stride;
    double s = 0.0;
    for (int i = lane; i < dim; i += 64) { const double v = (double)x[i]; s = fma(v, v, s); }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) { out[row] = sqrt(s); row_flag[row] = 0u; }
}

// The contraction.  One-dimensional grid: workgroup = vector tile * function tiles + function tile (the function
// tile runs fastest, so that neighbouring workgroups share their vector rows in L2; a second grid dimension would
// cap a call at 65535 vector tiles).  Wave w of the workgroup owns vector rows [32 w, 32 w + 32) of the tile and all 64
// functions: 2 x 4 accumulator tiles of 16 x 16.
// v_mfma_f64_16x16x4_f64 operand maps: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], one fp64
// per lane; C/D[i = (lane >> 4) + 4 r][j = lane & 15] in register r of the lane.
template <typename T, int FAMILY>
__global__ __launch_bounds__(256, 2) void k_project_mfma(const T *__restrict__ X, int64_t x_stride, int64_t n, HashParams p,
                                                          const double *__restrict__ row_norm, uint16_t *__restrict__ bits16,
                                                          int32_t *__restrict__ codes, int ft16, uint32_t *__restrict__ row_flag,
                                                          int32_t *__restrict__ flag_rows, uint32_t *__restrict__ flag_count,
                                                          unsigned long long *__restrict__ flag_total)
{
    __shared__ double Xs[BM][XS];
    __shared__ double Bs[KC][BS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dim = p.dim, F = p.F;
    const unsigned ftiles = (unsigned)((F + BN - 1) / BN);
    const int f0 = (int)(blockIdx.x % ftiles) * BN;
    const int64_t r0 = (int64_t)(blockIdx.x / ftiles) * BM;

    f64x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    // Staging registers: 16 vector elements and 8 function elements per thread and stage.  The loads are
    // unconditional (addresses clamped into the arrays) so that all 24 are in flight together while the
    // previous stage is multiplied; what lies outside the matrices is zeroed when the stage is stored to LDS.
    T xr[16];
    double br[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int idx = tid + 256 * u;                  // consecutive threads -> consecutive dimensions of one row
            const int r = idx / KC, k = idx % KC;
            const int64_t row = r0 + r < n ? r0 + r : n - 1;
            const int kk = k0 + k < dim ? k0 + k : dim - 1;
            xr[u] = X[row * x_stride + kk];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = tid + 256 * u;                  // consecutive threads -> consecutive functions of one dimension
            const int k = idx / BN, f = idx % BN;
            const int kk = k0 + k < dim ? k0 + k : dim - 1;
            const int ff = f0 + f < F ? f0 + f : F - 1;
            br[u] = p.fnsT[(size_t)kk * F + ff];
        }
    };
    auto stash = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int idx = tid + 256 * u;
            Xs[idx / KC][idx % KC] = k0 + idx % KC < dim ? (double)xr[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = tid + 256 * u;
            Bs[idx / BN][idx % BN] = k0 + idx / BN < dim ? br[u] : 0.0;
        }
    };

    const int i16 = lane & 15, kq = lane >> 4;
    fetch(0);
    for (int k0 = 0; k0 < dim; k0 += KC) {
        __syncthreads();                                    // the previous stage has been consumed
        stash(k0);
        __syncthreads();
        if (k0 + KC < dim) fetch(k0 + KC);                  // in flight while this stage is multiplied
#pragma unroll
        for (int ks = 0; ks < KC / 4; ks++) {
            const int k = 4 * ks + kq;
            const double a0 = Xs[32 * wave + i16][k], a1 = Xs[32 * wave + 16 + i16][k];
            double b[4];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = Bs[k][16 * j + i16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[j], acc[1][j], 0, 0, 0);
            }
        }
    }

    // epilogue: decide, store, list the undecided rows
    const double gf = guard_factor(dim);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int f = f0 + 16 * j + i16;
        const bool f_ok = f < F;
        const double fnorm = f_ok ? p.fn_norm[f] : 0.0;
        double bf = 0.0, wf = 1.0;
        if (FAMILY == 1 && f_ok) { bf = p.b[f]; wf = (double)p.w[f]; }
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t row = r0 + 32 * wave + 16 * i + kq + 4 * r;
                const bool ok = f_ok && row < n;
                const double s = acc[i][j][r];
                const double band = ok ? gf * row_norm[row] * fnorm : 0.0;
                bool decided;
                if (FAMILY == 0) {
                    decided = fabs(s) > band;
                    const unsigned long long m = __ballot(ok && s > 0.0);
                    if (i16 == 0 && row < n && f0 + 16 * j < F)
                        bits16[row * ft16 + (f0 >> 4) + j] = (uint16_t)((m >> (16 * kq)) & 0xffffull);
                } else {
                    const double t = s + bf;
                    const double y = t / wf;
                    const double u = 1.1102230246251565e-16;
                    const double slack = (band + 4.0 * u * (fabs(s) + fabs(bf) + band)) / fabs(wf) + 4.0 * u * fabs(y);
                    decided = fabs(y - rint(y)) > slack && fabs(y) + slack < 2147483000.0;      // false for NaN / inf
                    if (ok) codes[row * F + f] = d2i_sat(y);
                }
                if (ok && !decided) {
                    if (atomicExch(&row_flag[row], 1u) == 0u) {
                        const uint32_t at = atomicAdd(flag_count, 1u);
                        flag_rows[at] = (int32_t)row;
                        atomicAdd(flag_total, 1ull);
                    }
                }
            }
        }
    }
}

// L hashes + sub-index ids per decided row, from the stored per-function results.  RB rows per workgroup; the
// angle family's sign bits of those rows are staged in LDS.
constexpr int ASM_ROWS = 32;
__global__ __launch_bounds__(256) void k_assemble(HashParams p, int64_t n, const uint16_t *__restrict__ bits16,
                                                  const int32_t *__restrict__ codes, int ft16, const uint32_t *__restrict__ row_flag,
                                                  int32_t *__restrict__ hashes, int32_t *__restrict__ parts)
{
    extern __shared__ uint16_t s_bits[];                    // [ASM_ROWS][ft16] (angle)
    const int64_t r0 = (int64_t)blockIdx.x * ASM_ROWS;
    const int rows = (int)(n - r0 < ASM_ROWS ? n - r0 : ASM_ROWS);
    if (p.family == 0) {
        for (int idx = threadIdx.x; idx < rows * ft16; idx += 256) s_bits[idx] = bits16[r0 * ft16 + idx];
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < rows * p.L; idx += 256) {
        const int q = idx / p.L, t = idx - q * p.L;
        const int64_t row = r0 + q;
        if (row_flag[row]) continue;                        // the exact kernel writes this row
        const uint16_t *rb = s_bits + (size_t)q * ft16;
        const int32_t *rc = codes + row * p.F;
        int32_t h, part;
        assemble_hash(p, t,
                      [&](int f) { return (uint32_t)((rb[f >> 4] >> (f & 15)) & 1u); },
                      [&](int f) { return rc[f]; },
                      h, part);
        hashes[row * p.L + t] = h;
        if (parts) parts[row * p.L + t] = part;
    }
}

}  // namespace

size_t project_codes_bytes(const HashParams &p, int64_t n)
{
    if (p.family == 0) return (size_t)n * (size_t)((p.F + 15) / 16) * 2;
    return (size_t)n * (size_t)p.F * 4;
}

hipError_t launch_hash_mfma(int dtype, const void *X, int64_t x_stride, int64_t n, const HashParams &p, const ProjectWs &ws,
                            int32_t *hashes, int32_t *parts, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int ft16 = (p.F + 15) / 16;
    const int64_t tiles = (int64_t)((p.F + BN - 1) / BN) * ((n + BM - 1) / BM);
    if (tiles > 0x7fffffffll) return hipErrorInvalidConfiguration;
    hipError_t e = hipMemsetAsync(ws.flag_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const dim3 ngrid((unsigned)((n + 3) / 4)), block(256);
    if (dtype == 0) hipLaunchKernelGGL((k_row_norm<float>), ngrid, block, 0, stream, (const float *)X, x_stride, n, p.dim, ws.row_norm, ws.row_flag);
    else hipLaunchKernelGGL((k_row_norm<double>), ngrid, block, 0, stream, (const double *)X, x_stride, n, p.dim, ws.row_norm, ws.row_flag);
    const dim3 grid((unsigned)tiles);
    uint16_t *bits16 = p.family == 0 ? (uint16_t *)ws.codes : nullptr;
    int32_t *codes = p.family == 0 ? nullptr : (int32_t *)ws.codes;
#define DPF_PROJECT(T, FAM)                                                                                              \
    hipLaunchKernelGGL((k_project_mfma<T, FAM>), grid, block, 0, stream, (const T *)X, x_stride, n, p, ws.row_norm, bits16, \
                       codes, ft16, ws.row_flag, ws.flag_rows, ws.flag_count, ws.flag_total)
    if (dtype == 0) { if (p.family == 0) DPF_PROJECT(float, 0); else DPF_PROJECT(float, 1); }
    else { if (p.family == 0) DPF_PROJECT(double, 0); else DPF_PROJECT(double, 1); }
#undef DPF_PROJECT
    const size_t lds = p.family == 0 ? (size_t)ASM_ROWS * ft16 * 2 : 0;
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)((n + ASM_ROWS - 1) / ASM_ROWS)), block, lds, stream, p, n, bits16, codes, ft16,
                       ws.row_flag, hashes, parts);
    // the undecided rows, whole, on the exact kernel
    launch_hash_rows(dtype, X, x_stride, n, p, hashes, parts, ws.flag_rows, ws.flag_count, stream);
    return hipGetLastError();
}

}  // namespace dpf
