// Micro-benchmark: how fast can waves pull 64-row tiles of 256-byte rows (random rows of a table,
// every row used `passes` times in different orders) into registers, by access shape?
//   shape 0: the MFMA operand shape: one instruction = 16 rows x 64 bytes (lane -> row l%16, piece l/16)
//   shape 1: whole rows: one instruction = 4 rows x 256 bytes (lane -> row l/16, piece l%16)
//   shape 2: shape 1 + transpose through LDS into the MFMA operand shape (swizzled, conflict-free)
//   shape 3: shape 0 on rows of 192 bytes (three 64-byte requests per row instead of four)
// Build: hipcc --offload-arch=gfx950 -O3 -o rowload_bench tools/rowload_bench.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256, 3) void k_pull(const uint8_t *__restrict__ table, const int32_t *__restrict__ ids, uint32_t ntiles,
                                                 uint32_t *__restrict__ sink)
{
    __shared__ u32x4 s_tile[SHAPE == 2 ? 4 : 1][SHAPE == 2 ? 512 : 1];   // 8 KiB per wave: half a tile at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t W = gridDim.x * 4u;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (uint32_t t = blockIdx.x * 4u + wave; t < ntiles; t += W) {
        u32x4 x[16];
        if (SHAPE == 3) {
            const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
            for (int rb = 0; rb < 4; rb++) {
                const size_t row = (size_t)ids[(size_t)t * 64 + rb * 16 + i16];
                const uint8_t *p = table + row * 192 + 16 * g;
#pragma unroll
                for (int k = 0; k < 3; k++) x[rb * 4 + k] = *(const u32x4 *)(p + 64 * k);
                x[rb * 4 + 3] = x[rb * 4];
            }
        } else if (SHAPE == 0) {
            const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
            for (int rb = 0; rb < 4; rb++) {
                const size_t row = (size_t)ids[(size_t)t * 64 + rb * 16 + i16];
                const uint8_t *p = table + row * 256 + 16 * g;
#pragma unroll
                for (int k = 0; k < 4; k++) x[rb * 4 + k] = *(const u32x4 *)(p + 64 * k);
            }
        } else {
            const int32_t mine = ids[(size_t)t * 64 + lane];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const size_t row = (size_t)__shfl(mine, 4 * j + (lane >> 4));
                x[j] = *(const u32x4 *)(table + row * 256 + 16 * (lane & 15));
            }
            if (SHAPE == 2) {
                const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    // rows 32h .. 32h+31 of the tile: 8 instructions' worth
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int row = 4 * j + (lane >> 4);                 // within the half
                        s_tile[wave][row * 16 + ((lane & 15) ^ (row & 15))] = x[8 * h + j];
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int rb = 0; rb < 2; rb++)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int row = rb * 16 + i16;
                            x[8 * h + rb * 4 + k] = s_tile[wave][row * 16 + ((4 * k + g) ^ (row & 15))];
                        }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) acc ^= x[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

int main(int argc, char **argv)
{
    const uint32_t nrows = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 20);
    const int passes = argc > 2 ? atoi(argv[2]) : 29;
    const int only = argc > 3 ? atoi(argv[3]) : -1;
    const size_t total = (size_t)nrows * passes;
    const uint32_t ntiles = (uint32_t)(total / 64);
    std::vector<int32_t> ids(total);
    uint64_t s = 88172645463325252ull;
    for (int p = 0; p < passes; p++) {
        // a different odd multiplier per pass: a permutation of the rows (nrows is a power of two)
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t a = (uint32_t)(s | 1u) * 2654435761u | 1u, b = (uint32_t)(s >> 32);
        for (uint32_t i = 0; i < nrows; i++) ids[(size_t)p * nrows + i] = (int32_t)((i * a + b) & (nrows - 1));
    }
    uint8_t *d_table;
    int32_t *d_ids;
    uint32_t *d_sink;
    CK(hipMalloc((void **)&d_table, (size_t)nrows * 256));
    CK(hipMemset(d_table, 1, (size_t)nrows * 256));
    CK(hipMalloc((void **)&d_ids, total * 4));
    CK(hipMemcpy(d_ids, ids.data(), total * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_sink, 4));
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid(cus * 3), block(256);
    for (int shape = 0; shape < 4; shape++) {
        if (only >= 0 && shape != only) continue;
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            if (shape == 0) hipLaunchKernelGGL(k_pull<0>, grid, block, 0, 0, d_table, d_ids, ntiles, d_sink);
            if (shape == 1) hipLaunchKernelGGL(k_pull<1>, grid, block, 0, 0, d_table, d_ids, ntiles, d_sink);
            if (shape == 2) hipLaunchKernelGGL(k_pull<2>, grid, block, 0, 0, d_table, d_ids, ntiles, d_sink);
            if (shape == 3) hipLaunchKernelGGL(k_pull<3>, grid, block, 0, 0, d_table, d_ids, ntiles, d_sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("shape %d: %u rows x %d B, %d passes: %.3f ms, %.2f TB/s of row bytes, %.1f G rows/s\n", shape, nrows, shape == 3 ? 192 : 256,
               passes, best, (double)total * (shape == 3 ? 192 : 256) / best / 1e9, (double)total / best / 1e6);
    }
    return 0;
}
