// Device helpers shared by the two hashing paths (the exact vector-ALU kernel and the matrix-core
// projection): hash post-transforms, JVM integer semantics, and the assembly of one table's hash and
// sub-index id from per-function results.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "dpf_device.hpp"

namespace dpf {

// ---------------------------------------------------------------------------
// hash post-transforms (LSH.scala:152-162)
// ---------------------------------------------------------------------------

// significantBits.continueBitsCount(key, Array(6,4,2,1)) -- significantBits.scala:11-67
__device__ inline int32_t continue_bits_count(int32_t key)
{
    const uint32_t k = (uint32_t)key;
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0, count = 0;
    for (int i = 0; i < 28; i++) {
        bool one = (k >> i) & 1u;
        if (one) count++;
        if (!one || i == 27) {
            if (count >= 6) { a0++; a1++; a2++; a3++; }
            else if (count >= 4) { a1++; a2++; a3++; }
            else if (count >= 2) { a2++; a3++; }
            else if (count >= 1) { a3++; }
            count = 0;
        }
    }
    uint32_t tmp = ((uint32_t)a3 << 21) + ((uint32_t)a2 << 14) + ((uint32_t)a1 << 7) + (uint32_t)a0;
    tmp += (k >> 28) << 28;
    return (int32_t)tmp;
}

// Sampling.samplingOneKey -- Sampling.scala:32-39
__device__ inline int32_t sampling_one_key(int32_t key, const int32_t *si)
{
    uint32_t tmp = 0;
    for (int j = 0; j < 32; j++) tmp += (((uint32_t)key >> (si[j] & 31)) & 1u) << (31 - j);
    return (int32_t)tmp;
}

// significantBits.newMethod -- significantBits.scala:113-127.  The angle it
// thresholds depends only on the popcount of the low 28 bits, so the label is a
// 29-entry table (computed on the host with the same formula).
__device__ inline int32_t angle_new_method(int32_t key, const int32_t *label_lut)
{
    const uint32_t k = (uint32_t)key, mask = 0x7f;
    uint32_t label = (uint32_t)label_lut[__popc(k & 0x0fffffffu)];
    return (int32_t)((k & mask) + (((k >> 7) & mask) << 7) + (label << 14) + (((k >> 21) & mask) << 21) +
                     (((k >> 28) & mask) << 28));
}

__device__ inline int32_t apply_type_of_index(int32_t h, const HashParams &p)
{
    switch (p.type_of_index) {
    case 1: return sampling_one_key(h, p.sampling_index);
    case 2: return continue_bits_count(h);
    case 3: return angle_new_method(h, p.angle_label_lut);
    default: return h;
    }
}

// Scala Double.toInt: NaN -> 0, saturating, toward zero (PStable:130)
__device__ inline int32_t d2i_sat(double x)
{
    if (x != x) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)x;
}

// java.util.Arrays.hashCode over the four big-endian bytes of idx (ByteArrayWrapper.scala:11-14)
__device__ inline uint32_t hash_int_bytes(uint32_t r, int32_t idx)
{
#pragma unroll
    for (int s = 24; s >= 0; s -= 8) {
        int32_t byte = (int32_t)(int8_t)(((uint32_t)idx >> s) & 0xff);
        r = 31u * r + (uint32_t)byte;
    }
    return r;
}

// One table's hash and sub-index id from per-function results: bit_of(f) = sign bit of pool function f
// (angle), code_of(f) = truncated bucket index of pool function f (pStable).
//   Angle:187-195   result = result << 1 | sign(dot), then << (32 - m)
//   PStable:122-143 java.util.Arrays.hashCode over the big-endian bytes of the m bucket indices
//   LSH:152-162     typeOfIndex post-transform
//   Part:40-59      set bits of h -> 32-dim 0/1 vector -> the table's own partitioner chain (ascending-bit
//                   fp64 sum, Sim:9-27) -> top pb bits
template <typename BitOf, typename CodeOf>
__device__ inline void assemble_hash(const HashParams &p, int t, BitOf bit_of, CodeOf code_of, int32_t &h_out, int32_t &part_out)
{
    const int32_t *chain = p.chain_fn + (size_t)t * p.m;
    int32_t h;
    if (p.family == 0) {
        uint32_t result = 0;
        for (int j = 0; j < p.m; j++) result = (result << 1) | bit_of(chain[j]);
        h = (int32_t)(result << ((32 - p.m) & 31));
    } else {
        uint32_t r = 1;
        for (int j = 0; j < p.m; j++) r = hash_int_bytes(r, code_of(chain[j]));
        h = (int32_t)r;
    }
    h = apply_type_of_index(h, p);
    int32_t part = 0;
    if (p.pb > 0) {
        const double *pf = p.part_fns + (size_t)t * p.pb * 32;
        uint32_t presult = 0, pr = 1;
        for (int j = 0; j < p.pb; j++) {
            double s = 0.0;
            for (int i = 0; i < 32; i++)
                if (((uint32_t)h >> i) & 1u) s = s + pf[j * 32 + i];
            if (p.family == 0) {
                presult = (presult << 1) | ((s <= 0) ? 0u : 1u);
            } else {
                double v = s + p.part_b[t * p.pb + j];
                v = v / (double)p.part_w[t * p.pb + j];
                pr = hash_int_bytes(pr, d2i_sat(v));
            }
        }
        int32_t ph = p.family == 0 ? (int32_t)(presult << ((32 - p.pb) & 31)) : (int32_t)pr;
        ph = apply_type_of_index(ph, p);
        part = (int32_t)((uint32_t)ph >> ((32 - p.pb) & 31));
    }
    h_out = h;
    part_out = part;
}

}  // namespace dpf
