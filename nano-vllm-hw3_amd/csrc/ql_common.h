// Shared device helpers for libqlinear (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/qlinear.h"

namespace ql {

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

void set_error(const char* fmt, ...);

// ---- number formats ------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned)b << 16);
}

// round-to-nearest-even f32 -> bf16 bit pattern (NaN stays NaN)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ float round_to_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

// f32 -> OCP e4m3fn, round-to-nearest-even, NaN pattern (0x7F) for anything that rounds above 448:
// the exact bit procedure torch uses for .to(float8_e4m3fn), so quantized weights are bit-identical
// to the reference's.
__device__ __forceinline__ unsigned char f32_to_e4m3fn(float f) {
    unsigned bits = __builtin_bit_cast(unsigned, f);
    const unsigned sign = bits & 0x80000000u;
    bits ^= sign;
    unsigned res;
    if (bits >= (1087u << 20)) {
        res = 0x7F;
    } else if (bits < (121u << 23)) {
        const unsigned dm = 141u << 23;
        float t = __builtin_bit_cast(float, bits) + __builtin_bit_cast(float, dm);
        res = (__builtin_bit_cast(unsigned, t) - dm) & 0xFFu;
    } else {
        const unsigned odd = (bits >> 20) & 1u;
        bits += ((unsigned)(7 - 127) << 23) + 0x7FFFFu + odd;
        res = (bits >> 20) & 0xFFu;
    }
    return (unsigned char)(res | (sign >> 24));
}

__device__ __forceinline__ float e4m3fn_to_f32(unsigned char b) {
    const unsigned s = (unsigned)(b & 0x80) << 24;
    const unsigned e = (b >> 3) & 0xF, m = b & 7;
    float v;
    if (e == 0xF && m == 7) v = __builtin_nanf("");
    else if (e == 0) v = (float)m * 0.001953125f;               // m/8 * 2^-6
    else v = __builtin_bit_cast(float, ((e + 120u) << 23) | (m << 20));
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) | s);
}

// bf16(silu(g) * u) on bf16-representable inputs: one definition shared by the stand-alone operator
// and the fused GEMM epilogue, so both produce the same bits (hardware reciprocal and conversion)
__device__ __forceinline__ unsigned short silu_mul_bf16(float g, float u) {
    const float s = g * __builtin_amdgcn_rcpf(1.f + __expf(-g));
    return __builtin_bit_cast(unsigned short, (bf16_t)(s * u));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

#define QL_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            ql::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return QL_ELAUNCH;                                                  \
        }                                                                       \
    } while (0)

}  // namespace ql
