// Per-head q/k RMSNorm + rotate-half RoPE + K/V cache store for ONE 16-element piece pair of a (token, head):
// shared by the stand-alone operator (ql_qknorm_rope_store) and by the epilogue of the fused qkv GEMM, so both
// produce the same bits (every multiply-add is spelled out: left to the compiler, the choice of which products
// are contracted into fused multiply-adds differs between the two call sites).  Eight consecutive lanes own a (token, head); lane s holds elements 8s .. 8s+7 (a1) and
// 64+8s .. 64+8s+7 (a2) of the 128-wide head, which are exactly the (x1, x2) pairs the rotation mixes, so only the
// norm's sum of squares crosses lanes (three xor-shuffle steps inside the 8-lane group).
#pragma once
#include "ql_common.h"

namespace ql {

struct RopeArgs {
    const int64_t* positions;       // [T], or null: cos_sin then holds one row per TOKEN, already gathered
    const float* cos_sin;           // [max_pos, 128] f32: cos[0:64] | sin[64:128]   ([T, 128] when positions is null)
    const unsigned short* qw;       // [128] bf16 q-norm weight or null
    const unsigned short* kw;       // [128] bf16 k-norm weight or null
    unsigned short* q_out;          // [T, Hq, 128]
    unsigned short* k_out;          // [T, Hkv, 128] or null
    unsigned short* v_out;          // [T, Hkv, 128] or null
    unsigned short* k_cache;        // [slots, Hkv, 128] or null
    unsigned short* v_cache;
    const int* slot_mapping;        // [T] or null
    float eps;
    int Hq, Hkv;
};

// What a (token, piece) needs from memory besides its own values: the cache slot, the cos/sin pieces of the token's
// position and the norm weights.  Loading them is a chain of dependent reads (position -> table row), so a caller
// with other work to do first (the GEMM) requests them early and keeps them in registers.
struct RopePre {
    float c[8], sn[8];
    u16x8 w1, w2;
    int slot;
};

__device__ __forceinline__ RopePre rope_prefetch(int t, int h, int s8, const RopeArgs& r) {
    RopePre pre;
    pre.slot = r.slot_mapping ? r.slot_mapping[t] : -1;
    const bool is_v = h >= r.Hq + r.Hkv;
    const unsigned short* nw = (h < r.Hq) ? r.qw : r.kw;
    pre.w1 = pre.w2 = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (!is_v) {
        if (nw) { pre.w1 = *(const u16x8*)(nw + s8); pre.w2 = *(const u16x8*)(nw + 64 + s8); }
        const float* cs = r.cos_sin + (r.positions ? r.positions[t] : (int64_t)t) * 128 + s8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pre.c[j] = cs[j]; pre.sn[j] = cs[64 + j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { pre.c[j] = 1.f; pre.sn[j] = 0.f; }
    }
    return pre;
}

// All eight lanes of the group must call this together (it shuffles); t = token, h = head index in
// [0, Hq + 2 Hkv) (q heads, then k heads, then v heads), s8 = 8 * (lane & 7).
__device__ __forceinline__ void rope_store_piece(const u16x8 a1, const u16x8 a2, int t, int h, int s8, const RopeArgs& r,
                                                 const RopePre& pre) {
    const int Hq = r.Hq, Hkv = r.Hkv;
    const int slot = pre.slot;
    if (h >= Hq + Hkv) {   // V: copy only
        const size_t hv = h - Hq - Hkv;
        if (r.v_out) {
            *(u16x8*)(r.v_out + ((size_t)t * Hkv + hv) * 128 + s8) = a1;
            *(u16x8*)(r.v_out + ((size_t)t * Hkv + hv) * 128 + 64 + s8) = a2;
        }
        if (slot >= 0) {
            *(u16x8*)(r.v_cache + ((size_t)slot * Hkv + hv) * 128 + s8) = a1;
            *(u16x8*)(r.v_cache + ((size_t)slot * Hkv + hv) * 128 + 64 + s8) = a2;
        }
        return;
    }
    float x1[8], x2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x1[j] = bf16_bits_to_f32(a1[j]); x2[j] = bf16_bits_to_f32(a2[j]); }
    const bool is_q = h < Hq;
    const unsigned short* nw = is_q ? r.qw : r.kw;
    if (nw) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ss = __builtin_fmaf(x1[j], x1[j], ss); ss = __builtin_fmaf(x2[j], x2[j], ss); }
        ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
        const float inv = rsqrtf(__builtin_fmaf(ss, 1.f / 128.f, r.eps));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x1[j] = round_to_bf16(__fmul_rn(round_to_bf16(__fmul_rn(x1[j], inv)), bf16_bits_to_f32(pre.w1[j])));
            x2[j] = round_to_bf16(__fmul_rn(round_to_bf16(__fmul_rn(x2[j], inv)), bf16_bits_to_f32(pre.w2[j])));
        }
    }
    u16x8 y1, y2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float c = pre.c[j], sn = pre.sn[j];
        y1[j] = f32_to_bf16_bits(__builtin_fmaf(x1[j], c, -__fmul_rn(x2[j], sn)));
        y2[j] = f32_to_bf16_bits(__builtin_fmaf(x2[j], c, __fmul_rn(x1[j], sn)));
    }
    if (is_q) {
        *(u16x8*)(r.q_out + ((size_t)t * Hq + h) * 128 + s8) = y1;
        *(u16x8*)(r.q_out + ((size_t)t * Hq + h) * 128 + 64 + s8) = y2;
    } else {
        const size_t hk = h - Hq;
        if (r.k_out) {
            *(u16x8*)(r.k_out + ((size_t)t * Hkv + hk) * 128 + s8) = y1;
            *(u16x8*)(r.k_out + ((size_t)t * Hkv + hk) * 128 + 64 + s8) = y2;
        }
        if (slot >= 0) {
            *(u16x8*)(r.k_cache + ((size_t)slot * Hkv + hk) * 128 + s8) = y1;
            *(u16x8*)(r.k_cache + ((size_t)slot * Hkv + hk) * 128 + 64 + s8) = y2;
        }
    }
}

}  // namespace ql
