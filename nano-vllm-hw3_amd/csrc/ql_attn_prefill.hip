// Causal self-attention over packed variable-length sequences (prefill) for gfx950, flash style:
// scores are never materialised; each workgroup walks the K/V tiles of its query block with an
// online softmax.  head_dim = 128, bf16 in / bf16 out, f32 accumulation, grouped-query heads.
//
// Workgroup = 4 waves = the G = Hq/Hkv query heads of ONE kv head x (128/G) consecutive query rows,
// so each K/V tile staged in LDS serves all query heads that share it.  Each wave owns 32 query rows
// of one head.  Both products are computed transposed so that everything a softmax row needs sits
// in one lane (pair):
//   S^T = K . Q^T   (A = K rows from LDS with 16-byte reads, B = Q^T kept in registers):
//                   lane l holds query (l & 31) and the keys (r&3) + 8*(r>>2) + 4*(l>>5) of a tile
//   O^T += V^T . P^T (A = V^T through the transposing LDS read, B = P^T = the lane's own probabilities):
//                   the accumulator keeps query (l & 31) per lane, so the online rescale is lane-local.
// The k index of the second product is the KEY; its order inside a 16-key MFMA step is free as long
// as both operands agree, and it is chosen to be exactly the order in which the S^T accumulator
// registers already hold the keys - P never moves between lanes.
// K/V tiles (64 keys) are staged through one LDS image per tensor, 256-byte rows with an XOR swizzle of
// the 16-byte chunk index that keeps both the row reads (K) and the transposed reads (V) conflict-free;
// the next tile's global loads are issued before the current tile's math and land in LDS after it.
#include "ql_common.h"

namespace ql {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kTileKeys = 64;

__device__ __forceinline__ int kv_off(int row, int ch) {
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

template <int G>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const unsigned short* __restrict__ q,
                                                           const unsigned short* __restrict__ k,
                                                           const unsigned short* __restrict__ v,
                                                           const int* __restrict__ cu_seqlens,
                                                           unsigned short* __restrict__ out, float c_log2, int Hq,
                                                           int Hkv, int kv_group) {
    constexpr int SUBS = 4 / G;            // 32-row query blocks per head in this workgroup
    constexpr int QROWS = 32 * SUBS;
    __shared__ __attribute__((aligned(256))) unsigned char k_lds[kTileKeys * 256];
    __shared__ __attribute__((aligned(256))) unsigned char v_lds[kTileKeys * 256];

    // blockIdx.y counts groups of G query heads; a group reads the K/V of kv head (group * G) / kv_group, which is
    // the group itself when a kv head's query heads form one group (kv_group == G: ratios 1, 2, 4); other ratios
    // are dealt to several groups per kv head
    const int seq = blockIdx.z, hgrp = blockIdx.y, hk = (hgrp * G) / kv_group;
    const int s0 = cu_seqlens[seq];
    const int L = cu_seqlens[seq + 1] - s0;
    const int q0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * QROWS;      // heaviest query blocks first
    if (q0 >= L) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave % G, sub = wave / G;
    const int hq = hgrp * G + g;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qw0 = q0 + sub * 32;
    const int qrow = qw0 + l31;                                      // this lane's query (within the sequence)
    const int kend = min(L, q0 + QROWS);
    const int ntiles = (kend + kTileKeys - 1) / kTileKeys;

    // Q^T as the B operand: 8 k-steps of 16 dims, lane holds dims 16*ks + 8*hi .. +8 of its query
    bf16x8 qf[8];
    {
        const unsigned short* qp = q + ((size_t)(s0 + min(qrow, L - 1)) * Hq + hq) * 128 + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + 16 * ks);
    }

    // cooperative tile staging: thread moves chunks c = tid + 256*j: row c >> 4, 16-byte chunk c & 15
    const int st_row = tid >> 4, st_ch = tid & 15;
    i32x4 kreg[4], vreg[4];
    auto load_tile = [&](int kt0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = min(kt0 + st_row + 16 * j, L - 1);
            const size_t off = ((size_t)(s0 + key) * Hkv + hk) * 128 + st_ch * 8;
            kreg[j] = *(const i32x4*)(k + off);
            vreg[j] = *(const i32x4*)(v + off);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = kv_off(st_row + 16 * j, st_ch);
            *(i32x4*)(k_lds + o) = kreg[j];
            *(i32x4*)(v_lds + o) = vreg[j];
        }
    };

    f32x16 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // transposed-read address pieces of this lane: 16-lane group gq picks dims 16*gq .. +16 of a 32-dim block
    const int gq = (lane >> 4) & 1, ti = lane & 15, tq = ti >> 2, tp = ti & 3;

    load_tile(0);
    store_tile();
    __syncthreads();
    // Pin the Q fragments as loaded HERE.  Otherwise their loads are sunk to the loop entry, and because the
    // wait counter is positional every MFMA of the loop that reads them also waits for the next tile's K/V
    // loads issued just before it: the prefetch would have to land within the first product of each tile.
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(__builtin_bit_cast(i32x4, qf[ks])));
    for (int it = 0; it < ntiles; ++it) {
        const int kt0 = it * kTileKeys;
        if (it + 1 < ntiles) load_tile(kt0 + kTileKeys);

        // ---- S^T = K . Q^T for the two 32-key halves of the tile
        f32x16 s[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(k_lds + kv_off(32 * j + l31, 2 * ks + hi));
                s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[j], 0, 0, 0);
            }
        }
        // ---- causal mask (only tiles that reach past this wave's first query) and online softmax
        if (kt0 + kTileKeys - 1 > qw0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key > qrow) s[j][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));                          // the other 32 keys of this query
        const float m_new = fmaxf(m_run, mx);                        // finite: key 0 is visible to every query
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_log2);
        const float mc = m_new * c_log2;
        float psum = 0.f;
        bf16x8 pf[2][2];                                             // P^T fragments: [32-key half][16-key step]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][8 * t + e], c_log2, -mc));
                    psum += p;
                    pf[j][t][e] = (bf16_t)p;
                }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {        // skip the rescale while no maximum moved
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r0 = 32 * j + 16 * t + 4 * hi;
                    const int c0 = 4 * d + 2 * gq;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(v_lds + kv_off(r0 + tq, c0 + (tp >> 1)) + 8 * (tp & 1)));
                    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(v_lds + kv_off(r0 + 8 + tq, c0 + (tp >> 1)) + 8 * (tp & 1)));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[j][t], oacc[d], 0, 0, 0);
                }
        __syncthreads();                                             // every wave is done with this tile
        if (it + 1 < ntiles) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- normalise and store: lane holds dims 32*d + 8*(r>>2) + 4*hi + (r&3) of its query
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (qrow >= L) return;
    const float inv = 1.f / l_tot;
    unsigned short* op = out + ((size_t)(s0 + qrow) * Hq + hq) * 128 + 4 * hi;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            u16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = __builtin_bit_cast(unsigned short, (bf16_t)(oacc[d][4 * rq + e] * inv));
            *(u16x4*)(op + 32 * d + 8 * rq) = o4;
        }
}

}  // namespace ql

using namespace ql;

extern "C" int ql_attn_prefill(const void* q, const void* k, const void* v, const int* cu_seqlens, void* out, float scale,
                               int num_seqs, int max_seqlen, int Hq, int Hkv, int head_dim, void* stream) {
    if (num_seqs < 0 || max_seqlen < 0 || Hq <= 0 || Hkv <= 0 ||
        (num_seqs > 0 && max_seqlen > 0 && (!q || !k || !v || !cu_seqlens || !out))) {
        set_error("ql_attn_prefill: bad argument");
        return QL_EINVAL;
    }
    if (head_dim != 128 || Hq % Hkv) { set_error("ql_attn_prefill: head_dim must be 128 and Hq %% Hkv == 0"); return QL_ESHAPE; }
    const int kv_group = Hq / Hkv;
    const int G = (kv_group % 4 == 0) ? 4 : (kv_group % 2 == 0) ? 2 : 1;     // query heads per workgroup
    if (num_seqs == 0 || max_seqlen == 0) return QL_OK;
    if (num_seqs > 65535 || Hq / G > 65535) { set_error("ql_attn_prefill: too many sequences or heads"); return QL_ESHAPE; }
    const int qrows = 128 / G;
    dim3 grid((unsigned)((max_seqlen + qrows - 1) / qrows), (unsigned)(Hq / G), (unsigned)num_seqs), block(256);
    const float c = scale * 1.4426950408889634f;
    hipStream_t s = as_stream(stream);
#define QL_PF(G_)                                                                                              \
    hipLaunchKernelGGL(attn_prefill_kernel<G_>, grid, block, 0, s, (const unsigned short*)q, (const unsigned short*)k, \
                       (const unsigned short*)v, cu_seqlens, (unsigned short*)out, c, Hq, Hkv, kv_group)
    if (G == 1) QL_PF(1); else if (G == 2) QL_PF(2); else QL_PF(4);
#undef QL_PF
    QL_CHECK_LAUNCH("ql_attn_prefill");
    return QL_OK;
}
