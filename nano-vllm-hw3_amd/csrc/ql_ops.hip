// The operators between the quantized GEMMs of a decoder layer, each as ONE launch so that a decode
// step is ~10 kernels per layer: residual-add + RMSNorm, per-head q/k RMSNorm + rotary embedding +
// KV-cache store, SiLU*mul, and single-token (decode) attention over a paged KV cache.
// All are HBM/L2-bound: 16-byte accesses per lane, f32 math, bf16 storage.
#include "ql_common.h"
#include <type_traits>
#include "ql_rope.h"

namespace ql {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- y = RMSNorm(x [+ residual]) * w ; residual_out = bf16(x + residual) ------------------------
// One workgroup per row; the row stays in registers between the two passes (H <= 8192).
// QDT = QL_INT8 / QL_FP8: the normalised row is not stored as bf16 but quantized per row (dynamic
// activation quantization of the W8A8 linears, the reference's per_row_quant on x): q = cast(y * Q / max|y|),
// scale = max|y| / Q with y the bf16 values the plain kernel would have stored, so (q, scale) are bit-identical
// to ql_add_rmsnorm followed by ql_quant_per_row, without the launch and the bf16 round trip in between.
// Where x comes from when the producing GEMM left its split-K reduction to this launch (ql_gemm_w8a16_partial):
// x[t, n] = bf16( (sum over slices, in slice order, of part[s][t][n]) * sw[n] (+ bias[n]) ), the GEMM's own epilogue.
struct PartialSrc {
    const float* part;            // [S][T][H] f32, or null: x is read as bf16
    const float* sw;              // per-row [H] (stride 1), per-tensor [1] (stride 0), or null = 1 (per-group: already scaled)
    const unsigned short* bias;   // [H] bf16 or null
    int64_t slab;                 // T * H
    int S, sw_stride, bias_after_round;
    const float* sx;              // W8A8: per-token activation scales [T], applied before sw; else null
    int is_int;                   // the slabs hold int32 sums (int8 x int8): added as integers, converted once
};

template <int NV, int QDT>   // 16-byte vectors per thread
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(const unsigned short* __restrict__ x,
                                                          const unsigned short* __restrict__ res,
                                                          const unsigned short* __restrict__ w,
                                                          unsigned short* __restrict__ out,
                                                          unsigned short* __restrict__ res_out, int H, float eps,
                                                          unsigned char* __restrict__ q_out,
                                                          float* __restrict__ scale_out, const PartialSrc ps) {
    __shared__ float sh[8];
    const size_t row = blockIdx.x;
    const int nvec = H >> 3;
    float v[NV][8];
    u16x8 ww[NV];     // norm weights: requested with the row, so their latency is not paid behind the reduction
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nvec) {
            u16x8 a, b = {0, 0, 0, 0, 0, 0, 0, 0};
            ww[j] = *(const u16x8*)(w + c * 8);
            if (res) b = *(const u16x8*)(res + row * H + c * 8);
            if (ps.part) {
                // everything this piece needs is requested before the first wait: residual (above), scales, bias
                // and the slices' partial sums (four slices in flight at a time; the additions stay in slice order)
                float sc[8];
                u16x8 bb = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ps.sw && ps.sw_stride) {
                    const float4 s0 = *(const float4*)(ps.sw + c * 8), s1 = *(const float4*)(ps.sw + c * 8 + 4);
                    sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
                } else {
                    const float s0 = ps.sw ? ps.sw[0] : 1.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) sc[k] = s0;
                }
                if (ps.bias) bb = *(const u16x8*)(ps.bias + c * 8);
                float acc[8];
                int acci[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { acc[k] = 0.f; acci[k] = 0; }
                const float sxr = ps.sx ? ps.sx[row] : 1.f;
                const float* src = ps.part + row * H + c * 8;
                for (int s0 = 0; s0 < ps.S; s0 += 4) {
                    float4 p0[4], p1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float* q = src + (size_t)min(s0 + u, ps.S - 1) * ps.slab;
                        p0[u] = *(const float4*)q;
                        p1[u] = *(const float4*)(q + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (s0 + u < ps.S) {
                            if (ps.is_int) {
                                acci[0] += __builtin_bit_cast(int, p0[u].x); acci[1] += __builtin_bit_cast(int, p0[u].y);
                                acci[2] += __builtin_bit_cast(int, p0[u].z); acci[3] += __builtin_bit_cast(int, p0[u].w);
                                acci[4] += __builtin_bit_cast(int, p1[u].x); acci[5] += __builtin_bit_cast(int, p1[u].y);
                                acci[6] += __builtin_bit_cast(int, p1[u].z); acci[7] += __builtin_bit_cast(int, p1[u].w);
                            } else {
                                acc[0] += p0[u].x; acc[1] += p0[u].y; acc[2] += p0[u].z; acc[3] += p0[u].w;
                                acc[4] += p1[u].x; acc[5] += p1[u].y; acc[6] += p1[u].z; acc[7] += p1[u].w;
                            }
                        }
                }
                if (ps.is_int) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = (float)acci[k];
                }
                if (ps.sx) {                  // the W8A8 epilogue: (sum * sx[m]) * sw[n], in that order
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = __fmul_rn(acc[k], sxr);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float v = __fmul_rn(acc[k], sc[k]);
                    if (ps.bias) {
                        const float bv = bf16_bits_to_f32(bb[k]);
                        v = ps.bias_after_round ? __fadd_rn((float)(bf16_t)v, bv) : __fadd_rn(v, bv);
                    }
                    a[k] = __builtin_bit_cast(unsigned short, (bf16_t)v);
                }
            } else {
                a = *(const u16x8*)(x + row * H + c * 8);
            }
            if (res) {
                u16x8 r;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[j][k] = bf16_bits_to_f32(a[k]) + bf16_bits_to_f32(b[k]);
                    r[k] = f32_to_bf16_bits(v[j][k]);
                }
                *(u16x8*)(res_out + row * H + c * 8) = r;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[j][k] = bf16_bits_to_f32(a[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) ss += v[j][k] * v[j][k];
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = sh[0] + sh[1] + sh[2] + sh[3];
    const float inv = rsqrtf(ss / (float)H + eps);
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nvec) {
            u16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                o[k] = f32_to_bf16_bits(round_to_bf16(v[j][k] * inv) * bf16_bits_to_f32(ww[j][k]));
            if constexpr (QDT < 0) {
                *(u16x8*)(out + row * H + c * 8) = o;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[j][k] = bf16_bits_to_f32(o[k]);
                    amax = fmaxf(amax, fabsf(v[j][k]));
                }
            }
        }
    }
    if constexpr (QDT >= 0) {
        constexpr float QMAX = (QDT == QL_INT8) ? 127.f : 448.f;
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0) sh[4 + (threadIdx.x >> 6)] = amax;
        __syncthreads();
        const float row_max = fmaxf(fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7])), 1e-8f);
        const float sc_inv = QMAX / row_max;
        if (threadIdx.x == 0) scale_out[row] = row_max / QMAX;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < nvec) {
                unsigned long long o8 = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float y = v[j][k] * sc_inv;
                    unsigned b;
                    if (QDT == QL_INT8) b = (unsigned)(unsigned char)(signed char)(int)roundf(y);
                    else b = f32_to_e4m3fn(y);
                    o8 |= (unsigned long long)b << (8 * k);
                }
                *(unsigned long long*)(q_out + row * H + c * 8) = o8;
            }
        }
    }
}

// ---- out[t, i] = silu(x[t, i]) * x[t, I + i] -------------------------------------------------------
__global__ __launch_bounds__(256) void silu_mul_kernel(const unsigned short* __restrict__ x,
                                                       unsigned short* __restrict__ out, int64_t T, int I) {
    const int64_t vec_per_row = I >> 3;
    const int64_t total = T * vec_per_row;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t t = idx / vec_per_row, c = idx % vec_per_row;
        const u16x8 a = *(const u16x8*)(x + t * 2 * I + c * 8);
        const u16x8 b = *(const u16x8*)(x + t * 2 * I + I + c * 8);
        u16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o[k] = silu_mul_bf16(bf16_bits_to_f32(a[k]), bf16_bits_to_f32(b[k]));
        }
        *(u16x8*)(out + t * I + c * 8) = o;
    }
}

// ---- per-head RMSNorm (optional) + rotate-half RoPE on q and k, and K/V scatter into the paged cache
// Eight lanes per (token, head); the arithmetic lives in ql_rope.h (shared with the fused qkv GEMM epilogue).
__global__ __launch_bounds__(256) void qknorm_rope_store_kernel(const unsigned short* __restrict__ qkv, int T, const RopeArgs r) {
    const int heads = r.Hq + 2 * r.Hkv;
    const int64_t pair = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;      // (token, head) index
    const int s8 = (threadIdx.x & 7) * 8;
    if (pair >= (int64_t)T * heads) return;
    const int t = (int)(pair / heads), h = (int)(pair % heads);
    const unsigned short* src = qkv + ((size_t)t * heads + h) * 128 + s8;
    const u16x8 a1 = *(const u16x8*)src, a2 = *(const u16x8*)(src + 64);
    rope_store_piece(a1, a2, t, h, s8, r, rope_prefetch(t, h, s8, r));
}

// ---- decode attention over a paged KV cache ------------------------------------------------------
// grid (batch, kv_head, chunk): one workgroup per chunk of CHUNK tokens of a sequence, so long and
// short sequences load the chip evenly (a grid of one workgroup per sequence ends on its longest
// sequence).  The G = Hq/Hkv query heads of a kv head share every K/V byte that is read.
// A wave covers 8 tokens per step: lane = 8 * token_slot + dim_segment, two 16-byte (8 x bf16) loads
// of K and of V per lane and token (dimensions [8 seg, 8 seg + 8) and 64 further on, so a token's lanes
// read contiguous 128-byte runs), 4 tokens per slot in flight (2 with four heads per kv head).  Online
// softmax over a slot's tokens in flight; slot pairs merge by a DPP exchange and the resulting 16 states
// of the workgroup are merged through LDS.  A sequence with more than one chunk
// stores its per-chunk (max, sum, weighted V) state and a second, tiny launch merges the chunks in
// order.  (Merging in-kernel by the last chunk to arrive measured slower both ways: behind agent-scope
// fences the L2 write-back / invalidate per chunk halves the stream rate, and with fence-free
// write-through states the completion wait at the end of every workgroup still costs ~2 us per launch.)
#ifndef QL_ATTN_CHUNK
#define QL_ATTN_CHUNK 512
#endif
#ifndef QL_ATTN_U
#define QL_ATTN_U 4
#endif
#ifndef QL_ATTN_LPT
#define QL_ATTN_LPT 8
#endif
constexpr int kAttnChunk = QL_ATTN_CHUNK;
template <int G>
__global__ __launch_bounds__(256) void paged_attn_decode_kernel(
    const unsigned short* __restrict__ q, const unsigned short* __restrict__ k_cache,
    const unsigned short* __restrict__ v_cache, const int* __restrict__ block_tables,
    const int* __restrict__ context_lens, unsigned short* __restrict__ out, float scale, int Hq, int Hkv,
    int block_size, int max_blocks, int max_chunks, float* __restrict__ partial, float* __restrict__ rowmax_out,
    int kv_heads, int kv_group) {
    // Hkv / hk count GROUPS of G query heads that share one launch workgroup.  Normally a group is a kv head
    // (kv_group == G, kv_heads == Hkv); for head ratios no kernel is built for (5, 8, ...) a kv head's query heads
    // are dealt to several groups of 1, 2 or 4, each of which reads the K/V of kv head (hk * G) / kv_group.
    __shared__ float st_m[16][G], st_l[16][G];
    __shared__ float st_acc[16][G][128];
    const int b = blockIdx.x, hk = blockIdx.y, sp = blockIdx.z;
    const int ctx = context_lens[b];
    const int nsp = (ctx + kAttnChunk - 1) / kAttnChunk;
    if (ctx == 0) {
        if (sp == 0) {
            for (int o = threadIdx.x; o < G * 128; o += 256) out[((size_t)b * Hq + hk * G) * 128 + o] = 0;
            if (rowmax_out && threadIdx.x == 0) rowmax_out[(size_t)b * Hkv + hk] = 0.f;
        }
        return;
    }
    if (sp >= nsp) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // LPT lanes share a cached token, each holding DPL = 128 / LPT of its dimensions.  Everything that is per token and
    // head (the lane reduction of q . k, the softmax arithmetic) is repeated on all of a token's lanes, so fewer lanes
    // per token mean less repeated work: 8 lanes x 16 dimensions (two 16-byte loads per tensor and token) instead
    // of 16 x 8 cuts the vector instructions per token by a quarter.
    constexpr int LPT = QL_ATTN_LPT, DPL = 128 / LPT, NP = DPL / 2, NL = DPL / 8, SPW = 64 / LPT;
    static_assert(LPT == 8 || LPT == 16, "lanes per token");
    const int slot = lane / LPT, seg = lane % LPT;
    // the lane's h-th run of 8 dimensions: a token's lanes together read whole contiguous 16 * LPT bytes per load
    auto dim0 = [&](int h) { return h * (8 * LPT) + seg * 8; };
    // The arithmetic per cached token is what bounds this kernel once a kv head serves four query heads, so it is
    // written on pairs: q . k as bf16 dot-product instructions on the operands as they lie in memory (products
    // exact, f32 sums; the softmax scale is applied to the finished dot product), the reduction over the lanes of
    // a token by DPP permutes inside the adds (no LDS traffic), and the output update as packed f32 operations.
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    int qp[G][NP];                // packed bf16 pairs of q, as loaded
    f32x2_t acc[G][NP];
    float m[G], l[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int h = 0; h < NL; ++h) {
            const i32x4 a = *(const i32x4*)(q + ((size_t)b * Hq + hk * G + g) * 128 + dim0(h));
#pragma unroll
            for (int j = 0; j < 4; ++j) { qp[g][4 * h + j] = a[j]; acc[g][4 * h + j] = f32x2_t{0.f, 0.f}; }
        }
        m[g] = -1e30f;
        l[g] = 0.f;
    }
    const float scale_log2 = scale * 1.4426950408889634f;
    const int* bt = block_tables + (size_t)b * max_blocks;
    const int tbeg = sp * kAttnChunk;
    const int tend = min(ctx, tbeg + kAttnChunk);                 // tokens [tbeg, tend) of the sequence
    const size_t hoff = (size_t)((hk * G) / kv_group) * 128;
    // tokens per slot per iteration: 2 * NL * U 16-byte loads in flight per lane (four heads x 16 dimensions of state
    // leave registers for two)
    constexpr int U = (LPT == 8 && G == 4) ? QL_ATTN_U / 2 : QL_ATTN_U;
    constexpr int TPI = 4 * SPW;   // tokens the workgroup's slots take per step of u
    // A chunk normally spans at most two cache blocks: their table entries are fetched once (uniform, so scalar
    // loads), which takes a dependent table load and an integer division out of every iteration.
    const int b0 = tbeg / block_size;
    const int bound = (b0 + 1) * block_size;                      // first token of the chunk's second block
    const bool two = (tbeg + kAttnChunk - 1) / block_size <= b0 + 1;
    const int e0 = bt[b0], e1 = bt[min(b0 + 1, max_blocks - 1)];
    const int base0 = e0 * block_size - b0 * block_size;          // cache row of token t = base + t
    const int base1 = e1 * block_size - bound;
    // sum over a token's lanes, the total left in every one of them
    auto lane_sum = [&](float s) -> float {
        auto dpp = [](float v, auto ctrl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, false));
        };
        if constexpr (LPT == 16) {                                // one DPP row: rotations by 8, 4, 2, 1
            s += dpp(s, std::integral_constant<int, 0x128>{});
            s += dpp(s, std::integral_constant<int, 0x124>{});
            s += dpp(s, std::integral_constant<int, 0x122>{});
            s += dpp(s, std::integral_constant<int, 0x121>{});
        } else {                                                  // half a row: lane ^ 1, lane ^ 2, then 7 - lane
            s += dpp(s, std::integral_constant<int, 0xB1>{});     // quad_perm [1, 0, 3, 2]
            s += dpp(s, std::integral_constant<int, 0x4E>{});     // quad_perm [2, 3, 0, 1]
            s += dpp(s, std::integral_constant<int, 0x141>{});    // row_half_mirror
        }
        return s;
    };
    for (int t0 = tbeg + wave * SPW + slot; t0 < tend; t0 += TPI * U) {
        i32x4 kk[U][NL], vv[U][NL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(t0 + TPI * u, tend - 1);            // clamped: no divergent loads
            const size_t row = two ? (size_t)((t < bound ? base0 : base1) + t)
                                   : (size_t)bt[t / block_size] * block_size + (t % block_size);
            const size_t off = row * kv_heads * 128 + hoff;
            // read-once stream: the non-temporal hint measured +5 % on the whole launch
#pragma unroll
            for (int h = 0; h < NL; ++h) {
                kk[u][h] = __builtin_nontemporal_load((const i32x4*)(k_cache + off + dim0(h)));
                vv[u][h] = __builtin_nontemporal_load((const i32x4*)(v_cache + off + dim0(h)));
            }
        }
        // ---- scores of the U tokens of this slot for the G heads
        // q . k: bf16 dot-product instructions written inline (the compiler's selection of the builtin reused one
        // operand pair for all four of a head).  The compiler does not know these are dot instructions, whose result
        // needs up to four wait states before another instruction may read or overwrite it, so the statements are
        // kept in order (volatile) and every result passes through a final wait statement.
        float sc[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                sc[u][g] = 0.f;
#pragma unroll
                for (int h = 0; h < NL; ++h)
                    asm volatile("s_nop 0\n\tv_dot2c_f32_bf16 %0, %1, %5\n\tv_dot2c_f32_bf16 %0, %2, %6\n\t"
                                 "v_dot2c_f32_bf16 %0, %3, %7\n\tv_dot2c_f32_bf16 %0, %4, %8"
                                 : "+v"(sc[u][g])
                                 : "v"(qp[g][4 * h]), "v"(qp[g][4 * h + 1]), "v"(qp[g][4 * h + 2]), "v"(qp[g][4 * h + 3]),
                                   "v"(kk[u][h][0]), "v"(kk[u][h][1]), "v"(kk[u][h][2]), "v"(kk[u][h][3]));
            }
        }
        // (volatile statements keep their order: the wait follows the last dot block, and the empty statements, which
        // every result passes through, follow the wait)
        asm volatile("s_nop 3" : "+v"(sc[U - 1][G - 1]));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) asm volatile("" : "+v"(sc[u][g]));
        // ---- online softmax over the U tokens at once: one rescale of the running state per iteration
        // (scores in log2 units: exp2 is the hardware's exponential; m is converted back when it leaves the loop)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sv[U], mn = m[g];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float s = lane_sum(sc[u][g]);
                sv[u] = (t0 + TPI * u < tend) ? s * scale_log2 : -1e30f;     // (tokens past the chunk: weight 0)
                mn = fmaxf(mn, sv[u]);
            }
            const float alpha = __builtin_amdgcn_exp2f(m[g] - mn);
            float pe[U], psum = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) { pe[u] = __builtin_amdgcn_exp2f(sv[u] - mn); psum += pe[u]; }
            l[g] = l[g] * alpha + psum;
            m[g] = mn;
            const f32x2_t a2 = {alpha, alpha};
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[g][j] = acc[g][j] * a2;
#pragma unroll
            for (int u = 0; u < U; ++u) sc[u][g] = pe[u];
        }
        // ---- output update: V of each token widened once, shared by the G heads
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const unsigned vd = (unsigned)vv[u][j >> 2][j & 3];
                const f32x2_t vf = {__builtin_bit_cast(float, vd << 16), __builtin_bit_cast(float, vd & 0xFFFF0000u)};
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g][j] = __builtin_elementwise_fma(f32x2_t{sc[u][g], sc[u][g]}, vf, acc[g][j]);
            }
        }
    }
    if constexpr (LPT == 8) {
        // two slots share a 16-lane row: merge their states (lane ^ 8 holds the other slot's same dimensions), so that
        // the workgroup hands 16 states to the reduction below, as with 16 lanes per token
#pragma unroll
        for (int g = 0; g < G; ++g) {
            auto other = [](float v) {
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
            };
            const float mo = other(m[g]), lo = other(l[g]);
            const float mn = fmaxf(m[g], mo);
            const float wa = __builtin_amdgcn_exp2f(m[g] - mn), wb = __builtin_amdgcn_exp2f(mo - mn);
            l[g] = l[g] * wa + lo * wb;
            m[g] = mn;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const float o0 = other(acc[g][j][0]), o1 = other(acc[g][j][1]);
                acc[g][j] = f32x2_t{acc[g][j][0] * wa + o0 * wb, acc[g][j][1] * wa + o1 * wb};
            }
        }
    }
    // (the per-slot and per-chunk states below carry m in natural-log units)
#pragma unroll
    for (int g = 0; g < G; ++g) m[g] *= 0.6931471805599453f;
    const int sidx = wave * 4 + (lane >> 4);
    if (LPT == 16 || (lane & 8) == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (seg == 0) { st_m[sidx][g] = m[g]; st_l[sidx][g] = l[g]; }
#pragma unroll
            for (int j = 0; j < DPL; ++j) st_acc[sidx][g][dim0(j >> 3) + (j & 7)] = acc[g][j >> 1][j & 1];
        }
    }
    __syncthreads();
    const size_t tile = (size_t)b * Hkv + hk;
    float* mypart = partial + ((tile * max_chunks + sp) * G) * 130;
    float amax = 0.f;                                              // |max| of the bf16 outputs this thread stores
    for (int o = threadIdx.x; o < G * 128; o += 256) {
        const int g = o >> 7, d = o & 127;
        float mm = -1e30f;
#pragma unroll
        for (int s = 0; s < 16; ++s) mm = fmaxf(mm, st_m[s][g]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float wgt = __expf(st_m[s][g] - mm);
            num += wgt * st_acc[s][g][d];
            den += wgt * st_l[s][g];
        }
        if (nsp == 1) {
            const unsigned short ob = f32_to_bf16_bits(num / den);
            out[((size_t)b * Hq + hk * G + g) * 128 + d] = ob;
            amax = fmaxf(amax, fabsf(bf16_bits_to_f32(ob)));
        } else {
            mypart[g * 130 + 2 + d] = num;
            if (d == 0) { mypart[g * 130] = mm; mypart[g * 130 + 1] = den; }
        }
    }
    // what a consumer that quantizes this row on the fly needs: the |max| over this kv head's G x 128 outputs
    // (single-chunk sequences; the merge launch writes it for the others)
    if (rowmax_out && nsp == 1) {
        __syncthreads();                                           // (st_m is free again)
        amax = wave_max(amax);
        if (lane == 0) st_m[wave][0] = amax;
        __syncthreads();
        if (threadIdx.x == 0)
            rowmax_out[(size_t)b * Hkv + hk] = fmaxf(fmaxf(st_m[0][0], st_m[1][0]), fmaxf(st_m[2][0], st_m[3][0]));
    }
}

// merge the per-chunk states of sequences longer than one chunk (grid (batch, kv_head))
template <int G>
__global__ __launch_bounds__(256) void paged_attn_combine_kernel(const int* __restrict__ context_lens,
                                                                 const float* __restrict__ partial,
                                                                 unsigned short* __restrict__ out, int Hq, int Hkv,
                                                                 int max_chunks, float* __restrict__ rowmax_out) {
    __shared__ float sh[4];
    const int b = blockIdx.x, hk = blockIdx.y;
    const int nsp = (context_lens[b] + kAttnChunk - 1) / kAttnChunk;
    if (nsp <= 1) return;
    const float* parts = partial + (((size_t)b * Hkv + hk) * max_chunks) * G * 130;
    float amax = 0.f;
    for (int o = threadIdx.x; o < G * 128; o += 256) {
        const int g = o >> 7, d = o & 127;
        float mm = -1e30f;
        for (int s = 0; s < nsp; ++s) mm = fmaxf(mm, parts[(s * G + g) * 130]);
        float num = 0.f, den = 0.f;
        for (int s = 0; s < nsp; ++s) {
            const float* ps = parts + (s * G + g) * 130;
            const float wgt = __expf(ps[0] - mm);
            num += wgt * ps[2 + d];
            den += wgt * ps[1];
        }
        const unsigned short ob = f32_to_bf16_bits(num / den);
        out[((size_t)b * Hq + hk * G + g) * 128 + d] = ob;
        amax = fmaxf(amax, fabsf(bf16_bits_to_f32(ob)));
    }
    if (rowmax_out) {
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) rowmax_out[(size_t)b * Hkv + hk] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    }
}

// The merge launch for a W8A8 o_proj: one workgroup per sequence finishes ALL its heads (single-chunk heads were
// written by the attention launch, the others are merged here with paged_attn_combine_kernel's arithmetic), takes the
// row's |max| and writes the row quantized per ql_quant_per_row - the per_row_quant(x) of per_row_matmul
// (nanovllm/utils/quantization.py:323) without a launch of its own.  Two passes over the row's own bf16 outputs.
template <int G, int DT>
__global__ __launch_bounds__(256) void paged_attn_finalize_quant_kernel(const int* __restrict__ context_lens,
                                                                        const float* __restrict__ partial,
                                                                        unsigned short* __restrict__ out,
                                                                        unsigned char* __restrict__ q_out,
                                                                        float* __restrict__ scale_out, int Hq, int Hkv,
                                                                        int max_chunks) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const int nsp = (context_lens[b] + kAttnChunk - 1) / kAttnChunk;
    const int E = Hq * 128;
    unsigned short* orow = out + (size_t)b * E;
    float amax = 0.f;
    for (int e = threadIdx.x * 8; e < E; e += 256 * 8) {
        u16x8 o8;
        if (nsp <= 1) {
            o8 = *(const u16x8*)(orow + e);
        } else {
            const int h = e >> 7, hk = h / G, g = h % G, d0 = e & 127;
            const float* parts = partial + (((size_t)b * Hkv + hk) * max_chunks) * G * 130;
            float mm = -1e30f;
            for (int s = 0; s < nsp; ++s) mm = fmaxf(mm, parts[(s * G + g) * 130]);
            float num[8], den = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) num[k] = 0.f;
            for (int s = 0; s < nsp; ++s) {
                const float* ps = parts + (s * G + g) * 130;
                const float wgt = __expf(ps[0] - mm);
#pragma unroll
                for (int k = 0; k < 8; ++k) num[k] += wgt * ps[2 + d0 + k];
                den += wgt * ps[1];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) o8[k] = f32_to_bf16_bits(num[k] / den);
            *(u16x8*)(orow + e) = o8;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(bf16_bits_to_f32(o8[k])));
    }
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = amax;
    __syncthreads();
    constexpr float QMAX = (DT == QL_INT8) ? 127.f : 448.f;
    const float row_max = fmaxf(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])), 1e-8f);
    const float sc_inv = QMAX / row_max;
    if (threadIdx.x == 0) scale_out[b] = row_max / QMAX;
    for (int e = threadIdx.x * 8; e < E; e += 256 * 8) {
        const u16x8 o8 = *(const u16x8*)(orow + e);            // (this thread's own stores, or the attention launch's)
        unsigned long long o = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float y = bf16_bits_to_f32(o8[k]) * sc_inv;
            unsigned bq;
            if (DT == QL_INT8) bq = (unsigned)(unsigned char)(signed char)(int)roundf(y);
            else bq = f32_to_e4m3fn(y);
            o |= (unsigned long long)bq << (8 * k);
        }
        *(unsigned long long*)(q_out + (size_t)b * E + e) = o;
    }
}

// ---- greedy sampling: index of the row maximum of bf16 logits (smallest index among equal maxima) ---------
// One workgroup per row, 16-byte loads; the reference's sampler takes the argmax of the logits
// (nanovllm/layers/sampler.py:10-12).  NaNs never win.
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const unsigned short* __restrict__ x,
                                                           int64_t* __restrict__ out, int64_t V) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const unsigned short* row = x + (size_t)blockIdx.x * V;
    float best = -INFINITY;
    int bi = 0x7FFFFFFF;
    const int64_t nvec = V >> 3;
    for (int64_t c = threadIdx.x; c < nvec; c += 1024) {
        const u16x8 a = __builtin_nontemporal_load((const u16x8*)(row + c * 8));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = bf16_bits_to_f32(a[k]);
            if (v > best) { best = v; bi = (int)(c * 8 + k); }     // ascending index within a thread: first maximum wins
        }
    }
    for (int64_t i = (nvec << 3) + threadIdx.x; i < V; i += 1024) {
        const float v = bf16_bits_to_f32(row[i]);
        if (v > best) { best = v; bi = (int)i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        out[blockIdx.x] = (bi == 0x7FFFFFFF) ? 0 : bi;
    }
}

}  // namespace ql

using namespace ql;

extern "C" {

static int add_rmsnorm_launch(const char* fn, int qdt, const void* x, const void* residual, const void* weight, float eps,
                              void* out, void* residual_out, void* q_out, float* scale_out, int64_t T, int64_t H,
                              void* stream, PartialSrc ps = PartialSrc{}) {
    if (!weight || T < 0 || H <= 0 || (T > 0 && !x && !ps.part) || (residual && !residual_out) ||
        (T > 0 && (qdt < 0 ? !out : (!q_out || !scale_out)))) {
        set_error("%s: bad argument", fn);
        return QL_EINVAL;
    }
    if (qdt >= 0 && qdt != QL_INT8 && qdt != QL_FP8) { set_error("%s: dtype=%d", fn, qdt); return QL_EINVAL; }
    if (H % 8 || H > 8192) { set_error("%s: H=%lld must be a multiple of 8 and <= 8192", fn, (long long)H); return QL_ESHAPE; }
    if (T == 0) return QL_OK;
    if (T > 0x7FFFFFFF) { set_error("%s: T too large", fn); return QL_ESHAPE; }
    const int nv = (int)((H / 8 + 255) / 256);
    dim3 grid((unsigned)T), block(256);
    hipStream_t s = as_stream(stream);
#define QL_RMS(NV, QDT) hipLaunchKernelGGL((add_rmsnorm_kernel<NV, QDT>), grid, block, 0, s, (const unsigned short*)x, \
        (const unsigned short*)residual, (const unsigned short*)weight, (unsigned short*)out, (unsigned short*)residual_out, \
        (int)H, eps, (unsigned char*)q_out, scale_out, ps)
#define QL_RMS_NV(QDT) do { if (nv == 1) QL_RMS(1, QDT); else if (nv == 2) QL_RMS(2, QDT); else QL_RMS(4, QDT); } while (0)
    if (qdt < 0) QL_RMS_NV(-1); else if (qdt == QL_INT8) QL_RMS_NV(QL_INT8); else QL_RMS_NV(QL_FP8);
#undef QL_RMS_NV
#undef QL_RMS
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

int ql_add_rmsnorm(const void* x, const void* residual, const void* weight, float eps, void* out, void* residual_out,
                   int64_t T, int64_t H, void* stream) {
    return add_rmsnorm_launch("ql_add_rmsnorm", -1, x, residual, weight, eps, out, residual_out, nullptr, nullptr, T, H,
                              stream);
}

int ql_add_rmsnorm_quant(int dtype, const void* x, const void* residual, const void* weight, float eps, void* q_out,
                         float* scale_out, void* residual_out, int64_t T, int64_t H, void* stream) {
    return add_rmsnorm_launch("ql_add_rmsnorm_quant", dtype, x, residual, weight, eps, nullptr, residual_out, q_out,
                              scale_out, T, H, stream);
}

int ql_add_rmsnorm_partial(const float* partials, int slices, int granularity, const float* sw, const void* bias,
                           const void* residual, const void* weight, float eps, void* out, void* residual_out, int64_t T,
                           int64_t H, void* stream) {
    const char* fn = "ql_add_rmsnorm_partial";
    if (!partials || slices < 1 || granularity < QL_PER_ROW || granularity > QL_PER_GROUP ||
        (granularity != QL_PER_GROUP && !sw) || !residual || !residual_out) {
        set_error("%s: bad argument (a deferred GEMM output always joins a residual)", fn);
        return QL_EINVAL;
    }
    PartialSrc ps{};
    ps.part = partials; ps.S = slices; ps.slab = T * H;
    ps.sw = granularity == QL_PER_GROUP ? nullptr : sw;
    ps.sw_stride = granularity == QL_PER_ROW ? 1 : 0;
    ps.bias = (const unsigned short*)bias;
    ps.bias_after_round = granularity == QL_PER_ROW ? 0 : 1;
    return add_rmsnorm_launch(fn, -1, nullptr, residual, weight, eps, out, residual_out, nullptr, nullptr, T, H, stream, ps);
}

int ql_add_rmsnorm_partial_w8a8(int out_dtype, const void* partials, int slices, int acc_is_int, const float* sx,
                                const float* sw, const void* bias, const void* residual, const void* weight, float eps,
                                void* out, void* residual_out, void* q_out, float* scale_out, int64_t T, int64_t H,
                                void* stream) {
    const char* fn = "ql_add_rmsnorm_partial_w8a8";
    if (!partials || slices < 1 || !sx || !sw || !residual || !residual_out) {
        set_error("%s: bad argument (a deferred GEMM output always joins a residual)", fn);
        return QL_EINVAL;
    }
    PartialSrc ps{};
    ps.part = (const float*)partials; ps.S = slices; ps.slab = T * H;
    ps.sw = sw; ps.sw_stride = 1;
    ps.bias = (const unsigned short*)bias; ps.bias_after_round = 0;
    ps.sx = sx; ps.is_int = acc_is_int ? 1 : 0;
    return add_rmsnorm_launch(fn, out_dtype, nullptr, residual, weight, eps, out, residual_out, q_out, scale_out, T, H,
                              stream, ps);
}

int ql_silu_mul(const void* x, void* out, int64_t T, int64_t I, void* stream) {
    if (T < 0 || I <= 0 || (T > 0 && (!x || !out))) { set_error("ql_silu_mul: bad argument"); return QL_EINVAL; }
    if (I % 8) { set_error("ql_silu_mul: I must be a multiple of 8"); return QL_ESHAPE; }
    if (T == 0) return QL_OK;
    int64_t blocks = (T * (I / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const unsigned short*)x,
                       (unsigned short*)out, T, (int)I);
    QL_CHECK_LAUNCH("ql_silu_mul");
    return QL_OK;
}

int ql_qknorm_rope_store(const void* qkv, const int64_t* positions, const float* cos_sin, const void* q_norm_w,
                         const void* k_norm_w, float eps, void* q_out, void* k_out, void* v_out, void* k_cache,
                         void* v_cache, const int* slot_mapping, int64_t T, int Hq, int Hkv, int head_dim, void* stream) {
    if (T < 0 || Hq <= 0 || Hkv <= 0 || (T > 0 && (!qkv || !cos_sin || !q_out)) ||
        (slot_mapping && (!k_cache || !v_cache))) {
        set_error("ql_qknorm_rope_store: bad argument");
        return QL_EINVAL;
    }
    if (head_dim != 128) { set_error("ql_qknorm_rope_store: head_dim=%d (only 128 is built)", head_dim); return QL_ESHAPE; }
    if (T == 0) return QL_OK;
    const int64_t pairs = T * (Hq + 2 * Hkv);                 // 8 lanes each, 32 per 256-thread block
    const int64_t blocks = (pairs + 31) / 32;
    if (blocks > 0x7FFFFFFF) { set_error("ql_qknorm_rope_store: too many tokens"); return QL_ESHAPE; }
    RopeArgs r{positions, cos_sin, (const unsigned short*)q_norm_w, (const unsigned short*)k_norm_w, (unsigned short*)q_out,
               (unsigned short*)k_out, (unsigned short*)v_out, (unsigned short*)k_cache, (unsigned short*)v_cache, slot_mapping,
               eps, Hq, Hkv};
    hipLaunchKernelGGL(qknorm_rope_store_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       (const unsigned short*)qkv, (int)T, r);
    QL_CHECK_LAUNCH("ql_qknorm_rope_store");
    return QL_OK;
}

int ql_argmax_rows(const void* logits, int64_t* out, int64_t T, int64_t V, void* stream) {
    if (T < 0 || V <= 0 || (T > 0 && (!logits || !out))) { set_error("ql_argmax_rows: bad argument"); return QL_EINVAL; }
    if (V > 0x7FFFFFFF || T > 0x7FFFFFFF) { set_error("ql_argmax_rows: shape too large"); return QL_ESHAPE; }
    if (V % 8) { set_error("ql_argmax_rows: V=%lld must be a multiple of 8", (long long)V); return QL_ESHAPE; }
    if (T == 0) return QL_OK;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)T), dim3(1024), 0, as_stream(stream), (const unsigned short*)logits,
                       out, V);
    QL_CHECK_LAUNCH("ql_argmax_rows");
    return QL_OK;
}

size_t ql_paged_attn_workspace_bytes(int64_t B, int Hq, int Hkv, int block_size, int max_blocks) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || max_blocks <= 0 || block_size <= 0) return 0;
    const size_t chunks = ((size_t)max_blocks * block_size + kAttnChunk - 1) / kAttnChunk;
    if (chunks <= 1) return 0;
    return (size_t)B * Hkv * chunks * (Hq / Hkv) * 130 * 4;
}

static int paged_attn_decode(const void* q, const void* k_cache, const void* v_cache, const int* block_tables,
                             const int* context_lens, void* out, float scale, int64_t B, int Hq, int Hkv, int head_dim,
                             int block_size, int max_blocks, void* workspace, size_t workspace_bytes, void* stream,
                             float* rowmax_out, int qdt = -1, void* q_out = nullptr, float* scale_out = nullptr) {
    if (B < 0 || Hq <= 0 || Hkv <= 0 || block_size <= 0 || max_blocks <= 0 ||
        (B > 0 && (!q || !k_cache || !v_cache || !block_tables || !context_lens || !out))) {
        set_error("ql_paged_attn_decode: bad argument");
        return QL_EINVAL;
    }
    if (head_dim != 128 || Hq % Hkv) { set_error("ql_paged_attn_decode: head_dim must be 128 and Hq %% Hkv == 0"); return QL_ESHAPE; }
    // query heads per workgroup: the whole kv head's group where a kernel is built for it (1, 2, 4), else the
    // largest of those that divides it (8 -> two workgroups of 4 per kv head, 5 -> five of 1: K/V are then read
    // once per workgroup, i.e. the ratio's quotient times)
    const int kv_group = Hq / Hkv, kv_heads = Hkv;
    const int G = (kv_group % 4 == 0) ? 4 : (kv_group % 2 == 0) ? 2 : 1;
    if (G != kv_group && rowmax_out) {
        set_error("ql_paged_attn_decode_rowmax: Hq/Hkv=%d (row maxima per kv head need 1, 2 or 4)", kv_group);
        return QL_ESHAPE;
    }
    Hkv = Hq / G;                                    // from here on: groups of G query heads
    if (B == 0) return QL_OK;
    const size_t chunks = ((size_t)max_blocks * block_size + kAttnChunk - 1) / kAttnChunk;
    if (B > 0x7FFFFFFF || Hkv > 65535 || chunks > 65535) { set_error("ql_paged_attn_decode: batch too large"); return QL_ESHAPE; }
    const size_t need = ql_paged_attn_workspace_bytes(B, Hq, Hkv, block_size, max_blocks);
    if (need && (!workspace || workspace_bytes < need)) {
        set_error("ql_paged_attn_decode: workspace of %zu bytes needed, got %zu", need, workspace_bytes);
        return QL_EWORKSPACE;
    }
    dim3 grid((unsigned)B, (unsigned)Hkv, (unsigned)chunks), block(256);
    hipStream_t s = as_stream(stream);
#define QL_ATT(G_)                                                                                                  \
    do {                                                                                                            \
        hipLaunchKernelGGL(paged_attn_decode_kernel<G_>, grid, block, 0, s, (const unsigned short*)q,               \
                           (const unsigned short*)k_cache, (const unsigned short*)v_cache, block_tables, context_lens, \
                           (unsigned short*)out, scale, Hq, Hkv, block_size, max_blocks, (int)chunks, (float*)workspace, \
                           rowmax_out, kv_heads, kv_group);                                                         \
        if (qdt == QL_INT8)                                                                                         \
            hipLaunchKernelGGL((paged_attn_finalize_quant_kernel<G_, QL_INT8>), dim3((unsigned)B), block, 0, s,     \
                               context_lens, (const float*)workspace, (unsigned short*)out, (unsigned char*)q_out,  \
                               scale_out, Hq, Hkv, (int)chunks);                                                    \
        else if (qdt == QL_FP8)                                                                                     \
            hipLaunchKernelGGL((paged_attn_finalize_quant_kernel<G_, QL_FP8>), dim3((unsigned)B), block, 0, s,      \
                               context_lens, (const float*)workspace, (unsigned short*)out, (unsigned char*)q_out,  \
                               scale_out, Hq, Hkv, (int)chunks);                                                    \
        else if (chunks > 1)                                                                                        \
            hipLaunchKernelGGL(paged_attn_combine_kernel<G_>, dim3((unsigned)B, (unsigned)Hkv), block, 0, s,        \
                               context_lens, (const float*)workspace, (unsigned short*)out, Hq, Hkv, (int)chunks,   \
                               rowmax_out);                                                                         \
    } while (0)
    if (G == 1) QL_ATT(1); else if (G == 2) QL_ATT(2); else QL_ATT(4);
#undef QL_ATT
    QL_CHECK_LAUNCH("ql_paged_attn_decode");
    return QL_OK;
}

int ql_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache, const int* block_tables,
                         const int* context_lens, void* out, float scale, int64_t B, int Hq, int Hkv, int head_dim,
                         int block_size, int max_blocks, void* workspace, size_t workspace_bytes, void* stream) {
    return paged_attn_decode(q, k_cache, v_cache, block_tables, context_lens, out, scale, B, Hq, Hkv, head_dim, block_size,
                             max_blocks, workspace, workspace_bytes, stream, nullptr);
}

int ql_paged_attn_decode_quant(int dtype, const void* q, const void* k_cache, const void* v_cache, const int* block_tables,
                               const int* context_lens, void* out, void* q_out, float* scale_out, float scale, int64_t B,
                               int Hq, int Hkv, int head_dim, int block_size, int max_blocks, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if ((dtype != QL_INT8 && dtype != QL_FP8) || (B > 0 && (!q_out || !scale_out))) {
        set_error("ql_paged_attn_decode_quant: bad argument (dtype=%d)", dtype);
        return QL_EINVAL;
    }
    return paged_attn_decode(q, k_cache, v_cache, block_tables, context_lens, out, scale, B, Hq, Hkv, head_dim, block_size,
                             max_blocks, workspace, workspace_bytes, stream, nullptr, dtype, q_out, scale_out);
}

int ql_paged_attn_decode_rowmax(const void* q, const void* k_cache, const void* v_cache, const int* block_tables,
                                const int* context_lens, void* out, float* rowmax_part_out, float scale, int64_t B, int Hq,
                                int Hkv, int head_dim, int block_size, int max_blocks, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (B > 0 && !rowmax_part_out) { set_error("ql_paged_attn_decode_rowmax: bad argument"); return QL_EINVAL; }
    return paged_attn_decode(q, k_cache, v_cache, block_tables, context_lens, out, scale, B, Hq, Hkv, head_dim, block_size,
                             max_blocks, workspace, workspace_bytes, stream, rowmax_part_out);
}

}  // extern "C"
