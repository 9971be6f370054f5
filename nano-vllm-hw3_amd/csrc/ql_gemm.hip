// Fused dequant GEMM for gfx950:  y[M,N] = x[M,K] . dequant(qw[N,K])^T
//
// One kernel family covers decode (M <= 256, HBM-bound on the 1-byte weight stream) and prefill.
// Structure (per 256-thread workgroup = 4 waves, arranged WN x WK over N and K):
//   * every wave owns a strip of 32 output columns and ALL rows of the block's M tile
//     (MT fragments of 32 rows).  Its 1-byte weights go HBM -> VGPR directly (16 B per lane, no LDS:
//     a weight byte is read by exactly one wave), are widened int8/fp8 -> bf16 in registers ONCE
//     and reused for every row fragment, so the VALU cost of the dequant is amortised over MT MFMAs.
//   * the activation tile (shared by the waves) is staged through LDS in 16-B pieces, XOR-swizzled so
//     that the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots of the 256-B bank row.
//   * MFMA 32x32x16 (bf16 / fp8) or 32x32x32 (int8): A = activations (rows = m), B = weights
//     (cols = n), f32 / i32 accumulators; lane l holds column n = l&31 and rows
//     (r&3) + 8*(r>>2) + 4*(l>>5) of each 32x32 fragment.
//   * a lane's 16 consecutive k of a weight row (one 16-B load) feed two K=16 MFMAs (bytes 0-7, 8-15);
//     the activation fragment is read with the same k order, so the contraction is a permutation
//     of the k sum and no byte shuffling is needed.
//   * scales never touch the inner loop for per-row / per-tensor (applied to the f32 accumulator in
//     the epilogue); per-group keeps a second accumulator per fragment that is folded into the
//     running sum with its f32 group scale at each group boundary.
//   * K can be split across the block's waves (WK, reduced through LDS in a fixed order) and across
//     workgroups (blockIdx.z, f32/i32 slabs + a small reduce kernel, also fixed order): results are
//     bit-reproducible run to run.
#include "ql_common.h"
#include <type_traits>

namespace ql {

enum { ACT_BF16 = 0, ACT_I8 = 1, ACT_FP8 = 2 };

struct GemmParams {
    const unsigned char* x;      // [M,K] bf16 (ACT_BF16) or 1-byte (ACT_I8 / ACT_FP8)
    const unsigned char* w;      // [N,K] 1-byte
    const float* sw;             // [N] | [1] | [N, K/group]
    const float* sx;             // [M] for W8A8, else null
    const unsigned short* bias;  // [N] bf16 or null
    unsigned short* y;           // [M,N] bf16
    void* partial;               // [S][M][N] f32 / i32 when gridDim.z > 1
    int M, N, K;
    int group;                   // group size (per-group only)
    int sw_stride;               // 1 per-row, 0 per-tensor
    int k_per_split;             // multiple of the block's K chunk
    int bias_after_round;        // 1: bf16(bf16(v) + b)   0: bf16(v + b)
};

__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {
    // {hi[31:16], lo[31:16]}: exact for values representable in bf16
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}

// 8 one-byte weights (two dwords) -> 8 bf16 (exact: |int8| <= 128 and e4m3 both fit bf16's 8-bit significand)
template <bool FP8W>
__device__ __forceinline__ bf16x8 widen8(int d0, int d1) {
    unsigned o[4];
    if (FP8W) {
        auto a = __builtin_amdgcn_cvt_pk_f32_fp8(d0, false);
        auto b = __builtin_amdgcn_cvt_pk_f32_fp8(d0, true);
        auto c = __builtin_amdgcn_cvt_pk_f32_fp8(d1, false);
        auto d = __builtin_amdgcn_cvt_pk_f32_fp8(d1, true);
        o[0] = pack_hi16(a[0], a[1]); o[1] = pack_hi16(b[0], b[1]);
        o[2] = pack_hi16(c[0], c[1]); o[3] = pack_hi16(d[0], d[1]);
    } else {
        const int dd[2] = {d0, d1};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = dd[h];
            const float f0 = (float)(int)(signed char)(d);
            const float f1 = (float)(int)(signed char)(d >> 8);
            const float f2 = (float)(int)(signed char)(d >> 16);
            const float f3 = (float)(d >> 24);
            o[2 * h] = pack_hi16(f0, f1);
            o[2 * h + 1] = pack_hi16(f2, f3);
        }
    }
    i32x4 v = {(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int RB>
__device__ __forceinline__ int lds_off(int row, int ci) {
    constexpr int CPR = RB / 16;                       // 16-byte pieces per row
    constexpr int RPB = (RB >= 256) ? 1 : 256 / RB;    // rows per 256-byte bank row
    constexpr int MASK = (CPR >= 16) ? 15 : CPR - 1;
    const int f = (row / RPB) & MASK;
    return row * RB + (((ci & ~MASK) | ((ci ^ f) & MASK)) << 4);
}

template <int ACT, bool FP8W, int MT, int WN, int WK, int CP, bool GROUP>
struct Cfg {
    static constexpr int AB = (ACT == ACT_BF16) ? 2 : 1;          // activation bytes per element
    static constexpr int BM = 32 * MT;
    static constexpr int BN = 32 * WN;
    static constexpr int KW = 32 * CP;                            // k per wave per chunk
    static constexpr int KB = KW * WK;                            // k per block per chunk
    static constexpr int RB = KB * AB;                            // LDS row bytes
    static constexpr int CPR = RB / 16;
    static constexpr int NP = BM * CPR / 256;                     // 16-B pieces staged per thread
    static constexpr int XBYTES = BM * RB;
    static constexpr int RED = (WK > 1) ? (WK - 1) * WN * MT * 16 * 64 * 4 : 0;
    static constexpr int LDS = (2 * XBYTES > RED) ? 2 * XBYTES : RED;
    static_assert(WN * WK == 4, "4 waves per block");
    static_assert(NP >= 1 && (BM * CPR) % 256 == 0, "staging must tile the block");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int ACT, bool FP8W, int MT, int WN, int WK, int CP, bool GROUP>
__global__ __launch_bounds__(256) void qgemm_kernel(const GemmParams p) {
    using C = Cfg<ACT, FP8W, MT, WN, WK, CP, GROUP>;
    constexpr bool INT = (ACT == ACT_I8);
    using acc_t = typename std::conditional<INT, i32x16, f32x16>::type;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % WN;
    const int wk = wave / WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const int m0 = blockIdx.y * C::BM;
    const int n0 = blockIdx.x * C::BN + wn * 32;
    const int k0 = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, k0 + p.k_per_split);
    const int nchunks = (kend - k0 + C::KB - 1) / C::KB;

    // this lane's weight row (clamped: out-of-range columns compute garbage that is never stored)
    const int nrow = min(n0 + l31, p.N - 1);
    const unsigned char* wrow = p.w + (size_t)nrow * p.K + wk * C::KW + 16 * lh;

    acc_t acc[MT];
    acc_t accg[GROUP ? MT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    if (GROUP) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) accg[i][r] = 0;
    }

    i32x4 wcur[CP], wnext[CP], sreg[C::NP];

    auto load_w = [&](int kc, i32x4 (&dst)[CP]) {
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const int kb = kc + wk * C::KW + c * 32;
            if (kb < kend) dst[c] = *(const i32x4*)(wrow + kc + c * 32);
            else dst[c] = i32x4{0, 0, 0, 0};
        }
    };
    auto load_x = [&](int kc) {
#pragma unroll
        for (int j = 0; j < C::NP; ++j) {
            const int pidx = tid + 256 * j;
            const int row = pidx / C::CPR, ci = pidx % C::CPR;
            const int gm = m0 + row;
            const int k = kc + ci * (16 / C::AB);
            if (gm < p.M && k < kend)
                sreg[j] = *(const i32x4*)(p.x + ((size_t)gm * p.K + k) * C::AB);
            else
                sreg[j] = i32x4{0, 0, 0, 0};
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int j = 0; j < C::NP; ++j) {
            const int pidx = tid + 256 * j;
            const int row = pidx / C::CPR, ci = pidx % C::CPR;
            *(i32x4*)(smem + buf * C::XBYTES + lds_off<C::RB>(row, ci)) = sreg[j];
        }
    };
    auto flush_group = [&](int kb) {
        // fold the finished group's partial sums into the running sum with the group's f32 scale
        const float s = p.sw[(size_t)nrow * (p.K / p.group) + kb / p.group];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][r] = __builtin_fmaf((float)accg[i][r], s, (float)acc[i][r]);
                accg[i][r] = 0;
            }
    };
    auto compute = [&](int buf, int kc, const i32x4 (&wr)[CP]) {
        const unsigned char* xb = smem + buf * C::XBYTES;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const int kb = kc + wk * C::KW + c * 32;   // first k of this pair-step (wave-uniform)
            if (kb < kend) {
                if (ACT == ACT_BF16) {
                    const bf16x8 blo = widen8<FP8W>(wr[c][0], wr[c][1]);
                    const bf16x8 bhi = widen8<FP8W>(wr[c][2], wr[c][3]);
                    const int ci = (wk * CP + c) * 4 + 2 * lh;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int row = 32 * i + l31;
                        const bf16x8 alo = *(const bf16x8*)(xb + lds_off<C::RB>(row, ci));
                        const bf16x8 ahi = *(const bf16x8*)(xb + lds_off<C::RB>(row, ci + 1));
                        if (GROUP) {
                            accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, blo, accg[i], 0, 0, 0);
                            accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, accg[i], 0, 0, 0);
                        } else {
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, blo, acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[i], 0, 0, 0);
                        }
                    }
                } else {
                    const int ci = (wk * CP + c) * 2 + lh;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int row = 32 * i + l31;
                        const i32x4 a = *(const i32x4*)(xb + lds_off<C::RB>(row, ci));
                        if constexpr (ACT == ACT_I8) {
                            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, wr[c], acc[i], 0, 0, 0);
                        } else {
                            const long a0 = ((long)(unsigned)a[1] << 32) | (unsigned)a[0];
                            const long a1 = ((long)(unsigned)a[3] << 32) | (unsigned)a[2];
                            const long b0 = ((long)(unsigned)wr[c][1] << 32) | (unsigned)wr[c][0];
                            const long b1 = ((long)(unsigned)wr[c][3] << 32) | (unsigned)wr[c][2];
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, b0, acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, acc[i], 0, 0, 0);
                        }
                    }
                }
                if (GROUP) {
                    // flush at every group end, and at the end of a wave chunk (the next chunk of this
                    // wave may belong to another group, or continue this one: both are handled by
                    // scaling what has been summed so far with the current group's scale)
                    const int gq = (p.group < C::KW) ? p.group : C::KW;
                    if (((kb + 32) % gq) == 0 || kb + 32 >= kend) flush_group(kb);
                }
            }
        }
    };

    // ---- main loop: LDS double buffer for x, register double buffer for the weights ----------
    if (nchunks > 0) {
        load_w(k0, wcur);
        load_x(k0);
        store_x(0);
    }
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int kc = k0 + it * C::KB;
        const int buf = it & 1;
        const bool more = (it + 1) < nchunks;
        if (more) {
            load_w(kc + C::KB, wnext);
            load_x(kc + C::KB);
        }
        compute(buf, kc, wcur);
        if (more) {
            store_x(buf ^ 1);
#pragma unroll
            for (int c = 0; c < CP; ++c) wcur[c] = wnext[c];
        }
        __syncthreads();
    }

    // ---- reduce the WK waves of a column strip through LDS (fixed order) ----------------------
    if (WK > 1) {
        using raw_t = typename std::conditional<INT, int, float>::type;
        raw_t* red = (raw_t*)smem;
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[((((wk - 1) * WN + wn) * MT + i) * 16 + r) * 64 + lane] = acc[i][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int j = 1; j < WK; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][r] += red[((((j - 1) * WN + wn) * MT + i) * 16 + r) * 64 + lane];
        }
    }
    if (wk != 0) return;

    // ---- epilogue ------------------------------------------------------------------------------
    const int n = n0 + l31;
    if (n >= p.N) return;
    const bool split = gridDim.z > 1;
    float sw = 1.f, bias = 0.f;
    if (!split) {
        if (!GROUP) sw = p.sw[(size_t)n * p.sw_stride];
        if (p.bias) bias = bf16_bits_to_f32(p.bias[n]);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < p.M) {
                if (split) {
                    using raw_t = typename std::conditional<INT, int, float>::type;
                    ((raw_t*)p.partial)[((size_t)blockIdx.z * p.M + m) * p.N + n] = acc[i][r];
                } else {
                    float v = (float)acc[i][r];
                    if (ACT != ACT_BF16) v = __fmul_rn(v, p.sx[m]);
                    v = __fmul_rn(v, sw);
                    if (p.bias) {
                        if (p.bias_after_round) v = __fadd_rn(round_to_bf16(v), bias);
                        else v = __fadd_rn(v, bias);
                    }
                    p.y[(size_t)m * p.N + n] = f32_to_bf16_bits(v);
                }
            }
        }
    }
}

// Sum the S split-K slabs in slab order and apply the same epilogue as the fused path.
template <bool INT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, int S, int has_sx, int group_mode) {
    const size_t total = (size_t)p.M * p.N;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int m = (int)(idx / p.N), n = (int)(idx % p.N);
    float v;
    if (INT) {
        int a = 0;
        for (int s = 0; s < S; ++s) a += ((const int*)p.partial)[(size_t)s * total + idx];
        v = (float)a;
    } else {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a = __fadd_rn(a, ((const float*)p.partial)[(size_t)s * total + idx]);
        v = a;
    }
    if (has_sx) v = __fmul_rn(v, p.sx[m]);
    if (!group_mode) v = __fmul_rn(v, p.sw[(size_t)n * p.sw_stride]);
    if (p.bias) {
        const float b = bf16_bits_to_f32(p.bias[n]);
        if (p.bias_after_round) v = __fadd_rn(round_to_bf16(v), b);
        else v = __fadd_rn(v, b);
    }
    p.y[idx] = f32_to_bf16_bits(v);
}

// ---- host-side dispatch ------------------------------------------------------------------------
struct TileCfg { int mt, wn, wk, cp; };
static const TileCfg kTiles[] = {
    {1, 1, 4, 2}, {2, 1, 4, 2}, {1, 4, 1, 4}, {2, 4, 1, 4}, {4, 4, 1, 4}, {8, 4, 1, 4}, {2, 2, 2, 2}, {4, 2, 2, 2},
};
static thread_local int f_mt = 0, f_wn = 0, f_wk = 0, f_split = 0;

typedef void (*kern_t)(const GemmParams);

template <int ACT, bool FP8W, bool GROUP>
static kern_t pick_kernel(int mt, int wn, int wk) {
#define QL_TILE(MT_, WN_, WK_, CP_) \
    if (mt == MT_ && wn == WN_ && wk == WK_) return (kern_t)qgemm_kernel<ACT, FP8W, MT_, WN_, WK_, CP_, GROUP>;
    QL_TILE(1, 1, 4, 2)
    QL_TILE(2, 1, 4, 2)
    QL_TILE(1, 4, 1, 4)
    QL_TILE(2, 4, 1, 4)
    QL_TILE(4, 4, 1, 4)
    QL_TILE(2, 2, 2, 2)
    QL_TILE(4, 2, 2, 2)
    if constexpr (!GROUP) { QL_TILE(8, 4, 1, 4) }
#undef QL_TILE
    return nullptr;
}

static bool tile_built(int mt, int wn, int wk, bool group) {
    for (const TileCfg& t : kTiles)
        if (t.mt == mt && t.wn == wn && t.wk == wk) return !(group && mt == 8);
    return false;
}

static int tile_cp(int mt, int wn, int wk) {
    for (const TileCfg& t : kTiles)
        if (t.mt == mt && t.wn == wn && t.wk == wk) return t.cp;
    return 4;
}

struct Plan { int mt, wn, wk, cp, split, k_per_split; dim3 grid; };

static Plan make_plan(int64_t M, int64_t N, int64_t K, bool group) {
    Plan pl;
    // Decode-sized problems (M <= 256) are latency/L2-bound, not MFMA-bound, on a 256-CU chip: the
    // measured optimum is many small workgroups (32 rows x 32 columns, the four waves splitting K) at
    // high occupancy.  Prefill-sized problems use the tallest tile so that each weight byte is
    // converted once per 256 rows.
    int mt, wn, wk;
    if (M <= 256) { mt = 1; wn = 1; wk = 4; }
    else { mt = group ? 4 : 8; wn = 4; wk = 1; }
    if (f_mt > 0) mt = f_mt;
    if (f_wn > 0 && f_wk > 0) { wn = f_wn; wk = f_wk; }
    if (!tile_built(mt, wn, wk, group)) { mt = group && mt > 4 ? 4 : mt; wn = 4; wk = 1; }
    pl.mt = mt; pl.wn = wn; pl.wk = wk; pl.cp = tile_cp(mt, wn, wk);
    const int kb = 32 * pl.cp * wk;
    const int64_t mblocks = (M + 32 * mt - 1) / (32 * mt);
    const int64_t nblocks = (N + 32 * wn - 1) / (32 * wn);
    const int64_t kchunks = (K + kb - 1) / kb;
    // split K across workgroups only when the grid would leave most CUs idle, and keep >= 512 k per
    // slice: below that the extra reduce pass costs more than the parallelism buys
    int64_t split = 1;
    const int64_t base = mblocks * nblocks;
    const int64_t max_split = K / 512 > 0 ? K / 512 : 1;
    if (base < 128) split = (256 + base - 1) / base;
    else if (base < 512 && K >= 2048) split = 2;
    if (split > max_split) split = max_split;
    if (f_split > 0) split = f_split;
    if (split > 64) split = 64;
    if (split > kchunks) split = kchunks;
    if (split < 1) split = 1;
    const int64_t cps = (kchunks + split - 1) / split;      // chunks per slice
    split = (kchunks + cps - 1) / cps;
    pl.split = (int)split;
    pl.k_per_split = (int)(cps * kb);
    pl.grid = dim3((unsigned)nblocks, (unsigned)mblocks, (unsigned)split);
    return pl;
}

static int check_common(const char* fn, int64_t M, int64_t N, int64_t K) {
    if (M < 0 || N <= 0 || K <= 0) { set_error("%s: bad shape M=%lld N=%lld K=%lld", fn, (long long)M, (long long)N, (long long)K); return QL_EINVAL; }
    if (K % 32) { set_error("%s: K=%lld must be a multiple of 32", fn, (long long)K); return QL_ESHAPE; }
    if (M > 0x3FFFFFFF || N > 0x3FFFFFFF || K > 0x3FFFFFFF || (M + 255) / 32 > 65535) {
        set_error("%s: shape too large", fn);
        return QL_ESHAPE;
    }
    return QL_OK;
}

static int run(kern_t k, bool is_int, GemmParams& p, const Plan& pl, bool has_sx, bool group, void* ws, size_t ws_bytes,
               hipStream_t s, const char* fn) {
    if (!k) { set_error("%s: tile (%d,%d,%d) not built", fn, pl.mt, pl.wn, pl.wk); return QL_EINVAL; }
    p.k_per_split = pl.k_per_split;
    if (pl.split > 1) {
        const size_t need = (size_t)pl.split * p.M * p.N * 4;
        if (!ws || ws_bytes < need) {
            set_error("%s: workspace of %zu bytes needed for split-K=%d, got %zu", fn, need, pl.split, ws_bytes);
            return QL_EWORKSPACE;
        }
        p.partial = ws;
    }
    hipLaunchKernelGGL(k, pl.grid, dim3(256), 0, s, p);
    if (pl.split > 1) {
        const size_t total = (size_t)p.M * p.N;
        dim3 g((unsigned)((total + 255) / 256));
        if (is_int) hipLaunchKernelGGL(splitk_reduce_kernel<true>, g, dim3(256), 0, s, p, pl.split, has_sx ? 1 : 0, group ? 1 : 0);
        else hipLaunchKernelGGL(splitk_reduce_kernel<false>, g, dim3(256), 0, s, p, pl.split, has_sx ? 1 : 0, group ? 1 : 0);
    }
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

}  // namespace ql

using namespace ql;

extern "C" {

size_t ql_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int granularity) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32) return 0;
    const Plan pl = make_plan(M, N, K, granularity == QL_PER_GROUP);
    return pl.split > 1 ? (size_t)pl.split * (size_t)M * (size_t)N * 4 : 0;
}

int ql_gemm_force_config(int mt, int wn, int wk, int split_k) {
    if (mt > 0 && wn > 0 && wk > 0 && !tile_built(mt, wn, wk, false)) {
        set_error("ql_gemm_force_config: tile (%d,%d,%d) not built", mt, wn, wk);
        return QL_EINVAL;
    }
    f_mt = mt > 0 ? mt : 0;
    f_wn = wn > 0 ? wn : 0;
    f_wk = wk > 0 ? wk : 0;
    f_split = split_k > 0 ? split_k : 0;
    return QL_OK;
}

int ql_gemm_w8a16(int wdtype, int granularity, const void* x, const void* qw, const float* sw, const void* bias,
                  void* y, int64_t M, int64_t N, int64_t K, int group, void* workspace, size_t workspace_bytes,
                  void* stream) {
    const char* fn = "ql_gemm_w8a16";
    if ((wdtype != QL_INT8 && wdtype != QL_FP8) || granularity < QL_PER_ROW || granularity > QL_PER_GROUP ||
        (M > 0 && (!x || !y)) || !qw || !sw) {
        set_error("%s: bad argument (wdtype=%d granularity=%d)", fn, wdtype, granularity);
        return QL_EINVAL;
    }
    int rc = check_common(fn, M, N, K);
    if (rc) return rc;
    const bool grp = granularity == QL_PER_GROUP;
    if (grp) {
        const bool ok = group >= 32 && K % group == 0 && ((group % 128) == 0 || (128 % group) == 0);
        if (!ok) { set_error("%s: group=%d unsupported for K=%lld", fn, group, (long long)K); return QL_ESHAPE; }
    }
    if (M == 0) return QL_OK;
    GemmParams p{};
    p.x = (const unsigned char*)x; p.w = (const unsigned char*)qw; p.sw = sw; p.sx = nullptr;
    p.bias = (const unsigned short*)bias; p.y = (unsigned short*)y; p.partial = nullptr;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group = grp ? group : 0;
    p.sw_stride = granularity == QL_PER_ROW ? 1 : 0;
    p.bias_after_round = granularity == QL_PER_ROW ? 0 : 1;
    const Plan pl = make_plan(M, N, K, grp);
    kern_t k;
    if (wdtype == QL_INT8) k = grp ? pick_kernel<ACT_BF16, false, true>(pl.mt, pl.wn, pl.wk) : pick_kernel<ACT_BF16, false, false>(pl.mt, pl.wn, pl.wk);
    else k = grp ? pick_kernel<ACT_BF16, true, true>(pl.mt, pl.wn, pl.wk) : pick_kernel<ACT_BF16, true, false>(pl.mt, pl.wn, pl.wk);
    return run(k, false, p, pl, false, grp, workspace, workspace_bytes, as_stream(stream), fn);
}

int ql_gemm_w8a8(int dtype, const void* qx, const float* sx, const void* qw, const float* sw, const void* bias,
                 void* y, int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "ql_gemm_w8a8";
    if ((dtype != QL_INT8 && dtype != QL_FP8) || (M > 0 && (!qx || !sx || !y)) || !qw || !sw) {
        set_error("%s: bad argument (dtype=%d)", fn, dtype);
        return QL_EINVAL;
    }
    int rc = check_common(fn, M, N, K);
    if (rc) return rc;
    if (M == 0) return QL_OK;
    GemmParams p{};
    p.x = (const unsigned char*)qx; p.w = (const unsigned char*)qw; p.sw = sw; p.sx = sx;
    p.bias = (const unsigned short*)bias; p.y = (unsigned short*)y; p.partial = nullptr;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group = 0; p.sw_stride = 1; p.bias_after_round = 0;
    const Plan pl = make_plan(M, N, K, false);
    kern_t k = dtype == QL_INT8 ? pick_kernel<ACT_I8, false, false>(pl.mt, pl.wn, pl.wk)
                                : pick_kernel<ACT_FP8, true, false>(pl.mt, pl.wn, pl.wk);
    return run(k, dtype == QL_INT8, p, pl, true, false, workspace, workspace_bytes, as_stream(stream), fn);
}

}  // extern "C"
