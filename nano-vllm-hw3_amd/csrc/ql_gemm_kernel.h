// Fused dequant GEMM for gfx950:  y[M,N] = x[M,K] . dequant(qw[N,K])^T     (kernel templates)
//
// One kernel family covers decode (M <= 256, latency/HBM-bound on the 1-byte weight stream) and prefill.
// Structure (per 256-thread workgroup = 4 waves, arranged WN x WK over N and K):
//   (workgroups of 8 waves, 512 threads, exist too: WN x WK = 8)
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
//   * a K chunk of CP x 32 k per wave is loaded at once (CP x 16-B weight loads per lane in flight
//     before the first wait); DB = LDS double buffer (next chunk's loads overlap this chunk's MFMAs),
//     !DB = single buffer for problems whose K slice fits one chunk (everything in flight at once).
//   * K can be split across the block's waves (WK, reduced through LDS in a fixed order) and across
//     workgroups (blockIdx.z): each slice stores its f32/i32 partial tile to a slab, takes a ticket
//     on a per-tile counter, and the LAST arriver sums the slabs in slice order and runs the epilogue
//     (agent-scope release before the ticket, agent-scope acquire after it; no spinning, so no
//     co-residency requirement).  Results are bit-reproducible run to run.
#pragma once
#include "ql_common.h"
#include "ql_rope.h"
#include <type_traits>

namespace ql {

enum { ACT_BF16 = 0, ACT_I8 = 1, ACT_FP8 = 2 };
#ifndef QL_FOLD_PK
#define QL_FOLD_PK 1          // per-group fold on register pairs (v_pk_fma_f32); 0: one v_fma_f32 per register
#endif
#ifndef QL_FP8_K64
#define QL_FP8_K64 1          // fp8 x fp8 on the K = 64 block-scaled instruction (0: the K = 16 instruction pair)
#endif

constexpr int kCounterBytes = 64 * 1024;   // head of the workspace: one u32 ticket counter per output tile
constexpr int kMaxTiles = kCounterBytes / 4;

struct GemmParams {
    const unsigned char* x;      // [M,K] bf16 (ACT_BF16) or 1-byte (ACT_I8 / ACT_FP8)
    const unsigned char* w;      // [N,K] 1-byte (W16 kernels: [N,K] bf16, row-major, no scales)
    const float* sw;             // [N] | [1] | [N, K/group] (W16 kernels: null)
    const float* sx;             // [M] for W8A8, else null
    const unsigned short* bias;  // [N] bf16 or null
    unsigned short* y;           // [M,N] bf16
    unsigned* counters;          // [tiles] zero before the call, zero again after it (split-K only)
    void* partial;               // [tiles][S][WN][MT][16][64] f32 / i32 slabs (split-K only)
    float* partial_rm;           // deferred reduction: [S][M][N] f32 (int32 for int8 x int8), slice z stores its unscaled partial
                                 // sums row-major and the NEXT launch (ql_add_rmsnorm_partial) adds the slices up
    int M, N, K;
    int group;                   // group size (per-group only)
    int group_generic;           // 1: never take the group-aligned form of the main loop (A/B checks; same bits either way)
    unsigned gmagic;             // ceil(2^32 / (group / 32)): k / group = umulhi(k >> 5, gmagic) for every k < 2^31 the
                                 // kernels see (0 when group / 32 is 1: the quotient is k >> 5 itself)
    int sw_stride;               // 1 per-row, 0 per-tensor
    int k_per_split;             // multiple of the block's K chunk
    int bias_after_round;        // 1: bf16(bf16(v) + b)   0: bf16(v + b)
    int silu;                    // 1: weight rows alternate 32-row strips of gate and up; y[M, N/2] = silu(gate) * up
    int swz_gn;                  // 0, or the number of XCD groups along N (1, 2, 4, 8) of the XCD-aware tile order,
                                 // or -R: a 1-D grid in which the R row tiles of a column tile are 8 workgroup ids apart
    int ntile_n;                 // column tiles (the grid's x extent unless swz_gn < 0)
    int wtiled;                  // 1: w is stored as [N/32 strips][K/32 steps][64 lanes][16 bytes] (see ql_tile_weight)
    float* rowmax_out;           // SiLU epilogue only, may be null: [M][nparts_out] f32, |max| of every output row over each
    int nparts_out;              // 32-column strip of y (what a DYNQ consumer of y takes as rowmax_part)
    const float* rowmax_part;    // DYNQ kernels (W8A8 with the activation quantization in the staging): x is bf16 [M,K],
    int nparts;                  // rowmax_part [M][nparts] holds partial |max| of every row (any split of its columns)
    int2* amax;                  // AMAX kernels (output projection with the greedy sampler in the epilogue): y is not
                                 // stored; [M][gridDim.x] pairs {f32 bits of the row's maximum over the tile's columns as
                                 // the bf16 the plain kernel would store, its smallest column}, reduced by argmax_finish
    int rope_on;                 // 1 on the ROPE kernels (qkv projection: y is not stored; q/k norm + RoPE + KV store in the epilogue)
    RopeArgs rope;
    // Stream-K schedule (SK kernels): a 1-D grid of sk_G workgroups shares the (output tile, k chunk) units evenly,
    // each workgroup taking one contiguous range of them (see the kernel).  ntile_n = column tiles.
    int sk_G;                    // workgroups of the grid
    int sk_C;                    // k chunks (of the tile's KB) per output tile
    int sk_R;                    // row tiles that walk the same (column tile, chunk) range side by side on one XCD: all
                                 // of the problem's row tiles, or 1 (row tiles are then units like any other)
    int sk_mblocks;              // row tiles of the problem
    // the schedule's divisions as multiply-high (an integer division is ~30 dependent instructions, and the
    // schedule needs ten of them before a workgroup's first load): umulhi(n, ceil(2^32 / d)) == n / d while n d < 2^32
    // (0 stands for d = 1).  per = units per workgroup (the first `rem` workgroups of a unit range take one more).
    int sk_Gr, sk_per, sk_rem;
    unsigned sk_mC, sk_mper, sk_mper1, sk_mmb, sk_mR;
};

typedef void (*kern_t)(const GemmParams);

__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {
    // {hi[31:16], lo[31:16]}: exact for values representable in bf16
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}

// 8 one-byte weights (two dwords) -> 8 bf16 (exact: |int8| <= 128 and e4m3 both fit bf16's 8-bit significand)
template <bool FP8W>
__device__ __forceinline__ bf16x8 widen8(int d0, int d1) {
    unsigned o[4];
    if (FP8W) {
        // v_cvt_scalef32_pk_bf16_fp8: two e4m3 bytes -> two packed bf16 in ONE instruction (scale 1.0, exact)
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        o[0] = __builtin_bit_cast(unsigned, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, false));
        o[1] = __builtin_bit_cast(unsigned, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, true));
        o[2] = __builtin_bit_cast(unsigned, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, false));
        o[3] = __builtin_bit_cast(unsigned, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, true));
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

// what one 32-k step of a lane's weight column occupies: 16 one-byte weights, or (W16) 16 bf16 in two registers sets
struct W2 { i32x4 a, b; };

// 16-byte load through an explicitly GLOBAL pointer: pointers that arrive inside the by-value parameter struct
// are generic to the compiler, and a generic (flat) load also counts against lgkmcnt, so every LDS wait in the
// main loop would wait for the weight stream as well
__device__ __forceinline__ i32x4 gload16(const void* ptr) {
    return *(const __attribute__((address_space(1))) i32x4*)ptr;
}

template <int RB>
__device__ __forceinline__ int lds_off(int row, int ci) {
    constexpr int CPR = RB / 16;                       // 16-byte pieces per row
    constexpr int RPB = (RB >= 256) ? 1 : 256 / RB;    // rows per 256-byte bank row
    constexpr int MASK = (CPR >= 16) ? 15 : CPR - 1;
    const int f = (row / RPB) & MASK;
    return row * RB + (((ci & ~MASK) | ((ci ^ f) & MASK)) << 4);
}

template <int ACT, int MT, int WN, int WK, int CP, bool DB>
struct Cfg {
    static constexpr int AB = (ACT == ACT_BF16) ? 2 : 1;          // activation bytes per element
    static constexpr int BM = 32 * MT;
    static constexpr int BN = 32 * WN;
    static constexpr int KW = 32 * CP;                            // k per wave per chunk
    static constexpr int KB = KW * WK;                            // k per block per chunk
    static constexpr int RB = KB * AB;                            // LDS row bytes
    static constexpr int CPR = RB / 16;
    static constexpr int NW = WN * WK;                            // waves per workgroup (4 or 8)
    static constexpr int NT = 64 * NW;                            // threads per workgroup
    static constexpr int NP = BM * CPR / NT;                      // 16-B pieces staged per thread
    static constexpr int XBYTES = BM * RB;
    static constexpr int RED = (WK > 1) ? WK * WN * MT * 16 * 64 * 4 : 0;       // K reduction across the block's waves
    static constexpr int XALL = (DB ? 2 : 1) * XBYTES;
    static constexpr int PAIR = (WN >= 2) ? WK * (WN / 2) * MT * 16 * 64 * 2 : 0;   // gate/up exchange of the SiLU epilogue
    static constexpr int LDS0 = (XALL > RED) ? XALL : RED;
    static constexpr int LDS1 = (LDS0 > PAIR) ? LDS0 : PAIR;
    // qkv epilogue (tiles whose 128 or 256 columns are whole heads): the bf16 output tile, rows padded by 16 bytes
    static constexpr int ROPE_RS = BN + 8;                        // elements per row of that image
    static constexpr int ROPE = BM * ROPE_RS * 2;
    static constexpr int LDS = LDS1;
    static constexpr int LDS_ROPE = (LDS1 > ROPE) ? LDS1 : ROPE;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
    static_assert(NP >= 1 && (BM * CPR) % NT == 0, "staging must tile the block");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int ACT, bool FP8W, int MT, int WN, int WK, int CP, bool DB, bool GROUP, bool ROPE = false, bool W16 = false,
          bool AMAX = false, bool DYNQ = false, bool SK = false>
// (second launch bound = waves per SIMD the register allocation must leave room for: the 128 x 256 weight-only tile
// with 64-k chunks is held to 128 registers so that TWO of its 8-wave workgroups share a CU and one's prologue and
// epilogue run under the other's MFMAs - worth more on the K <= 2048 shapes than the registers it gives up)
// (likewise the 64 x 128 tile on the stream-K schedule, whose segment loop costs it a dozen registers)
__global__ __launch_bounds__(64 * WN * WK, ((ACT == ACT_BF16 && MT == 4 && WN == 8 && WK == 1 && CP == 2 && !GROUP && !W16) ||
                                            (SK && ACT == ACT_BF16 && MT == 2 && WN == 4 && WK == 2 && CP == 4 && !GROUP)) ? 4 : 1)
void qgemm_kernel(const GemmParams p) {
    static_assert(!ROPE || WN == 4 || WN == 8, "the qkv epilogue needs whole 128-column heads per workgroup");
    static_assert(!W16 || (ACT == ACT_BF16 && !GROUP && !ROPE && !FP8W), "16-bit weights: plain bf16 x bf16");
    static_assert(!AMAX || (W16 && WK == 1 && WN >= 4), "the sampler epilogue is built for the output projection's tiles");
    static_assert(!DYNQ || (ACT != ACT_BF16 && !ROPE && !W16 && !GROUP), "dynamic activation quantization: the W8A8 kernels");
    static_assert(!SK || (!W16 && !AMAX && !DYNQ), "the stream-K schedule is built for the quantized-weight GEMMs");
    using wreg_t = typename std::conditional<W16, W2, i32x4>::type;
    using xreg_t = typename std::conditional<DYNQ, W2, i32x4>::type;   // one staged piece: 16 activations as they lie in memory
    using C = Cfg<ACT, MT, WN, WK, CP, DB>;
    constexpr bool INT = (ACT == ACT_I8);
    constexpr int RB = C::RB;
    constexpr int SWZ_RPB = (RB >= 256) ? 1 : 256 / RB;
    constexpr int SWZ_MASK = (C::CPR >= 16) ? 15 : C::CPR - 1;
    constexpr int APS = (ACT == ACT_BF16) ? 4 : 2;               // 16-B activation pieces per 32-k step and row
    using acc_t = typename std::conditional<INT, i32x16, f32x16>::type;
    using raw_t = typename std::conditional<INT, int, float>::type;
    using TrueT = std::integral_constant<bool, true>;
    using FalseT = std::integral_constant<bool, false>;
    constexpr int LDS_MAIN = (ROPE || AMAX) ? C::LDS_ROPE : C::LDS;
    constexpr int DQ_OFF = (LDS_MAIN + 15) & ~15;                // DYNQ: per-row {QMAX / max, max / QMAX} behind everything else
    // Per-group scales staged in LDS on the single-buffer tiles: the scales of a chunk's groups, [group in chunk]
    // [tile column] f32, travel with the chunk's activation tile - requested with it, stored with it - so that folding
    // a finished group costs an LDS read instead of a global load in the middle of the weight stream.  Those tiles keep
    // a ring of up to eight weight loads in flight per lane, and the wait counter is positional: waiting for a scale
    // requested at the fold also waits for every weight load requested before it, i.e. drains the ring at every group
    // (measured on the Qwen3-4B shapes: -4..9 % with the scales staged).  The double-buffered tiles request a chunk's
    // weights a whole chunk ahead, so the fold's load waits for nothing that has not landed; staged scales cost them
    // 10 % (more LDS traffic between the fragment reads) and they keep the load.
    constexpr bool SLDS = GROUP && !DB;
    constexpr int SPC = C::KB / 32;                              // groups per chunk at most (group >= 32)
    constexpr int SC_BYTES = SLDS ? C::BN * SPC * 4 : 0;         // per buffer
    constexpr int SC_OFF = (LDS_MAIN + 15) & ~15;
    constexpr int NSL = SLDS ? (C::BN * SPC + C::NT - 1) / C::NT : 1;   // scale loads per thread and chunk at most
    static_assert(!SLDS || SC_OFF + (DB ? 2 : 1) * SC_BYTES <= 160 * 1024, "LDS budget with the staged group scales");
    __shared__ __attribute__((aligned(128))) unsigned char smem[DYNQ ? DQ_OFF + 8 * C::BM : SLDS ? SC_OFF + (DB ? 2 : 1) * SC_BYTES : LDS_MAIN];   // base must be a multiple of 32 (address ^ 16 below)

    // The kernel is VALU-issue-bound at decode sizes, so everything that is uniform per wave is kept
    // in scalar registers (the wave index is made provably uniform) and every per-lane address is
    // split into a lane-constant part computed once and a uniform part.
    // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2), so with the
    // plain (x = N tile, y = M tile) order every XCD touches every activation tile.  For prefill-sized grids the
    // host asks for an XCD-aware order instead: the XCDs form a (8/gn) x gn grid over (M tiles, N tiles); one XCD
    // walks its N tiles fastest, so concurrent workgroups share an activation tile in L2 and the XCD's slice of
    // the weights (N*K/gn bytes) stays L2-resident.
    unsigned bx = blockIdx.x, by = blockIdx.y;
    // Stream-K (SK kernels).  A grid of p.sk_G workgroups, about one or two per CU, shares the problem's
    // (output tile, k chunk) units evenly: workgroup g takes U / G (+ 1) consecutive units of the tile-major order,
    // i.e. the end of one tile's K range, whole tiles, and the start of another's.  Every CU is busy for the same
    // time whatever the tile count (152 tiles of 128 x 256 on 256 CUs otherwise leave 104 of them idle), and a tile
    // that several workgroups share is combined exactly like a split-K tile: slabs, a ticket per tile, the last
    // arriver sums the slabs in ascending k order.  With sk_R row tiles the workgroups (w >> 3) % R = 0..R-1 of one
    // XCD walk the same (column tile, chunk) range on the R row tiles at the same time, so the weights they share
    // come from HBM once and from that XCD's L2 afterwards.
    int sk_u = 0, sk_uend = 0, sk_g = 0, sk_lane = 0, sk_Tl = 0;
    auto mdiv = [](unsigned n, unsigned magic) -> unsigned { return magic ? __umulhi(n, magic) : n; };
    if constexpr (SK) {
        const unsigned w = blockIdx.x, R = (unsigned)p.sk_R;
        if (R > 1) {
            const unsigned q8 = w >> 3, qq = mdiv(q8, p.sk_mR);
            sk_lane = (int)(q8 - qq * R);
            sk_g = (int)((w & 7u) + 8u * qq);
        } else {
            sk_g = (int)w;
        }
        sk_Tl = R > 1 ? p.ntile_n : p.ntile_n * p.sk_mblocks;
        // unit range of workgroup g: `per` units, one more for the first `rem` workgroups
        sk_u = sk_g * p.sk_per + min(sk_g, p.sk_rem);
        sk_uend = sk_u + p.sk_per + (sk_g < p.sk_rem ? 1 : 0);
    }
    if (SK) {
    } else if (p.swz_gn < 0) {
        // Few row tiles over a long weight stream (the output projection at 129+ rows): workgroup ids are dealt
        // round-robin over the XCDs, so ids 8 apart run on the same XCD at about the same time; giving them the
        // same column tile makes the second row tile's weight reads L2 hits instead of a second pass over HBM.
        const unsigned R = (unsigned)-p.swz_gn, lin = blockIdx.x;
        bx = (lin / (8u * R)) * 8u + (lin & 7u);
        by = (lin >> 3) % R;
        if (bx >= (unsigned)p.ntile_n) return;               // (the grid is padded to whole groups of 8 R)
    } else if (p.swz_gn > 0) {          // host guarantees gridDim.x % gn == 0, gridDim.y % (8 / gn) == 0, gridDim.z == 1
        const unsigned lin = by * gridDim.x + bx, xcd = lin & 7u, idx = lin >> 3;
        const unsigned gn = (unsigned)p.swz_gn, gm = 8u / gn, ntl = gridDim.x / gn;
        bx = (xcd % gn) * ntl + idx % ntl;
        by = (idx / ntl) * gm + xcd / gn;
    }
    do {                                                          // (one pass unless SK: a pass per tile segment)
    int tid = threadIdx.x;
    // (SK: opaque per segment.  Otherwise every lane-constant address part below is loop-invariant, gets hoisted out
    // of the segment loop and stays live through the main loop: 56 more registers on the 64 x 128 tile)
    if constexpr (SK) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN;
    const int wk = wave / WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    int k0, kend, S, slot;
    unsigned tile_id, slab0;                                      // ticket of the tile; slab of its first contributor
    if constexpr (SK) {
        const int Cn = p.sk_C;
        const int tl = (int)mdiv((unsigned)sk_u, p.sk_mC), c0 = sk_u - tl * Cn, c1 = min(Cn, c0 + (sk_uend - sk_u));
        if (p.sk_R > 1) { bx = (unsigned)tl; by = (unsigned)sk_lane; }
        else { bx = mdiv((unsigned)tl, p.sk_mmb); by = (unsigned)tl - bx * (unsigned)p.sk_mblocks; }
        k0 = c0 * C::KB;
        kend = min(p.K, c1 * C::KB);
        // the workgroups that hold units of this tile: owner(u) = the g whose unit range contains u
        const int thr = p.sk_rem * (p.sk_per + 1);
        auto owner = [&](int u) -> int {
            return u < thr ? (int)mdiv((unsigned)u, p.sk_mper1) : p.sk_rem + (int)mdiv((unsigned)(u - thr), p.sk_mper);
        };
        const int gfirst = owner(tl * Cn), glast = owner(tl * Cn + Cn - 1);
        S = glast - gfirst + 1;
        slot = sk_g - gfirst;
        tile_id = by * (unsigned)p.ntile_n + bx;
        // (workgroup, tile) pairs that share a tile ascend in g + tile: G + T slabs hold them all
        slab0 = (unsigned)(sk_lane * (p.sk_Gr + sk_Tl) + gfirst + tl);
        sk_u += c1 - c0;
    } else {
        k0 = blockIdx.z * p.k_per_split;
        kend = min(p.K, k0 + p.k_per_split);
        S = gridDim.z;
        slot = blockIdx.z;
        tile_id = by * gridDim.x + bx;
        slab0 = tile_id * (unsigned)S;
    }
    const int m0 = by * C::BM;
    const int n0 = bx * C::BN + wn * 32;
    const int nchunks = (kend - k0 + C::KB - 1) / C::KB;
    const bool fullM = (m0 + C::BM <= p.M);

    // this lane's weight row (clamped: out-of-range columns compute garbage that is never stored)
    const int nrow = min(n0 + l31, p.N - 1);
    // Row-major weights: lane (column l31, k half lh) reads 16 bytes of its own row - 32 rows per wave load,
    // 32 bytes used of every 128-byte line touched.  Tiled weights: the 32 columns x 32 k of one wave load are
    // stored as one contiguous KiB in lane order, so a load touches 8 full lines instead of 32 partial ones.
    // (W16: row-major only; the lane's 16 k of a step are 32 contiguous bytes)
    const unsigned kmul = W16 ? 2u : (p.wtiled ? 32u : 1u);       // bytes the load address advances per k
    // (a strip past the last one re-reads the last strip, like the clamped row above: never stored)
    const unsigned wstrip = min((unsigned)(n0 >> 5), (unsigned)((p.N + 31) / 32 - 1));
    const unsigned woff = W16 ? (unsigned)nrow * (unsigned)p.K * 2u + (unsigned)(32 * lh)
                        : p.wtiled ? wstrip * (unsigned)p.K * 32u + (unsigned)lane * 16u
                                   : (unsigned)nrow * (unsigned)p.K + (unsigned)(16 * lh);   // + uniform k offset per load

    // activation staging: thread t moves piece ci = t % CPR of rows row0 + j * RS (j < NP)
    static_assert(C::NT % C::CPR == 0 || C::CPR % C::NT == 0, "staging pattern");
    constexpr int RS = (C::NT >= C::CPR) ? C::NT / C::CPR : 1;   // row step between a thread's pieces
    static_assert(C::NT >= C::CPR, "a row's pieces must not outnumber the threads");
    const int st_ci = tid % C::CPR;
    const int st_row0 = tid / C::CPR;
    constexpr unsigned XB = DYNQ ? 2u : (unsigned)C::AB;           // bytes per activation element in memory
    constexpr unsigned XPB = 16u * XB / C::AB;                    // bytes of one staged piece in memory (16 or 32)
    const unsigned xoff0 = ((unsigned)(m0 + st_row0) * (unsigned)p.K) * XB + (unsigned)st_ci * XPB;
    const int st_k = st_ci * (16 / C::AB);                        // k of this thread's piece inside a chunk

    // A-fragment reads: row 32*i + l31; the XOR swizzle only involves lane-constant bits
    const int a_g4 = ((((l31 / SWZ_RPB) & SWZ_MASK) ^ (ACT == ACT_BF16 ? 2 * lh : lh)) & SWZ_MASK) << 4;
    const int a_hi = (ACT == ACT_BF16 ? 2 * lh : lh) & ~SWZ_MASK;   // 0 unless a row has < 4 pieces (never)
    const int a_row = l31 * RB + (a_hi << 4);

    // qkv epilogue: what the thread's first (row, head, piece) needs from memory (cache slot, cos/sin of the token's
    // position, norm weights) is requested NOW, so that this chain of dependent reads is over long before the tail
    // (only where one pass of the workgroup's threads covers the tile, i.e. the decode tiles: there the tail's
    // latency is what counts; the large prefill tiles have no registers to spare and load at the tail)
    constexpr bool ROPE_EARLY = ROPE && !SK && (C::BM * (C::BN / 128) * 8 <= C::NT);
    RopePre rope_pre;
    if constexpr (ROPE_EARLY) {
        constexpr int HPT = C::BN / 128;
        const int row = (tid >> 3) / HPT, hh = (tid >> 3) % HPT;
        if (tid < C::BM * HPT * 8 && m0 + row < p.M && (int)(bx * HPT) + hh < p.rope.Hq + 2 * p.rope.Hkv)
            rope_pre = rope_prefetch(m0 + row, (int)(bx * HPT) + hh, (tid & 7) * 8, p.rope);
    }

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

    constexpr int D = DB ? CP : (CP < 8 ? CP : 8);               // weight loads in flight per lane
    // DEEP (the 8-strip prefill tiles on 16-bit activations): activations and weights are requested TWO chunks
    // ahead.  Every chunk ends on a workgroup barrier behind the slowest of its 4096 16-byte requests, and with long
    // K slices one chunk of lead does not cover that tail; a second register set per operand buys a second chunk of
    // lead, while two LDS buffers still suffice.
    constexpr bool DEEP = DB && WN == 8 && ACT == ACT_BF16 && !W16 && (CP >= 4 || GROUP);
    wreg_t wcur[D], wnext[(DB && WN != 8) ? CP : 1], wcur2[DEEP ? CP : 1];
    xreg_t sreg[C::NP], sreg2[DEEP ? C::NP : 1];
    auto xload = [&](const unsigned char* ptr) -> xreg_t {
        if constexpr (DYNQ) return W2{gload16(ptr), gload16(ptr + 16)};
        else return gload16(ptr);
    };
    auto xzero = [&]() -> xreg_t {
        if constexpr (DYNQ) return W2{i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};
        else return i32x4{0, 0, 0, 0};
    };
    // DYNQ: the row maxima come in as partial maxima over any split of the row's columns (what the producing
    // kernels' workgroups each see); their maximum gives exactly the scale pair of ql_quant_per_row
    if constexpr (DYNQ) {
        float* dq = (float*)(smem + DQ_OFF);
        constexpr int TPR = (C::NT >= C::BM) ? C::NT / C::BM : 1;  // threads per row (a power of two, at most 16)
        static_assert(C::NT >= C::BM && (TPR & (TPR - 1)) == 0 && TPR <= 64, "row maxima: whole lane groups");
        constexpr float QMAX = (ACT == ACT_I8) ? 127.f : 448.f;
        const int r = tid / TPR, part0 = tid % TPR;
        float mx = 0.f;
        if (m0 + r < p.M)
            for (int i = part0; i < p.nparts; i += TPR) mx = fmaxf(mx, p.rowmax_part[(size_t)(m0 + r) * p.nparts + i]);
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (part0 == 0) {
            const float rm = fmaxf(mx, 1e-8f);
            dq[r] = QMAX / rm;
            dq[C::BM + r] = rm / QMAX;
        }
        __syncthreads();
    }

    // 16 weight bytes of step c of the chunk at kc (uniform base pointer + 32-bit lane offset)
    auto load_w_at = [&](int kb) -> wreg_t {
        if constexpr (W16) return W2{gload16(p.w + kb * kmul + woff), gload16(p.w + kb * kmul + woff + 16)};
        else {
            // Single-buffer (decode) tiles: the weights are a read-once stream for the die that reads them, so they are
            // loaded with the non-temporal hint (measured: -0.2..0.4 us per decode GEMM of Qwen3-0.6B, +1.5 % on the
            // whole bench).  The double-buffered (prefill) tiles re-read every weight slice from L2 across their
            // row tiles and keep the default policy.
            if constexpr (!DB) return __builtin_nontemporal_load((const __attribute__((address_space(1))) i32x4*)(p.w + kb * kmul + woff));
            else return gload16(p.w + kb * kmul + woff);
        }
    };
    auto load_w1 = [&](auto chk, int kc, int c) -> wreg_t {
        const int kb = kc + wk * C::KW + c * 32;
        if (!decltype(chk)::value || kb < kend) return load_w_at(kb);
        if constexpr (W16) return W2{i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};
        else return i32x4{0, 0, 0, 0};
    };
    auto load_x_to = [&](auto chk, int kc, xreg_t* sreg) {
        if (fullM) {
            const bool ok = !decltype(chk)::value || (kc + st_k < kend);
            const unsigned off = ok ? xoff0 : (unsigned)(m0 * p.K) * XB;         // any valid address when past the slice
            const unsigned char* xs = p.x + (size_t)kc * XB;
#pragma unroll
            for (int j = 0; j < C::NP; ++j)
                sreg[j] = xload(xs + (size_t)j * RS * p.K * XB + off);
        } else {
            // last row block: rows past M are not fetched (zeros: their outputs are never stored).  All lanes of a
            // wave hold pieces of the same row whenever a row has at least 64 pieces, so the branch is uniform.
#pragma unroll
            for (int j = 0; j < C::NP; ++j) {
                const int row = m0 + st_row0 + j * RS;
                const int k = (kc + st_k < kend) ? kc + st_k : k0;
                if (row < p.M) sreg[j] = xload(p.x + ((size_t)row * p.K + k) * XB);
                else sreg[j] = xzero();
            }
        }
    };
    auto store_x_from = [&](int buf, const xreg_t* sreg) {
#pragma unroll
        for (int j = 0; j < C::NP; ++j) {
            i32x4 piece;
            if constexpr (DYNQ) {
                // the arithmetic of quant_per_row_kernel: y = x * (QMAX / max), round half away (int8) or e4m3 RTNE
                const float inv = ((const float*)(smem + DQ_OFF))[st_row0 + j * RS];
                const u16x8 lo = __builtin_bit_cast(u16x8, sreg[j].a), hi = __builtin_bit_cast(u16x8, sreg[j].b);
                unsigned o[4] = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float y = bf16_bits_to_f32(e < 8 ? lo[e] : hi[e - 8]) * inv;
                    unsigned b;
                    if (ACT == ACT_I8) b = (unsigned)(unsigned char)(signed char)(int)roundf(y);
                    else b = f32_to_e4m3fn(y);
                    o[e >> 2] |= b << (8 * (e & 3));
                }
                piece = i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
            } else {
                piece = sreg[j];
            }
            *(i32x4*)(smem + buf * C::XBYTES + lds_off<RB>(st_row0 + j * RS, st_ci)) = piece;
        }
    };
    // group scales of the chunk at kc: thread t requests entries e = t + j * NT of [groups of the chunk][BN columns]
    float screg[NSL];
    // k / group without a division (an integer division costs ~30 instructions, and the main loop needs several per
    // chunk): every k here is a multiple of 32, so the quotient is (k / 32) / (group / 32) by multiply-high
    auto gdiv = [&](int k) -> int {
        const unsigned q = (unsigned)k >> 5;
        return p.gmagic ? (int)__umulhi(q, p.gmagic) : (int)q;
    };
    const int ngk = GROUP ? gdiv(p.K) : 1;                         // scales per weight row
    int s_cnt = 0;                                                // entries of the chunk whose scales are in screg
    auto load_s = [&](int kc) {
        if constexpr (SLDS) {
            const int g0 = gdiv(kc), cnt = C::BN * (gdiv(min(kc + C::KB, kend) - 32) - g0 + 1);
            s_cnt = cnt;
            const int nblk = (int)(bx * C::BN);
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                const int e = tid + j * C::NT;
                const int col = min(nblk + (e % C::BN), p.N - 1);
                if (e < cnt) screg[j] = p.sw[(size_t)col * ngk + g0 + e / C::BN];
            }
        }
    };
    auto store_s = [&](int buf, int kc) {
        if constexpr (SLDS) {
            const int cnt = s_cnt;                                // (the chunk load_s was last called for: kc)
            float* dst = (float*)(smem + SC_OFF + buf * SC_BYTES);
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                const int e = tid + j * C::NT;
                if (e < cnt) dst[e] = screg[j];
            }
        }
    };
    auto load_x = [&](auto chk, int kc) { load_x_to(chk, kc, sreg); load_s(kc); };
    auto store_x = [&](int buf, int kc) { store_x_from(buf, sreg); store_s(buf, kc); };
    // Double-buffered tiles: the scale of the group a wave is summing sits in a register, requested one fold ahead
    // (at the fold before it, for the group of the wave's next step).  Requested at the fold itself, the load's wait
    // drained every weight and activation request issued before it - on the prefill tiles the two chunks of
    // operands they keep in flight - at every group end.
    float s_ahead = 0.f;
    if constexpr (GROUP && !SLDS) s_ahead = p.sw[(size_t)nrow * ngk + min(gdiv(min(k0 + wk * C::KW, p.K - 32)), ngk - 1)];
    auto flush_group = [&](int kb, int kc, int buf, int c, bool clear) __attribute__((always_inline)) {
        // fold the finished group's partial sums into the running sum with the group's f32 scale
        float s;
        if constexpr (SLDS) {
            s = ((const float*)(smem + SC_OFF + buf * SC_BYTES))[(gdiv(kb) - gdiv(kc)) * C::BN + wn * 32 + l31];
        } else {
            s = s_ahead;
            // the wave's next step: the next 32 k of its chunk part, or its part of the next chunk
            const int kn = (c == CP - 1) ? kc + C::KB + wk * C::KW : kb + 32;
            s_ahead = p.sw[(size_t)nrow * ngk + gdiv(min(kn, p.K - 32))];
        }
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 s2 = {s, s};
#if !QL_FOLD_PK
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][r] = __builtin_fmaf((float)accg[i][r], s, (float)acc[i][r]);
                if (clear) accg[i][r] = 0;
            }
        return;
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // (two accumulator registers per instruction: v_pk_fma_f32, the same fused arithmetic per element)
                const f32x2 g2 = {(float)accg[i][r], (float)accg[i][r + 1]};
                const f32x2 a2 = {(float)acc[i][r], (float)acc[i][r + 1]};
                const f32x2 o2 = __builtin_elementwise_fma(g2, s2, a2);
                acc[i][r] = o2[0];
                acc[i][r + 1] = o2[1];
                if (clear) {
                    accg[i][r] = 0;
                    accg[i][r + 1] = 0;
                }
            }
    };
    // one 32-k step of this wave: widen the 16 weight bytes of the lane, MT x (A fragment reads + MFMAs)
    wreg_t w_even{};                                              // (fp8 x fp8: the even step's weights, see below)
    // GM (per-group kernels): where the groups lie relative to the steps, when the host-side shape makes that a
    // compile-time fact.  0: anything (tested per step).  1: group = 128 k and the wave's chunk part is a whole number
    // of groups (CP % 4 == 0): step ph starts a group when ph % 4 == 0 and ends one when ph % 4 == 3.  2 / 3: group =
    // 128 k, chunk part = 64 k (CP == 2, WK == 1), this chunk is the first / second half of a group.  In the modes
    // 1-3 the first MFMA of a group takes a ZERO C operand instead of the group accumulator, which then never needs
    // clearing (two thirds of the fold's instructions); the sums are the same bit for bit (0 + a b).
    // ph = the step's position in the chunk as the unrolled loops know it at compile time (ph % 4 == c % 4).
    auto step = [&](auto chk, auto gmc, int buf, int kc, int c, int ph, const wreg_t w) __attribute__((always_inline)) {
        constexpr int GM = decltype(gmc)::value;
        const bool gstart = (GM == 1 && (ph & 3) == 0) || (GM == 2 && ph == 0);
        const int kb = kc + wk * C::KW + c * 32;   // first k of this step (wave-uniform)
        if (decltype(chk)::value && kb >= kend) return;
        const int base_c = (wk * CP + c) * APS;    // first piece of the step in an LDS row (uniform)
        // piece index (base_c | lane part) with the low bits XOR-swizzled by the row; kept as an LDS byte OFFSET
        // (an integer), not a pointer: xor-ing a pointer through uintptr_t makes it generic and the read a flat load
        const int a0 = buf * C::XBYTES + a_row + ((base_c & ~SWZ_MASK) << 4) + (((base_c & SWZ_MASK) << 4) ^ a_g4);
        if constexpr (ACT == ACT_BF16) {
            bf16x8 blo, bhi;
            if constexpr (W16) {
                blo = __builtin_bit_cast(bf16x8, w.a);
                bhi = __builtin_bit_cast(bf16x8, w.b);
            } else {
                blo = widen8<FP8W>(w[0], w[1]);
                bhi = widen8<FP8W>(w[2], w[3]);
            }
            const int a1 = a0 ^ 16;                                // piece index + 1 == ^1 (index is even)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bf16x8 alo = *(const bf16x8*)(smem + a0 + i * 32 * RB);
                const bf16x8 ahi = *(const bf16x8*)(smem + a1 + i * 32 * RB);
                if (GROUP) {
                    if (gstart) accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, blo, f32x16{}, 0, 0, 0);
                    else accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, blo, accg[i], 0, 0, 0);
                    accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, accg[i], 0, 0, 0);
                } else {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, blo, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[i], 0, 0, 0);
                }
            }
        } else {
            if constexpr (ACT == ACT_FP8 && !W16 && !DYNQ && QL_FP8_K64) {
                // fp8 x fp8 at the chip's full fp8 rate: two 32-k steps per v_mfma_scale_f32_32x32x64_f8f6f4 (unit
                // scales, e4m3 operands), which runs 64 k in the cycles the K = 16 instruction pair above takes for
                // 32.  A lane's 32 operand bytes are its 16 of the even step and its 16 of the odd step, for A and
                // B alike, so the contraction is again a permutation of the k sum.  The even step only parks its
                // weights; a lone last step (K slice not a multiple of 64) takes the K = 16 path below.
                typedef int i32x8 __attribute__((ext_vector_type(8)));
                if ((c & 1) == 0 && (!decltype(chk)::value || kb + 32 < kend)) {
                    w_even = w;
                    return;
                }
                if (c & 1) {
                    const int bc0 = base_c - APS;                  // the even step's first piece
                    const int a_ev = buf * C::XBYTES + a_row + ((bc0 & ~SWZ_MASK) << 4) + (((bc0 & SWZ_MASK) << 4) ^ a_g4);
                    const i32x8 bb = {w_even[0], w_even[1], w_even[2], w_even[3], w[0], w[1], w[2], w[3]};
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const i32x4 ae = *(const i32x4*)(smem + a_ev + i * 32 * RB);
                        const i32x4 ao = *(const i32x4*)(smem + a0 + i * 32 * RB);
                        const i32x8 aa = {ae[0], ae[1], ae[2], ae[3], ao[0], ao[1], ao[2], ao[3]};
                        // (cbsz = blgp = 0: both operands e4m3; scale bytes 0x7F = 2^0)
                        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aa, bb, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                    }
                    return;
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const i32x4 a = *(const i32x4*)(smem + a0 + i * 32 * RB);
                if constexpr (ACT == ACT_I8) {
                    acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w, acc[i], 0, 0, 0);
                } else {
                    const long a0l = ((long)(unsigned)a[1] << 32) | (unsigned)a[0];
                    const long a1l = ((long)(unsigned)a[3] << 32) | (unsigned)a[2];
                    const long b0 = ((long)(unsigned)w[1] << 32) | (unsigned)w[0];
                    const long b1 = ((long)(unsigned)w[3] << 32) | (unsigned)w[2];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0l, b0, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1l, b1, acc[i], 0, 0, 0);
                }
            }
        }
        if (GROUP) {
            // flush at every true group end, and at the end of a wave chunk (the next chunk of this wave
            // starts at another k: it may belong to another group or continue this one, and both are handled
            // by scaling what has been summed so far with the current group's scale).  Any group size that is
            // a multiple of the 32-k step is exact this way, whatever its relation to the chunk length.
            if constexpr (GM == 0) {
                if (gdiv(kb + 32) * p.group == kb + 32 || c == CP - 1 || kb + 32 >= kend) flush_group(kb, kc, buf, c, true);
            } else {
                // (slices start and end on group boundaries in these modes: a group is never cut)
                if ((GM == 1 && (ph & 3) == 3) || (GM == 3 && ph == CP - 1)) flush_group(kb, kc, buf, c, false);
            }
        }
    };

    // one chunk, CHK = the slice may end inside this chunk (per-step bounds tests needed)
    // RING (the 8-strip prefill tiles, which live at the register limit): the weights of step c of the NEXT
    // chunk are requested as soon as step c of this chunk has taken its register, so one chunk of weight loads
    // is in flight with CP registers instead of 2 CP, which is what lets the 128 x 256 tile run 128-k chunks
    // without spilling (+8-10 % at M = 16384 over 64-k chunks).  The address is clamped into the slice instead
    // of branching around the load (past the end the last step is re-read and never used: step() skips it).
    constexpr bool RING = DB && WN == 8;
    auto load_w_ring = [&](int kc, int c) -> wreg_t {
        const int kb = min(kc + wk * C::KW + c * 32, kend - 32);
        return load_w_at(kb);
    };
    auto chunk_db = [&](auto chk, auto gmc, int it) __attribute__((always_inline)) {
        const int kc = k0 + it * C::KB;
        const int buf = it & 1;
        const bool more = (it + 1) < nchunks;
        if constexpr (RING) {
            if (more) load_x(TrueT{}, kc + C::KB);
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                const wreg_t w = wcur[c];
                wcur[c] = load_w_ring(kc + C::KB, c);
                step(chk, gmc, buf, kc, c, c, w);
            }
            if (more) store_x(buf ^ 1, kc + C::KB);
        } else {
            if (more) {
#pragma unroll
                for (int c = 0; c < CP; ++c) wnext[c] = load_w1(TrueT{}, kc + C::KB, c);
                load_x(TrueT{}, kc + C::KB);
            }
#pragma unroll
            for (int c = 0; c < CP; ++c) step(chk, gmc, buf, kc, c, c, wcur[c]);
            if (more) {
                store_x(buf ^ 1, kc + C::KB);
#pragma unroll
                for (int c = 0; c < CP; ++c) wcur[c] = wnext[c];
            }
        }
        __syncthreads();
    };
    // DEEP: chunk `it` (parity PAR) computes from LDS buffer PAR and weight set PAR, requests chunk it + 2 into the
    // registers it frees, and ends by writing chunk it + 1 (requested during chunk it - 1) to the other LDS buffer
    auto chunk_deep = [&](auto chk, auto gsel, auto par, int it) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        // (group mode of the steps: whole groups per chunk part, or - 64-k chunks - the half this chunk is)
        using GMC = std::integral_constant<int, !decltype(gsel)::value ? 0 : (CP % 4 == 0) ? 1 : 2 + PAR>;
        const int kc = k0 + it * C::KB;
        wreg_t* wset = PAR ? wcur2 : wcur;
        if (it + 2 < nchunks) load_x_to(TrueT{}, kc + 2 * C::KB, PAR ? sreg2 : sreg);
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const wreg_t w = wset[c];
            wset[c] = load_w_ring(kc + 2 * C::KB, c);
            step(chk, GMC{}, PAR, kc, c, c, w);
        }
        if (it + 1 < nchunks) store_x_from(PAR ^ 1, PAR ? sreg : sreg2);
        __syncthreads();
    };
    auto chunk_sb = [&](auto chk, auto gmc, int it) __attribute__((always_inline)) {
        const int kc = k0 + it * C::KB;
        // activations first: the wait counter is positional, so staging them must not wait for the (slower,
        // HBM) weight loads issued behind them
        load_x(chk, kc);
#pragma unroll
        for (int j = 0; j < D; ++j) wcur[j] = load_w1(chk, kc, j);
        store_x(0, kc);
        __syncthreads();
        for (int c0 = 0; c0 < CP; c0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const wreg_t w = wcur[j];
                if (c0 + D + j < CP) wcur[j] = load_w1(chk, kc, c0 + D + j);
                step(chk, gmc, 0, kc, c0 + j, j, w);                  // (c0 is a multiple of D = 4 or 8)
            }
        }
        __syncthreads();
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // gsel = 1: the group-aligned form of the steps (see step); one copy of the loops per form, chosen once per slice
    auto main_loop = [&](auto gsel) __attribute__((always_inline)) {
    constexpr bool GS = decltype(gsel)::value != 0;
    using GM1 = std::integral_constant<int, GS ? 1 : 0>;           // (the non-DEEP loops: whole groups per chunk part)
    if constexpr (DEEP) {
        if (nchunks > 0) {
            load_x_to(TrueT{}, k0, sreg);                         // (before the weights: positional wait counter)
#pragma unroll
            for (int c = 0; c < CP; ++c) wcur[c] = load_w1(TrueT{}, k0, c);
            if (nchunks > 1) load_x_to(TrueT{}, k0 + C::KB, sreg2);
#pragma unroll
            for (int c = 0; c < CP; ++c) wcur2[c] = load_w_ring(k0 + C::KB, c);
            store_x_from(0, sreg);
        }
        __syncthreads();
        // whole chunks in pairs (the register sets alternate with the chunk's parity), then what is left: at most
        // one whole chunk and at most one ragged one
        const int nfull = (kend - k0) / C::KB;
        int it = 0;
        for (; it + 1 < nfull; it += 2) {
            chunk_deep(FalseT{}, gsel, I0{}, it);
            chunk_deep(FalseT{}, gsel, I1{}, it + 1);
        }
        if (it < nfull) {
            chunk_deep(FalseT{}, gsel, I0{}, it);
            if (it + 1 < nchunks) chunk_deep(TrueT{}, gsel, I1{}, it + 1);
        } else if (it < nchunks) {
            chunk_deep(TrueT{}, gsel, I0{}, it);
        }
    } else if constexpr (DB) {
        // LDS double buffer for x, register double buffer for the weights
        if (nchunks > 0) {
            load_x(TrueT{}, k0);                                  // (before the weights: positional wait counter)
#pragma unroll
            for (int c = 0; c < CP; ++c) wcur[c] = load_w1(TrueT{}, k0, c);
            store_x(0, k0);
        }
        __syncthreads();
        // (whole chunks in a loop of ONE body, the ragged last chunk after it: with both variants inside the loop
        // the accumulators get a second home, which costs the large tiles 60-150 registers or spills)
        const int nfull = (kend - k0) / C::KB;
        int it = 0;
        for (; it < nfull; ++it) chunk_db(FalseT{}, GM1{}, it);
        if (it < nchunks) chunk_db(TrueT{}, GM1{}, it);
    } else {
        // single buffer: the activation tile of the whole K slice (normally ONE chunk) is resident in
        // LDS; after the one staging barrier every wave streams its weights through a register ring
        // of D loads in flight and never synchronises with the other waves again
        for (int it = 0; it < nchunks; ++it) {
            if (k0 + (it + 1) * C::KB <= kend) chunk_sb(FalseT{}, GM1{}, it);
            else chunk_sb(TrueT{}, GM1{}, it);
        }
    }
    };
    // (built where a second copy of the loops costs the tile no registers: the double-buffered tiles and the one-fragment
    // streaming tiles of up to 8 steps; the 16- and 32-step tiles, which live at the register limit, keep one form)
    if constexpr (GROUP && ACT == ACT_BF16 && (DB || (CP <= 8 && MT == 1)) && (CP % 4 == 0 || (CP == 2 && WK == 1 && DEEP))) {
        // 128-k groups that the steps meet at compile-time positions (the slice must start and end on a group
        // boundary when a chunk part is half a group; with whole groups per part it always does)
        const bool aligned = (CP % 4 == 0) || (((k0 | (kend - k0)) & 127) == 0);
        if (p.group == 128 && p.group_generic == 0 && aligned) main_loop(I1{});
        else main_loop(I0{});
    } else {
        main_loop(I0{});
    }

    // ---- tail: K reduction, split-K hand-over and epilogue, spread over ALL waves of the block ----
    // After the reduction wave wk of a strip owns accumulator registers [wk*RN, wk*RN + RN) of every
    // fragment, i.e. 8*RN/4 ... rows of each 32-row fragment, so the slab traffic, the scaling and the
    // stores are done by WK times more lanes than if wave 0 kept the whole tile.  Register indices must
    // be compile-time constants, hence one instantiation of the tail per wk.
    constexpr int RN = 16 / WK;
    const int n = n0 + l31;
    const bool nvalid = n < p.N;
    float sw = 1.f, bias = 0.f;
    if (!GROUP && !W16 && nvalid && p.sw) sw = p.sw[(size_t)n * p.sw_stride];   // (null: a deferred W8A8 launch, scaled by its consumer)
    const bool has_bias = p.bias != nullptr;
    if (has_bias && nvalid) bias = bf16_bits_to_f32(p.bias[n]);

    auto tail = [&](auto wkc) {
        constexpr int WKC = decltype(wkc)::value;
        constexpr int R0 = WKC * RN;
        if constexpr (WK > 1) {
            // every wave parks the registers it does not own; the owner sums the WK partials in wave order
            raw_t* red = (raw_t*)smem + (size_t)wn * (MT * 16 * 64) + lane;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r < R0 || r >= R0 + RN) red[((WKC * WN * MT + i) * 16 + r) * 64] = acc[i][r];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rr = 0; rr < RN; ++rr) {
                    raw_t sum = 0;
#pragma unroll
                    for (int j = 0; j < WK; ++j)
                        sum += (j == WKC) ? acc[i][R0 + rr] : red[((j * WN * MT + i) * 16 + R0 + rr) * 64];
                    acc[i][R0 + rr] = sum;
                }
        }
        if (p.partial_rm) {
            // deferred reduction: no ticket, no second pass here.  The consumer launch sums the slices in slice
            // order and applies the epilogue, which takes two dependent memory round trips (ticket, slab
            // reads) out of this kernel and lets K be cut as finely as the chip needs workgroups.
            // (int8 x int8: the slabs hold the int32 sums themselves, so the consumer adds integers exactly)
            if (!nvalid) return;
            raw_t* dst = (raw_t*)p.partial_rm + (size_t)slot * (size_t)p.M * (size_t)p.N + n;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rr = 0; rr < RN; ++rr) {
                    const int r = R0 + rr;
                    const int row = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < p.M) dst[(size_t)row * p.N] = (raw_t)acc[i][r];
                }
            return;
        }
        if (S > 1) {
            // split-K across workgroups: slab store, ticket, the last arriver sums the slabs in slice order
            const unsigned tile = tile_id;
            // (a slab holds the tile as register PAIRS: pair q of a lane at [strip][q][lane], 8 bytes per access)
            typedef unsigned long long u64;
            u64* slab = (u64*)p.partial + (size_t)slab0 * (WN * MT * 8 * 64) + (size_t)wn * (MT * 8 * 64) + lane;
            u64* mine = slab + (size_t)slot * (WN * MT * 8 * 64);
            auto pack2 = [](raw_t lo, raw_t hi) -> u64 {
                return ((u64)__builtin_bit_cast(unsigned, hi) << 32) | (u64)__builtin_bit_cast(unsigned, lo);
            };
            // Partial tiles travel between workgroups (possibly on different XCDs, whose L2s are not
            // coherent with each other) as agent-scope relaxed atomics: write-through stores, cache-
            // bypassing loads.  Completed stores (vmcnt(0) + barrier) before the ticket and the ticket
            // before the loads give the ordering without an L2 write-back / invalidate per workgroup.
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rr = 0; rr < RN; rr += 2)
                    __hip_atomic_store(mine + (i * 8 + (R0 + rr) / 2) * 64, pack2(acc[i][R0 + rr], acc[i][R0 + rr + 1]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                  // also: every wave is done with smem
            unsigned* flag = (unsigned*)smem;
            if (tid == 0) {
                const unsigned t = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned last = (t == (unsigned)(S - 1)) ? 1u : 0u;
                if (last) __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = last;
            }
            __syncthreads();
            if (*flag == 0u) return;
            // The last arriver still holds its own slice in registers and does not read it back.  The result must not
            // depend on WHO is last, so the sum keeps the slice order ((0 + p0) + p1) + ...: the slices in front of
            // this workgroup's are summed into temporaries, t + own equals own + t bit for bit (the sum they would
            // give in either order), and the slices behind it are added to that.
            // (two row fragments at a time: enough loads in flight per slice, and the temporaries stay few)
            constexpr int FB = (MT % 2 == 0) ? 2 : 1;
            auto fix_block = [&](auto i0c) {
                constexpr int I0 = decltype(i0c)::value;
                raw_t pre[FB][RN];
#pragma unroll
                for (int i = 0; i < FB; ++i)
#pragma unroll
                    for (int rr = 0; rr < RN; ++rr) pre[i][rr] = 0;
                auto add_slab = [&](int s2, auto to_pre) {
                    const u64* src = slab + (size_t)s2 * (WN * MT * 8 * 64);
#pragma unroll
                    for (int i = 0; i < FB; ++i)
#pragma unroll
                        for (int rr = 0; rr < RN; rr += 2) {
                            const u64 v = __hip_atomic_load(src + ((I0 + i) * 8 + (R0 + rr) / 2) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const raw_t lo = __builtin_bit_cast(raw_t, (unsigned)v), hi = __builtin_bit_cast(raw_t, (unsigned)(v >> 32));
                            if constexpr (decltype(to_pre)::value) { pre[i][rr] += lo; pre[i][rr + 1] += hi; }
                            else { acc[I0 + i][R0 + rr] += lo; acc[I0 + i][R0 + rr + 1] += hi; }
                        }
                };
                for (int s2 = 0; s2 < slot; ++s2) add_slab(s2, TrueT{});
#pragma unroll
                for (int i = 0; i < FB; ++i)
#pragma unroll
                    for (int rr = 0; rr < RN; ++rr) acc[I0 + i][R0 + rr] = pre[i][rr] + acc[I0 + i][R0 + rr];
                for (int s2 = slot + 1; s2 < S; ++s2) add_slab(s2, FalseT{});
            };
            auto all_blocks = [&](auto self, auto i0c) {
                constexpr int I0 = decltype(i0c)::value;
                if constexpr (I0 < MT) {
                    fix_block(i0c);
                    self(self, std::integral_constant<int, I0 + FB>{});
                }
            };
            all_blocks(all_blocks, std::integral_constant<int, 0>{});
        }

        // epilogue: scale (+ bias), one hardware RTNE conversion, 2-byte store per element
        // element (fragment i, register r) of this lane as the bf16 the plain GEMM stores
        auto outv = [&](int i, int r, int rowc) -> bf16_t {
            float v = (float)acc[i][r];
            if constexpr (DYNQ) v = __fmul_rn(v, ((const float*)(smem + DQ_OFF))[C::BM + rowc + 4 * lh]);
            else if (ACT != ACT_BF16) v = __fmul_rn(v, p.sx[min(m0 + rowc + 4 * lh, p.M - 1)]);
            v = __fmul_rn(v, sw);
            if (has_bias) {
                if (p.bias_after_round) v = __fadd_rn((float)(bf16_t)v, bias);
                else v = __fadd_rn(v, bias);
            }
            return (bf16_t)v;
        };
        if (p.silu) {
            // Fused SiLU * mul: strip 2s of the (pre-arranged) weight holds gate columns [32s, 32s+32),
            // strip 2s+1 the matching up columns, so waves wn and wn^1 hold gate and up of the same
            // output elements in the same lanes.  Each of the two finishes half of the pair's owned
            // registers and gets the other operand of that half through LDS.
            if constexpr (WN >= 2) {
                constexpr int RH = RN / 2;
                __syncthreads();                              // every wave is done with the x tile / reduce area
                unsigned short* ex = (unsigned short*)smem + (size_t)((WKC * (WN / 2) + (wn >> 1)) * MT) * (16 * 64) + lane;
                const unsigned I = (unsigned)p.N >> 1;
                const unsigned col = (unsigned)((bx * WN + wn) >> 1) * 32u + (unsigned)l31;
                const unsigned lane_off = (unsigned)(4 * lh) * I + col;
                // the gate wave finishes registers [R0, R0+RH), the up wave [R0+RH, R0+RN)
                auto pair = [&](auto upc) {
                    constexpr bool UP = decltype(upc)::value;
                    constexpr int RMINE = R0 + (UP ? RH : 0), ROTHER = R0 + (UP ? 0 : RH);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int rr = 0; rr < RH; ++rr) {
                            const int r = ROTHER + rr;
                            ex[(i * 16 + r) * 64] = __builtin_bit_cast(unsigned short, outv(i, r, 32 * i + (r & 3) + 8 * (r >> 2)));
                        }
                    __syncthreads();
                    const bool want_max = p.rowmax_out != nullptr;          // (uniform)
                    if (!nvalid && !want_max) return;
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int rr = 0; rr < RH; ++rr) {
                            const int r = RMINE + rr;
                            const int rowc = 32 * i + (r & 3) + 8 * (r >> 2);
                            const bool row_ok = fullM || m0 + rowc + 4 * lh < p.M;   // (uniform per 32-lane half)
                            if (!row_ok) continue;
                            unsigned short v = 0;
                            if (nvalid) {
                                const float own = (float)outv(i, r, rowc);
                                const float oth = bf16_bits_to_f32(ex[(i * 16 + r) * 64]);
                                v = UP ? silu_mul_bf16(oth, own) : silu_mul_bf16(own, oth);
                                (p.y + (size_t)(m0 + rowc) * I)[lane_off] = v;
                            }
                            if (want_max) {
                                // |v| as bf16 bits orders like the value; a NaN counts as 0 (fmaxf in the quantizer
                                // drops it too).  Maximum over the 32 columns of this strip, which lie in one half-wave.
                                int a = v & 0x7FFF;
                                if (a > 0x7F80) a = 0;
#pragma unroll
                                for (int o = 16; o > 0; o >>= 1) a = max(a, __shfl_xor(a, o));
                                // (a tile wider than what is left of N: strips past the last one have no slot)
                                if (l31 == 0 && (int)((bx * WN + wn) >> 1) < p.nparts_out)
                                    p.rowmax_out[(size_t)(m0 + rowc + 4 * lh) * p.nparts_out + ((bx * WN + wn) >> 1)] =
                                        bf16_bits_to_f32((unsigned short)a);
                            }
                        }
                };
                if (wn & 1) pair(TrueT{});
                else pair(FalseT{});
            }
            return;
        }
        if constexpr (AMAX) {
            // output projection with the greedy sampler in the epilogue: the bf16 values the plain kernel would store
            // go to an LDS image; BN / 8 lanes per row then find the row's maximum over the tile's columns (smallest
            // column among equal maxima, NaN never wins: the rules of argmax_rows_kernel) and one of them stores it
            __syncthreads();                                  // every wave is done with the x tile
            unsigned short* img = (unsigned short*)smem;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rr = 0; rr < RN; ++rr) {
                    const int r = R0 + rr;
                    const int rowc = 32 * i + (r & 3) + 8 * (r >> 2);
                    img[(rowc + 4 * lh) * C::ROPE_RS + wn * 32 + l31] = __builtin_bit_cast(unsigned short, outv(i, r, rowc));
                }
            __syncthreads();
            constexpr int PPR = C::BN / 8;                    // lanes per row: 16 or 32, an aligned part of one wave
            static_assert((C::BM * PPR) % C::NT == 0 && 64 % PPR == 0, "whole lane groups, all threads in every pass");
            const int col0 = (int)(bx * C::BN);
#pragma unroll
            for (int idx = tid; idx < C::BM * PPR; idx += C::NT) {
                const int row = idx / PPR, pc = idx % PPR;
                const u16x8 a = *(const u16x8*)(img + row * C::ROPE_RS + pc * 8);
                float best = -INFINITY;
                int bi = 0x7FFFFFFF;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v = bf16_bits_to_f32(a[k]);
                    const int col = col0 + pc * 8 + k;
                    if (col < p.N && v > best) { best = v; bi = col; }
                }
#pragma unroll
                for (int o = PPR / 2; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(best, o);
                    const int oi = __shfl_xor(bi, o);
                    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                }
                if (pc == 0 && m0 + row < p.M)
                    p.amax[(size_t)(m0 + row) * p.ntile_n + bx] = int2{__builtin_bit_cast(int, best), bi};
            }
            return;
        }
        if constexpr (ROPE) {
            {
                // qkv projection with the q/k norm + RoPE + KV-cache store in the epilogue: the tile's columns are
                // whole 128-wide heads (host: N % 128 == 0), so the bf16 outputs go to an LDS image instead of y and
                // eight lanes per (row, head) then run exactly the arithmetic of the stand-alone operator on them.
                __syncthreads();                              // every wave is done with the x tile / reduce area
                unsigned short* img = (unsigned short*)smem;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int rr = 0; rr < RN; ++rr) {
                        const int r = R0 + rr;
                        const int rowc = 32 * i + (r & 3) + 8 * (r >> 2);
                        img[(rowc + 4 * lh) * C::ROPE_RS + wn * 32 + l31] = __builtin_bit_cast(unsigned short, outv(i, r, rowc));
                    }
                __syncthreads();
                constexpr int HPT = C::BN / 128;              // heads per tile
                const int head0 = (int)(bx * HPT);
                for (int idx = tid; idx < C::BM * HPT * 8; idx += C::NT) {
                    const int s8 = (idx & 7) * 8, hh = (idx >> 3) % HPT, row = (idx >> 3) / HPT;
                    if (m0 + row >= p.M) break;               // (rows ascend with idx; whole 8-lane groups leave together)
                    // a 256-column tile over an odd number of heads: the last tile's second head does not exist
                    if (head0 + hh >= p.rope.Hq + 2 * p.rope.Hkv) continue;
                    const unsigned short* src = img + row * C::ROPE_RS + hh * 128 + s8;
                    RopePre pre;
                    if constexpr (ROPE_EARLY) pre = rope_pre;      // (one pass: idx == tid, and the row was valid at entry)
                    else pre = rope_prefetch(m0 + row, head0 + hh, s8, p.rope);
                    rope_store_piece(*(const u16x8*)src, *(const u16x8*)(src + 64), m0 + row, head0 + hh, s8, p.rope, pre);
                }
                return;
            }
        }
        if (!nvalid) return;
        const unsigned lane_off = (unsigned)(4 * lh) * (unsigned)p.N + (unsigned)n;   // within the block's rows
        auto emit = [&](auto chk) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int rr = 0; rr < RN; ++rr) {
                    const int r = R0 + rr;
                    const int rowc = 32 * i + (r & 3) + 8 * (r >> 2);                // uniform part of the row
                    if (decltype(chk)::value && m0 + rowc + 4 * lh >= p.M) continue;
                    unsigned short* yrow = p.y + (size_t)(m0 + rowc) * p.N;          // uniform pointer
                    yrow[lane_off] = __builtin_bit_cast(unsigned short, outv(i, r, rowc));
                }
            }
        };
        if (fullM) emit(FalseT{});
        else emit(TrueT{});
    };
    if constexpr (WK == 1) {
        tail(std::integral_constant<int, 0>{});
    } else {
        switch (wk) {
            case 0: tail(std::integral_constant<int, 0>{}); break;
            case 1: tail(std::integral_constant<int, 1>{}); break;
            case 2: if constexpr (WK > 2) tail(std::integral_constant<int, 2>{}); break;
            case 3: if constexpr (WK > 2) tail(std::integral_constant<int, 3>{}); break;
            case 4: if constexpr (WK > 4) tail(std::integral_constant<int, 4>{}); break;
            case 5: if constexpr (WK > 4) tail(std::integral_constant<int, 5>{}); break;
            case 6: if constexpr (WK > 4) tail(std::integral_constant<int, 6>{}); break;
            case 7: if constexpr (WK > 4) tail(std::integral_constant<int, 7>{}); break;
        }
    }
    if constexpr (SK) __syncthreads();                            // the next segment's staging reuses the tail's LDS
    } while (SK && sk_u < sk_uend);
}

// The tile configurations that are built: {MT, WN, WK, CP, DB}
struct TileCfg { int mt, wn, wk, cp, db; };
#define QL_TILE_LIST(X)                                                                              \
    X(1, 1, 4, 2, true) X(1, 1, 4, 4, true) X(1, 1, 4, 8, false)                                     \
    X(2, 1, 4, 2, true) X(2, 1, 4, 4, true)                                                          \
    X(1, 4, 1, 4, true) X(2, 4, 1, 4, true) X(4, 4, 1, 4, true)                                      \
    X(2, 2, 2, 2, true) X(2, 2, 2, 4, true) X(2, 2, 2, 8, false)                                     \
    X(4, 2, 2, 2, true) X(4, 2, 2, 4, true)                                                          \
    X(1, 2, 4, 8, false) X(1, 2, 4, 4, true) X(2, 2, 4, 8, false) X(2, 2, 4, 4, true) X(1, 1, 8, 4, false)     \
    X(2, 4, 2, 4, true) X(4, 4, 2, 4, true)                                                          \
    X(1, 4, 1, 32, false) X(1, 4, 2, 16, false) X(2, 4, 2, 16, false) X(1, 2, 4, 16, false) X(1, 1, 8, 8, false) \
    X(4, 8, 1, 2, true) X(4, 8, 1, 4, true)
#define QL_TILE_LIST_TALL(X) X(8, 4, 1, 4, true)
// tiles also built with the qkv (q/k norm + RoPE + KV store) epilogue: 128- or 256-column tiles
#define QL_TILE_LIST_ROPE(X) X(1, 4, 2, 16, false) X(1, 4, 1, 4, true) X(2, 4, 2, 4, true) X(2, 4, 1, 4, true) X(4, 8, 1, 2, true) X(4, 8, 1, 4, true)

// tiles also built with the stream-K schedule (the double-buffered tiles of 64+ rows or 128+ columns that the
// mid-size problems - 64..512 rows on the larger models' shapes - want, and the streaming tiles of the small ones)
#define QL_TILE_LIST_SK(X)                                                                           \
    X(1, 1, 4, 4, true) X(1, 2, 4, 4, true) X(2, 2, 4, 4, true) X(2, 2, 2, 4, true)                  \
    X(1, 4, 1, 4, true) X(2, 4, 1, 4, true) X(4, 4, 1, 4, true) X(2, 4, 2, 4, true) X(4, 4, 2, 4, true) \
    X(4, 8, 1, 2, true) X(4, 8, 1, 4, true)                                                          \
    X(1, 1, 8, 4, false) X(1, 2, 4, 8, false) X(1, 4, 2, 16, false)
#define QL_TILE_LIST_SK_TALL(X) X(8, 4, 1, 4, true)
#define QL_TILE_LIST_SK_ROPE(X) X(1, 4, 1, 4, true) X(2, 4, 2, 4, true) X(2, 4, 1, 4, true) X(4, 8, 1, 2, true)

// one translation unit per arithmetic variant instantiates these
// The output projection (bf16 weights shared with the embedding table, vocabulary-sized N, M = sequences of a step)
// (129..224 rows: 160-, 192- and 224-row tiles, so that a step's sequences cost the MFMAs of the row fragments they fill;
// the 160-row tile also as a 4-wave, 128-column workgroup, two of which share a CU: the faster form up to 160 rows)
#define QL_TILE_LIST_HEAD(X) X(1, 4, 1, 4, true) X(2, 4, 1, 4, true) X(4, 4, 1, 4, true) X(4, 8, 1, 4, true) X(8, 8, 1, 2, true) \
    X(5, 8, 1, 4, true) X(6, 8, 1, 4, true) X(7, 8, 1, 4, true) X(5, 4, 1, 4, true)
template <bool AMAX>
kern_t pick_head_impl(int mt, int wn, int wk, int cp) {
#define QL_PICK_HEAD(MT_, WN_, WK_, CP_, DB_) \
    if (mt == MT_ && wn == WN_ && wk == WK_ && cp == CP_) return (kern_t)qgemm_kernel<ACT_BF16, false, MT_, WN_, WK_, CP_, DB_, false, false, true, AMAX>;
    QL_TILE_LIST_HEAD(QL_PICK_HEAD)
#undef QL_PICK_HEAD
    return nullptr;
}

template <int ACT, bool FP8W, bool GROUP>
kern_t pick_kernel_impl(int mt, int wn, int wk, int cp, bool rope) {
    if (rope) {
#define QL_PICK(MT_, WN_, WK_, CP_, DB_) \
        if (mt == MT_ && wn == WN_ && wk == WK_ && cp == CP_) return (kern_t)qgemm_kernel<ACT, FP8W, MT_, WN_, WK_, CP_, DB_, GROUP, true>;
        QL_TILE_LIST_ROPE(QL_PICK)
#undef QL_PICK
        return nullptr;
    }
#define QL_PICK(MT_, WN_, WK_, CP_, DB_) \
    if (mt == MT_ && wn == WN_ && wk == WK_ && cp == CP_) return (kern_t)qgemm_kernel<ACT, FP8W, MT_, WN_, WK_, CP_, DB_, GROUP>;
    QL_TILE_LIST(QL_PICK)
    if constexpr (!GROUP) { QL_TILE_LIST_TALL(QL_PICK) }
#undef QL_PICK
    return nullptr;
}

// the weight-only kernels on the stream-K schedule (translation units of their own: ql_gemm_inst_sk_*.hip)
template <bool FP8W, bool GROUP>
kern_t pick_sk_impl(int mt, int wn, int wk, int cp, bool rope) {
    if (rope) {
#define QL_PICK(MT_, WN_, WK_, CP_, DB_) \
        if (mt == MT_ && wn == WN_ && wk == WK_ && cp == CP_) \
            return (kern_t)qgemm_kernel<ACT_BF16, FP8W, MT_, WN_, WK_, CP_, DB_, GROUP, true, false, false, false, true>;
        QL_TILE_LIST_SK_ROPE(QL_PICK)
#undef QL_PICK
        return nullptr;
    }
#define QL_PICK(MT_, WN_, WK_, CP_, DB_) \
    if (mt == MT_ && wn == WN_ && wk == WK_ && cp == CP_) \
        return (kern_t)qgemm_kernel<ACT_BF16, FP8W, MT_, WN_, WK_, CP_, DB_, GROUP, false, false, false, false, true>;
    QL_TILE_LIST_SK(QL_PICK)
    if constexpr (!GROUP) { QL_TILE_LIST_SK_TALL(QL_PICK) }
#undef QL_PICK
    return nullptr;
}

// the W8A8 kernels with the activation quantization in their staging (x arrives as bf16 + partial row maxima)
template <int ACT, bool FP8W>
kern_t pick_dynq_impl(int mt, int wn, int wk, int cp) {
#define QL_PICK(MT_, WN_, WK_, CP_, DB_) \
    if (mt == MT_ && wn == WN_ && wk == WK_ && cp == CP_) \
        return (kern_t)qgemm_kernel<ACT, FP8W, MT_, WN_, WK_, CP_, DB_, false, false, false, false, true>;
    QL_TILE_LIST(QL_PICK)
    QL_TILE_LIST_TALL(QL_PICK)
#undef QL_PICK
    return nullptr;
}

// host side, defined in ql_gemm.hip (see there)
int w8a16_dispatch(const char* fn, int wdtype, int granularity, const void* x, const void* qw, const float* sw,
                   const void* bias, void* y, int64_t M, int64_t N, int64_t K, int group, void* workspace,
                   size_t workspace_bytes, void* stream);
size_t w8a16_workspace_need(int64_t M, int64_t N, int64_t K, int granularity, size_t* ticket_bytes);

kern_t pick_bf16_i8_row(int, int, int, int, bool rope = false);
kern_t pick_bf16_i8_group(int, int, int, int, bool rope = false);
kern_t pick_bf16_fp8_row(int, int, int, int, bool rope = false);
kern_t pick_bf16_fp8_group(int, int, int, int, bool rope = false);
kern_t pick_i8_i8(int, int, int, int, bool rope = false);
kern_t pick_fp8_fp8(int, int, int, int, bool rope = false);
kern_t pick_head(int, int, int, int, bool amax);
kern_t pick_i8_i8_dynq(int, int, int, int);
kern_t pick_sk_bf16_i8_row(int, int, int, int, bool rope);
kern_t pick_sk_bf16_i8_group(int, int, int, int, bool rope);
kern_t pick_sk_bf16_fp8_row(int, int, int, int, bool rope);
kern_t pick_sk_bf16_fp8_group(int, int, int, int, bool rope);
kern_t pick_fp8_fp8_dynq(int, int, int, int);

}  // namespace ql
