// EXPERIMENT, not built into the library (measured slower than the strip-per-wave kernels: 47-53 us against 38 us for
// Qwen3-4B gate/up at 256 rows; bound by what one CU can take in from its L2).  To try it: copy next to the library's
// sources, declare wide_silu_eligible / wide_silu_launch in ql_gemm_kernel.h and call them from gemm_w8a16.
//
// gate/up projection + SiLU * mul at 129..256 rows on the larger models' shapes:  y[M, N/2] = silu(x Wg^T) * (x Wu^T)
//
// The strip-per-wave kernels of ql_gemm_kernel.h quantise badly here: every wave owns 32 columns and ALL rows, so a
// workgroup's tile is 128 x 256 at best and N = 19456 (Qwen3-4B) gives 152 tiles for 256 CUs, while their smaller tiles
// pay the int8 -> bf16 widening once per 32 or 64 rows and are bound by vector issue.  This kernel turns the tile
// around:
//   * a workgroup owns ALL rows (up to 256: one wave per 32 rows, 5..8 waves) and a few 16-column units of the
//     weight, U per workgroup with U = ceil(units / 256), so the grid is one round of equal workgroups (N = 19456:
//     1216 units, U = 5, 244 workgroups = 95 % of the chip);
//   * the weight tile of a 128-k chunk (16 U columns) is loaded and widened ONCE per workgroup, cooperatively, and
//     parked in LDS as bf16; every wave reads its B fragments from there (v_mfma_f32_16x16x32_bf16, B = 16 columns
//     x 32 k), so the widening is amortised over all rows and the main loop is MFMA-bound, not VALU-bound;
//   * a wave's A fragments (its own 32 rows, nobody else's) go L2 -> VGPR directly, one chunk ahead;
//   * a unit is 8 gate columns + 8 up columns of the same outputs (gathered from the interleaved 32-row gate / up
//     strips of the SiLU weight layout), so gate and up of an output sit 8 lanes apart in one accumulator and the
//     epilogue needs one lane exchange; outputs are rounded exactly like the GEMM + ql_silu_mul pair.
// Per-row or per-tensor scales (applied to the f32 sums) or 128-k groups (a chunk is one group: a group accumulator
// started with a zero C operand, folded into the running sum with the group's scale).
#include "ql_gemm_kernel.h"

#ifndef WIDE_STAGGER
#define WIDE_STAGGER 1
#endif
#ifndef WIDE_ORDER
#define WIDE_ORDER 0
#endif
#ifndef WIDE_KO
#define WIDE_KO 0   // development knock-outs (bits): 1 no A requests, 2 no weight requests, 4 no MFMAs, 8 no B fragment reads, 16 no barrier, 32 no LDS stores
#endif

namespace ql {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WideParams {
    const unsigned char* x;      // [M, K] bf16
    const unsigned char* w;      // [N, K] one-byte weights, rows in gate/up strips of 32; row-major or tiled (ql_tile_weight)
    const float* sw;             // [N] | [1] | [N, K / 128]
    unsigned short* y;           // [M, N / 2] bf16
    int M, N, K;
    int sw_stride;               // 1 per-row, 0 per-tensor (unused with groups)
    int wtiled;
    int units;                   // N / 16
};

typedef void (*kern_wide_t)(const WideParams);

constexpr int kWideChunk = 128;                 // k per chunk (= the group size of the per-group form)
constexpr int kWideColBytes = 64;               // LDS bytes per (k step, column): 32 bf16 in four 16-byte slots
// Slot of a column's k quarter q (8 k = 16 bytes): q ^ F[column / 4].  ds_read_b128 serves a wave in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32); with the B fragment layout (lane = column + 16 q) this choice of F
// gives every group 16 distinct (column mod 4, slot) pairs, i.e. all 64 banks once.
__device__ __forceinline__ int wide_slot(int col, int q) { return q ^ ((0x1320 >> (4 * ((col >> 2) & 3))) & 3); }

template <bool FP8W, bool GROUP, int U, int D>
__global__ __launch_bounds__(512) void wide_silu_kernel(const WideParams p) {
    constexpr int NPC = U * 128;                                  // 16-byte weight pieces per chunk
    constexpr int NPB = (NPC + 319) / 320;                        // pieces per thread at the smallest workgroup (5 waves)
    constexpr int BUF = 4 * 16 * U * kWideColBytes;               // one chunk's bf16 weight tile
    __shared__ __attribute__((aligned(128))) unsigned char smem[2 * BUF];

    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int unit0 = blockIdx.x * U;
    const int ksteps = p.K >> 5;
    const int nch = p.K / kWideChunk;
    // Every workgroup reads ALL of x, chunk by chunk; walking the chunks in the same order they would all ask the L2
    // for the same few lines at the same time.  Workgroup g starts at chunk (5 g) mod nch and wraps around: at any
    // moment the chip's requests spread over the whole of x (the sum over k is the same set of terms in a rotated,
    // still fixed, order).
    const int ch_start = WIDE_STAGGER ? (int)((blockIdx.x * 5u) % (unsigned)nch) : 0;
    auto phys = [&](int ch) -> int { const int c = ch + ch_start; return c >= nch ? c - nch : c; };

    // ---- weight pieces of this thread: piece q = (unit, k step, segment, column) ----
    // segment 0 / 1: the gate strip's first / second 16 k of the step, 2 / 3: the up strip's
    // (threads past the last piece repeat it: same bytes to the same LDS address, and no branch around a load - the
    // wait-counter bookkeeping of the main loop stays exact only in straight-line code)
    unsigned woff[NPB];
    int lds_w[NPB], lds_w2[NPB];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        const int q = min(tid + j * T, NPC - 1);
        const int c8 = q & 7, seg = (q >> 3) & 3, ks = (q >> 5) & 3, u = min(q >> 7, U - 1);
        const int gu = min(unit0 + u, p.units - 1);               // (a unit past the last one re-reads the last: never stored)
        const int strip = 2 * (gu >> 2) + (seg >> 1), col = 8 * (gu & 3) + c8;
        if (p.wtiled) woff[j] = ((unsigned)strip * (unsigned)ksteps + (unsigned)ks) * 1024u + (unsigned)(col + 32 * (seg & 1)) * 16u;
        else woff[j] = ((unsigned)strip * 32u + (unsigned)col) * (unsigned)p.K + (unsigned)(ks * 32 + 16 * (seg & 1));
        // (the piece's 16 k are the quarters 2 (seg & 1) and 2 (seg & 1) + 1 of its column)
        const int lc = (seg >> 1) * 8 + c8;
        lds_w[j] = (ks * 16 * U + u * 16 + lc) * kWideColBytes + 16 * wide_slot(lc, 2 * (seg & 1));
        lds_w2[j] = (ks * 16 * U + u * 16 + lc) * kWideColBytes + 16 * wide_slot(lc, 2 * (seg & 1) + 1);
    }
    const unsigned wstep = p.wtiled ? 4u * 1024u : (unsigned)kWideChunk;     // bytes per chunk along a piece's k
    auto load_b = [&](int ch, i32x4* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPB; ++j)
            if (WIDE_KO & 2) dst[j] = i32x4{ch, j, 3, 4};
            else dst[j] = __builtin_nontemporal_load((const __attribute__((address_space(1))) i32x4*)(p.w + woff[j] + (unsigned)ch * wstep));
    };
    auto store_b = [&](int buf, const i32x4* src) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const bf16x8 lo = widen8<FP8W>(src[j][0], src[j][1]), hi = widen8<FP8W>(src[j][2], src[j][3]);
            *(bf16x8*)(smem + buf * BUF + lds_w[j]) = lo;
            *(bf16x8*)(smem + buf * BUF + lds_w2[j]) = hi;
        }
    };

    // ---- A fragments: rows 32 wave + 16 rf + l15, k = 32 ks + 8 lq .. + 7 of the chunk ----
    unsigned xoff[2];
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) xoff[rf] = ((unsigned)min(32 * wave + 16 * rf + l15, p.M - 1) * (unsigned)p.K + (unsigned)(8 * lq)) * 2u;
    auto load_a = [&](int ch, i32x4* dst) __attribute__((always_inline)) {
        const unsigned char* xs = p.x + (size_t)ch * (kWideChunk * 2);
#pragma unroll
        for (int rf = 0; rf < 2; ++rf)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (WIDE_KO & 1) dst[rf * 4 + ks] = i32x4{ch, ks, rf, 1};
                else dst[rf * 4 + ks] = gload16(xs + xoff[rf] + ks * 64);
            }
    };

    // ---- scales of this lane's column in every unit (per-group: one per chunk, requested a chunk ahead) ----
    int wrow[U];
#pragma unroll
    for (int f = 0; f < U; ++f) {
        const int gu = min(unit0 + f, p.units - 1);
        wrow[f] = (2 * (gu >> 2) + (l15 >> 3)) * 32 + 8 * (gu & 3) + (l15 & 7);
    }
    auto load_s = [&](int ch, float* dst) __attribute__((always_inline)) {
        if constexpr (GROUP) {
#pragma unroll
            for (int f = 0; f < U; ++f) dst[f] = p.sw[(size_t)wrow[f] * nch + ch];
        }
    };

    f32x4 acc[2][U];
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
#pragma unroll
        for (int f = 0; f < U; ++f) acc[rf][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int b_rd = l15 * kWideColBytes + 16 * wide_slot(l15, lq);               // this lane's B fragment inside a (k step, unit) block
    auto compute = [&](int buf, const i32x4* a, const float* s) __attribute__((always_inline)) {
        f32x4 g[2][U];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 b[U];
#pragma unroll
            for (int f = 0; f < U; ++f) {
                if (WIDE_KO & 8) b[f] = __builtin_bit_cast(bf16x8, a[ks]);
                else b[f] = *(const bf16x8*)(smem + buf * BUF + (ks * 16 * U + f * 16) * kWideColBytes + b_rd);
            }
#pragma unroll
            for (int rf = 0; rf < 2; ++rf) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, a[rf * 4 + ks]);
#pragma unroll
                for (int f = 0; f < U; ++f) {
                    if constexpr (GROUP) {
                        if (ks == 0) g[rf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[f], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else g[rf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[f], g[rf][f], 0, 0, 0);
                    } else {
                        if (WIDE_KO & 4) acc[rf][f][ks] += (float)av[0] + (float)b[f][0];
                        else acc[rf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[f], acc[rf][f], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (GROUP) {
#pragma unroll
            for (int rf = 0; rf < 2; ++rf)
#pragma unroll
                for (int f = 0; f < U; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[rf][f][i] = __builtin_fmaf(g[rf][f][i], s[f], acc[rf][f][i]);
        }
    };

    // ---- pipeline ----
    // The weight stream needs depth: a CU has one workgroup, and what it keeps in flight divided by the HBM latency
    // is its share of the bandwidth (two chunks = 20 KB per CU measured 1.4 TB/s over the chip).  Chunk c's raw
    // weights live in register set c % D of a ring of D sets, requested D - 1 chunks before the end of the iteration
    // that parks them in LDS; the loop is unrolled D times (D even, K a multiple of D chunks: host), so set indices
    // are compile-time and nothing moves between sets.  Within an iteration the requests go out as A(ch + 1),
    // scales(ch + 1), THEN weights(ch + D - 1): the wait counter is positional, so waiting for A(ch + 1) at the start of
    // the next iteration leaves every weight request behind it in flight.
    i32x4 a0[8], a1[8], braw[D][NPB];
    float sg0[U], sg1[U];
    load_b(phys(0), braw[0]);
    store_b(0, braw[0]);
    load_a(phys(0), a0);
    load_s(phys(0), sg0);
#pragma unroll
    for (int j = 1; j < D - 1; ++j) load_b(phys(min(j, nch - 1)), braw[j]);
    __syncthreads();
    auto chunk = [&](auto jc, int ch) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value, PAR = J & 1;       // ch % D == J
        i32x4* acur = PAR ? a1 : a0;
        i32x4* anext = PAR ? a0 : a1;
        // (past the end the last chunk is requested again and parked in the buffer nobody reads any more: a branch
        // around a request would make every wait behind it assume the worst of both paths)
        load_a(phys(min(ch + 1, nch - 1)), anext);
        load_s(phys(min(ch + 1, nch - 1)), PAR ? sg0 : sg1);
        load_b(phys(min(ch + D - 1, nch - 1)), braw[(J + D - 1) % D]);  // (the set of chunk ch - 1, in LDS since the last iteration)
        // (the requests stay HERE: left alone, the scheduler sinks them behind the MFMAs to shorten their registers'
        // live ranges, i.e. to the end of the iteration, and the next iteration starts by waiting a full memory
        // latency for them)
        __builtin_amdgcn_sched_barrier(0);
        // chunk ch + 1 goes to the other LDS buffer (free since the barrier that ended the last iteration) in front of
        // this chunk's MFMAs, so that its widening and LDS stores can issue between them
        if (WIDE_ORDER && !(WIDE_KO & 32)) store_b(PAR ^ 1, braw[(J + 1) % D]);
        compute(PAR, acur, PAR ? sg1 : sg0);
        if (!WIDE_ORDER && !(WIDE_KO & 32)) { __builtin_amdgcn_sched_barrier(0); store_b(PAR ^ 1, braw[(J + 1) % D]); }
        if (!(WIDE_KO & 16)) __syncthreads();
    };
    auto round = [&](auto self, auto jc, int base) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        if constexpr (J < D) {
            chunk(jc, base + J);
            self(self, std::integral_constant<int, J + 1>{}, base);
        }
    };
    for (int base = 0; base < nch; base += D) round(round, std::integral_constant<int, 0>{}, base);

    // ---- epilogue: scale, round to bf16 as the plain GEMM would, gate/up exchange 8 lanes apart, store ----
    const unsigned I = (unsigned)p.N >> 1;
#pragma unroll
    for (int f = 0; f < U; ++f) {
        const int gu = unit0 + f;
        if (gu >= p.units) break;                                 // (uniform)
        float s = 1.f;
        if constexpr (!GROUP) s = p.sw[(size_t)wrow[f] * p.sw_stride];
        const unsigned ocol = 32u * (unsigned)(gu >> 2) + 8u * (unsigned)(gu & 3) + (unsigned)(l15 & 7);
#pragma unroll
        for (int rf = 0; rf < 2; ++rf)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float own = (float)(bf16_t)__fmul_rn(acc[rf][f][i], s);
                const float oth = __shfl_xor(own, 8);
                const int row = 32 * wave + 16 * rf + 4 * lq + i;
                if (l15 < 8 && row < p.M) p.y[(size_t)row * I + ocol] = silu_mul_bf16(own, oth);
            }
    }
}

// ring depth for a K of nch chunks: the deepest built one that divides it
static int wide_depth(int64_t nch) { return nch % 8 == 0 ? 8 : nch % 10 == 0 ? 10 : nch % 4 == 0 ? 4 : nch % 6 == 0 ? 6 : 0; }

template <bool FP8W, bool GROUP, int D>
static kern_wide_t pick_wide_u(int U) {
    switch (U) {
        case 2: return (kern_wide_t)wide_silu_kernel<FP8W, GROUP, 2, D>;
        case 3: return (kern_wide_t)wide_silu_kernel<FP8W, GROUP, 3, D>;
        case 4: return (kern_wide_t)wide_silu_kernel<FP8W, GROUP, 4, D>;
        case 5: return (kern_wide_t)wide_silu_kernel<FP8W, GROUP, 5, D>;
        case 6: return (kern_wide_t)wide_silu_kernel<FP8W, GROUP, 6, D>;
    }
    return nullptr;
}
template <bool FP8W, bool GROUP>
static kern_wide_t pick_wide(int U, int D) {
    switch (D) {
        case 4: return pick_wide_u<FP8W, GROUP, 4>(U);
        case 6: return pick_wide_u<FP8W, GROUP, 6>(U);
        case 8: return pick_wide_u<FP8W, GROUP, 8>(U);
        case 10: return pick_wide_u<FP8W, GROUP, 10>(U);
    }
    return nullptr;
}

bool wide_silu_eligible(int64_t M, int64_t N, int64_t K, bool group, int group_size, bool has_bias) {
    if (has_bias || M < 129 || M > 256 || N % 64 || K % kWideChunk || (group && group_size != kWideChunk)) return false;
    const int64_t units = N / 16, U = (units + 255) / 256;
    // (32-bit byte offsets into the operands)
    if ((uint64_t)N * (uint64_t)K >= (1ull << 32) || (uint64_t)M * (uint64_t)K * 2 >= (1ull << 32)) return false;
    return U >= 2 && U <= 6 && wide_depth(K / kWideChunk) != 0;
}

int wide_silu_launch(const char* fn, bool fp8w, bool group, bool tiled, const void* x, const void* qw, const float* sw,
                     int sw_stride, void* y, int64_t M, int64_t N, int64_t K, hipStream_t stream) {
    WideParams p{};
    p.x = (const unsigned char*)x; p.w = (const unsigned char*)qw; p.sw = sw; p.y = (unsigned short*)y;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.sw_stride = sw_stride; p.wtiled = tiled ? 1 : 0;
    p.units = (int)(N / 16);
    const int U = (p.units + 255) / 256;
    const int D = wide_depth(K / kWideChunk);
    const kern_wide_t k = fp8w ? (group ? pick_wide<true, true>(U, D) : pick_wide<true, false>(U, D))
                               : (group ? pick_wide<false, true>(U, D) : pick_wide<false, false>(U, D));
    if (!k) { set_error("%s: no wide tile for %d units per workgroup", fn, U); return QL_EINVAL; }
    const unsigned grid = (unsigned)((p.units + U - 1) / U), threads = 64u * (unsigned)((M + 31) / 32);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, stream, p);
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

}  // namespace ql
