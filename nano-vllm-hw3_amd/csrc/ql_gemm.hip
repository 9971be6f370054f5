// Host-side dispatch of the fused dequant GEMMs: tile selection, split-K planning, argument checks,
// and the extern "C" entry points.  The kernels live in ql_gemm_kernel.h / ql_gemm_inst_*.hip.
#include "ql_gemm_kernel.h"
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <tuple>
#include <vector>

namespace ql {

static const TileCfg kTiles[] = {
#define QL_ROW(MT_, WN_, WK_, CP_, DB_) {MT_, WN_, WK_, CP_, DB_},
    QL_TILE_LIST(QL_ROW) QL_TILE_LIST_TALL(QL_ROW)
#undef QL_ROW
};
static thread_local int f_mt = 0, f_wn = 0, f_wk = 0, f_cp = 0, f_split = 0;

static bool tile_built(int mt, int wn, int wk, int cp, bool group) {
    for (const TileCfg& t : kTiles)
        if (t.mt == mt && t.wn == wn && t.wk == wk && t.cp == cp) return !(group && mt == 8);
    return false;
}

static bool rope_tile_built(int mt, int wn, int wk, int cp) {
#define QL_ROW(MT_, WN_, WK_, CP_, DB_) if (mt == MT_ && wn == WN_ && wk == WK_ && cp == CP_) return true;
    QL_TILE_LIST_ROPE(QL_ROW)
#undef QL_ROW
    return false;
}

struct Plan {
    int mt, wn, wk, cp, split, k_per_split;
    dim3 grid;
    bool pair_rows = false;
    // stream-K schedule (sk_G > 0; `split` is then reported as -sk_G): workgroups, row tiles walked side by side,
    // k chunks per output tile, column / row tiles of the problem
    int sk_G = 0, sk_R = 1, sk_C = 0, sk_nb = 0, sk_mb = 0;
};

// tiles built with the stream-K schedule (QL_TILE_LIST_SK*, ql_gemm_kernel.h)
static bool sk_tile_built(int mt, int wn, int wk, int cp, bool group, bool rope) {
    if (rope) {
#define QL_ROW(MT_, WN_, WK_, CP_, DB_) if (mt == MT_ && wn == WN_ && wk == WK_ && cp == CP_) return true;
        QL_TILE_LIST_SK_ROPE(QL_ROW)
        return false;
    }
    QL_TILE_LIST_SK(QL_ROW)
    if (!group) { QL_TILE_LIST_SK_TALL(QL_ROW) }
#undef QL_ROW
    return false;
}

// GemmParams::gmagic for a group size (a multiple of 32): with d = group / 32 and m = ceil(2^32 / d),
// umulhi(q, m) == q / d for all q < 2^32 / d, far beyond any K / 32 that passes check_common
// A/B switch for the per-group kernels: QL_GROUP_GENERIC=1 keeps them off the group-aligned form of their main loop
// (identical results; read once, not per launch)
static int group_generic() {
    static const int v = getenv("QL_GROUP_GENERIC") != nullptr ? 1 : 0;
    return v;
}

static unsigned group_magic(int group) {
    const unsigned d = (unsigned)group / 32u;
    return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d);
}

// ---- plan table -------------------------------------------------------------------------------------------------
// Tile and split choices MEASURED on the device for given shapes (tools/autotune_plans.py) take precedence over the
// heuristics below, which were tuned on the Qwen3-0.6B layer shapes and leave 10-30 % on larger ones.  An entry
// covers one launch form (QL_PLAN_* bits of include/qlinear.h), one (N, K) and a range of row counts.
struct PlanEntry { int64_t m_lo, m_hi; int mt, wn, wk, cp, split; };
static std::map<std::tuple<int, int64_t, int64_t>, std::vector<PlanEntry>> g_plans;
static std::shared_mutex g_plans_mu;

// set while the allocation-free entry point plans a call: its fixed scratch array is sized for the heuristics' splits
static thread_local bool tl_skip_table = false;
struct SkipTable {
    bool prev;
    SkipTable() : prev(tl_skip_table) { tl_skip_table = true; }
    ~SkipTable() { tl_skip_table = prev; }
};

static bool plan_lookup(int form, int64_t M, int64_t N, int64_t K, PlanEntry* out) {
    if (tl_skip_table) return false;
    std::shared_lock<std::shared_mutex> lock(g_plans_mu);
    if (g_plans.empty()) return false;
    const auto it = g_plans.find(std::make_tuple(form, N, K));
    if (it == g_plans.end()) return false;
    for (const PlanEntry& e : it->second)
        if (M >= e.m_lo && M <= e.m_hi) { *out = e; return true; }
    return false;
}

static Plan make_plan(int64_t M, int64_t N, int64_t K, bool group, bool silu = false, bool rope = false, bool w8a8 = false) {
    Plan pl;
    // Tile ladder, smallest to largest output tile.  On a 256-CU chip these GEMMs are bound by
    // instruction issue and L2->LDS traffic per workgroup, not by HBM or MFMA peak, so the measured
    // optimum (tools/bench_kernels.py --sweep) is the LARGEST tile that still yields about one
    // workgroup per CU: bigger tiles re-read less of x and w through L2 and amortise the int8->bf16
    // widening over more rows, smaller tiles only win while the grid would otherwise leave CUs idle.
    static const TileCfg ladder[] = {
        {1, 1, 8, 4, 0},    //  32 x  32, 8 waves split K, x resident in LDS
        {1, 2, 4, 8, 0},    //  32 x  64
        {2, 2, 4, 8, 0},    //  64 x  64
        {2, 4, 2, 4, 1},    //  64 x 128
    };
    const int nladder = (int)(sizeof(ladder) / sizeof(ladder[0]));
    int pick = silu ? 1 : 0;                               // the fused SiLU epilogue pairs two column strips (WN >= 2)
    for (int i = pick; i < nladder; ++i) {
        const TileCfg& t = ladder[i];
        if (group && t.mt == 8) break;
        const int64_t blocks = ((M + 32 * t.mt - 1) / (32 * t.mt)) * ((N + 32 * t.wn - 1) / (32 * t.wn));
        if (blocks >= (i == nladder - 1 ? 144 : 160)) pick = i;       // (the 64 x 128 tile pays off from 144 tiles)
    }
    // at most 32 rows: a taller tile would only re-read the clamped last row, and the 32-column tile keeps
    // the most weight bytes in flight per CU (measured on the 0.6B and 4B layer shapes)
    if (M <= 32 && !silu) pick = 0;
    int mt = ladder[pick].mt, wn = ladder[pick].wn, wk = ladder[pick].wk, cp = ladder[pick].cp;
    // K = 1024 and one round of 32 x 128 tiles (176..256 of them): eight waves, the whole K resident in one
    // chunk, every weight load of the workgroup in flight at once (measured 0.6-0.85 us faster than the ladder's
    // pick on qkv at M = 192..256 and gate_up at M = 128 of Qwen3-0.6B)
    if (K == 1024 && M > 32 && M <= 1024) {
        const int64_t b128 = ((M + 31) / 32) * ((N + 127) / 128);
        if (b128 >= 176 && b128 <= 256) { mt = 1; wn = 4; wk = 2; cp = 16; }
    }
    // the same on one round of 64 x 128 tiles (gate/up at 161..320 rows: -4..8 % against four 256-k chunks; with
    // more tiles than CUs the chunked form wins again, -22 % at 512 rows)
    if (pick == nladder - 1 && K == 1024 && ((M + 63) / 64) * ((N + 127) / 128) <= 256) cp = 16;
    // K up to 2048 on the 32 x 32 tile: one 2048-k chunk (the whole row of activations resident, no split, no
    // second staging round) measured 0.6 us faster than two 1024-k chunks or slices
    if (pick == 0 && K > 1024 && K <= 2048) cp = 8;
    // prefill-sized problems: thousands of 64 x 128 tiles; four waves each running the whole K with
    // the double-buffered pipeline measured best (the taller tiles lose occupancy to registers)
    if (M > 1024) {
        mt = 2; wn = 4; wk = 1; cp = 4;
        // with enough 128 x 256 tiles to fill the chip, the wide 8-wave tile halves the L2->LDS bytes
        // per MFMA (measured +25 % at M = 16384); 128-k chunks give its loads a whole chunk of MFMAs to land.
        // Per-group scales keep a second accumulator set, which fits beside 64-k chunks only (measured -24 % at
        // M = 16384 against the 64 x 128 tile, and ahead of the 128-k form on every shape).
        const int64_t wide = ((M + 127) / 128) * ((N + 255) / 256);
        if (wide >= 256) { mt = 4; wn = 8; wk = 1; cp = group ? 2 : 4; }
        // Two rounds of those tiles or more and K up to 2048 (bf16 activations, per-row / per-tensor scales): the
        // 64-k-chunk form of the tile, built to 128 registers so that two workgroups share a CU and one's prologue
        // and epilogue overlap the other's main loop (measured at M = 16384: qkv +14 %, gate_up +8 %, o_proj +4 %;
        // down_proj, K = 3072, keeps the 128-k chunks with the two-chunk-deep prefetch: -3 % otherwise)
        if (wide >= 512 && K <= 2048 && !group && !w8a8) cp = 2;
    }
    if (rope) {
        // the q/k norm + RoPE epilogue needs whole 128-column heads in one workgroup (4 or 8 column strips).
        // Decode sizes: the 32 x 128 eight-wave tile with the whole K (up to 1024) of a workgroup in flight at once;
        // from two rounds of those on, 64 x 128.  Prefill sizes keep the tiles chosen above (128 or 256 columns).
        if (M <= 1024) {
            const int64_t b32 = ((M + 31) / 32) * ((N + 127) / 128);
            if (b32 > 512) { mt = 2; wn = 4; wk = 2; cp = 4; }
            else { mt = 1; wn = 4; wk = 2; cp = 16; }
            pick = -1;
        }
    }
    // a measured plan for this shape, if the table holds one (entries were checked when they were set)
    int t_split = 0;
    PlanEntry te;
    if (plan_lookup((group ? QL_PLAN_GROUP : 0) | (silu ? QL_PLAN_SILU : 0) | (rope ? QL_PLAN_ROPE : 0) | (w8a8 ? QL_PLAN_W8A8 : 0),
                    M, N, K, &te)) {
        mt = te.mt; wn = te.wn; wk = te.wk; cp = te.cp; t_split = te.split;
    }
    if (f_mt > 0 && f_wn > 0 && f_wk > 0 && f_cp > 0 && tile_built(f_mt, f_wn, f_wk, f_cp, group) && !(silu && (f_wn & 1)) &&
        !(rope && !rope_tile_built(f_mt, f_wn, f_wk, f_cp))) {
        mt = f_mt; wn = f_wn; wk = f_wk; cp = f_cp; t_split = 0;
    }
    pl.mt = mt; pl.wn = wn; pl.wk = wk; pl.cp = cp;
    const int kb = 32 * cp * wk;
    const int64_t mblocks = (M + 32 * mt - 1) / (32 * mt);
    const int64_t nblocks = (N + 32 * wn - 1) / (32 * wn);
    const int64_t kchunks = (K + kb - 1) / kb;
    const int64_t base = mblocks * nblocks;
    // fewer tiles than CUs even at 32 x 32 (N = hidden size): split K so that each slice is one 1024-k
    // chunk, combined in-kernel by the last arriver (measured: down_proj 11.1 -> 9.5 us at M = 64..128)
    // (K >= 8192 is cut into 2048-k slices instead: two chunks per slice halve the slab traffic, measured on the
    // Qwen3-4B down_proj shape; a grid of 144+ tiles of the 64 x 128 tile is not split at all)
    int64_t split = 1;
    if (base < 144) split = K >= 8192 ? (K + 2047) / 2048 : (K + 1023) / 1024;
    else if (base < 160 && pick < nladder - 1) split = (K + 1023) / 1024;
    else if (base < 512 && K >= 4096) split = (K + 4095) / 4096;   // long K: at most four chunks per slice
    // (gate/up with the SiLU epilogue: gate and up meet in the tile's LAST workgroup, which holds the whole reduced
    // tile; the heuristics were measured without slices there and keep one, a table entry may ask for more)
    if (silu) split = 1;
    if (t_split != 0) split = t_split;
    if (f_split != 0) split = f_split;
    if (split < 0) {
        // stream-K: -split workgroups share the (tile, chunk) units evenly (see the kernel)
        // (never more workgroups than units: a workgroup without units would be counted among a tile's
        // contributors and never arrive)
        const int64_t G = std::min<int64_t>(-split, base * kchunks);
        static const bool no_pair = getenv("QL_SK_NOPAIR") != nullptr;
        // (the kernel divides by multiply-high: units x chunks per tile and units x units per workgroup below 2^32)
        const int64_t U = base * kchunks;
        if (sk_tile_built(mt, wn, wk, cp, group, rope) && !w8a8 && base <= kMaxTiles && G >= 1 && U * kchunks < (1ll << 31) &&
            U * (U / G + 2) < (1ll << 31)) {
            pl.split = (int)-G;
            pl.k_per_split = (int)(kchunks * kb);
            pl.sk_G = (int)G;
            pl.sk_R = (!no_pair && mblocks > 1 && mblocks <= 8 && G % (8 * mblocks) == 0 && G / mblocks <= nblocks * kchunks)
                          ? (int)mblocks : 1;
            pl.sk_C = (int)kchunks;
            pl.sk_nb = (int)nblocks;
            pl.sk_mb = (int)mblocks;
            pl.grid = dim3((unsigned)G, 1, 1);
            return pl;
        }
        split = 1;
    }
    if (split > 64) split = 64;
    if (split > kchunks) split = kchunks;
    if (base > kMaxTiles) split = 1;                        // one ticket counter per output tile
    if (split < 1) split = 1;
    const int64_t cps = (kchunks + split - 1) / split;      // chunks per slice
    split = (kchunks + cps - 1) / cps;
    pl.split = (int)split;
    pl.k_per_split = (int)(cps * kb);
    pl.grid = dim3((unsigned)nblocks, (unsigned)mblocks, (unsigned)split);
    return pl;
}

// Plan of the deferred-reduction form (ql_gemm_w8a16_partial): K is cut into as many slices as it takes to give the
// chip about one workgroup per CU; nothing is combined in the kernel, so slices cost no tail.
static Plan make_plan_partial(int64_t M, int64_t N, int64_t K, bool group) {
    Plan pl;
    // Measured on the Qwen3-0.6B o_proj / down_proj shapes (tools/bench_partial.py): one 1024-k chunk per
    // workgroup wherever possible.  The 32 x 32 tile (eight waves over the chunk, two workgroups per CU) is best
    // up to one workgroup per CU and again when there are too many tiles for anything else; in between, the
    // 32 x 64 tile (one workgroup per CU) halves the activation re-reads.
    const int64_t mb = (M + 31) / 32, kc = (K + 1023) / 1024;
    const int64_t base32 = mb * ((N + 31) / 32), base64 = mb * ((N + 63) / 64);
    int mt = 1, wn = 1, wk = 8, cp = 4;
    int64_t cap = 512;
    if (base32 * kc > 256 && base64 * kc <= 256) { wn = 2; wk = 4; cp = 8; cap = 256; }
    if (M > 256) { mt = 2; wn = 2; wk = 4; cp = 8; cap = 256; }          // 64 x 64
    int t_split = 0;
    PlanEntry te;
    if (plan_lookup(QL_PLAN_PARTIAL | (group ? QL_PLAN_GROUP : 0), M, N, K, &te)) {
        mt = te.mt; wn = te.wn; wk = te.wk; cp = te.cp; t_split = te.split;   // (one slice: the caller does not defer)
    }
    if (f_mt > 0 && f_wn > 0 && f_wk > 0 && f_cp > 0 && tile_built(f_mt, f_wn, f_wk, f_cp, group)) {
        mt = f_mt; wn = f_wn; wk = f_wk; cp = f_cp; t_split = 0;
    }
    pl.mt = mt; pl.wn = wn; pl.wk = wk; pl.cp = cp;
    const int kb = 32 * cp * wk;
    const int64_t mblocks = (M + 32 * mt - 1) / (32 * mt), nblocks = (N + 32 * wn - 1) / (32 * wn);
    const int64_t kchunks = (K + kb - 1) / kb, base = mblocks * nblocks;
    int64_t split = kchunks;                                 // one chunk per slice ...
    while (split > 1 && base * split > cap) --split;         // ... unless that is more than one round of workgroups
    // two slices of a grid that already has a workgroup per CU gain less than the consumer's extra reads cost
    // (measured): report one slice, and the caller keeps the self-contained GEMM
    if (split == 2 && base32 * kc > 256) split = 1;
    if (t_split > 0) split = t_split;
    if (f_split > 0) split = f_split;
    if (split > kchunks) split = kchunks;
    if (split > 16) split = 16;
    if (split < 1) split = 1;
    const int64_t cps = (kchunks + split - 1) / split;
    split = (kchunks + cps - 1) / cps;
    pl.split = (int)split;
    pl.k_per_split = (int)(cps * kb);
    pl.grid = dim3((unsigned)nblocks, (unsigned)mblocks, (unsigned)split);
    return pl;
}

static size_t plan_workspace(const Plan& pl) {
    if (pl.sk_G > 0)    // one slab per (workgroup, tile) pair that shares a tile: at most G + tiles of them
        return (size_t)kCounterBytes + ((size_t)pl.sk_G + (size_t)pl.sk_nb * pl.sk_mb) * (size_t)(pl.wn * pl.mt * 16 * 64 * 4);
    if (pl.split <= 1) return 0;
    const size_t tiles = (size_t)pl.grid.x * pl.grid.y;
    return (size_t)kCounterBytes + tiles * pl.split * (size_t)(pl.wn * pl.mt * 16 * 64 * 4);
}

static int check_common(const char* fn, int64_t M, int64_t N, int64_t K) {
    if (M < 0 || N <= 0 || K <= 0) { set_error("%s: bad shape M=%lld N=%lld K=%lld", fn, (long long)M, (long long)N, (long long)K); return QL_EINVAL; }
    if (K % 32) { set_error("%s: K=%lld must be a multiple of 32", fn, (long long)K); return QL_ESHAPE; }
    // the kernels address weights and activations with 32-bit byte offsets from a uniform base
    if (M > 0x3FFFFFFF || N > 0x3FFFFFFF || K > 0x3FFFFFFF || (M + 255) / 32 > 65535 ||
        (uint64_t)N * (uint64_t)K >= (1ull << 32) || (uint64_t)(M + 256) * (uint64_t)K * 2 >= (1ull << 32)) {
        set_error("%s: shape too large", fn);
        return QL_ESHAPE;
    }
    return QL_OK;
}

static int run(kern_t k, GemmParams& p, const Plan& pl, void* ws, size_t ws_bytes, hipStream_t s, const char* fn) {
    if (!k) { set_error("%s: tile (%d,%d,%d,%d) not built", fn, pl.mt, pl.wn, pl.wk, pl.cp); return QL_EINVAL; }
    p.k_per_split = pl.k_per_split;
    if (pl.sk_G > 0) {
        const size_t need = plan_workspace(pl);
        if (!ws || ws_bytes < need) {
            set_error("%s: workspace of %zu bytes needed for the stream-K schedule on %d workgroups, got %zu", fn, need, pl.sk_G, ws_bytes);
            return QL_EWORKSPACE;
        }
        p.counters = (unsigned*)ws;
        p.partial = (unsigned char*)ws + kCounterBytes;
        p.sk_G = pl.sk_G; p.sk_C = pl.sk_C; p.sk_R = pl.sk_R; p.sk_mblocks = pl.sk_mb;
        {
            // (make_plan checked that every dividend times its divisor stays below 2^32)
            auto magic = [](int64_t d) -> unsigned { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); };
            const int64_t Gr = pl.sk_G / pl.sk_R;
            const int64_t Ul = (pl.sk_R > 1 ? (int64_t)pl.sk_nb : (int64_t)pl.sk_nb * pl.sk_mb) * pl.sk_C;
            p.sk_Gr = (int)Gr; p.sk_per = (int)(Ul / Gr); p.sk_rem = (int)(Ul % Gr);
            p.sk_mC = magic(pl.sk_C); p.sk_mper = magic(p.sk_per); p.sk_mper1 = magic(p.sk_per + 1);
            p.sk_mmb = magic(pl.sk_mb); p.sk_mR = magic(pl.sk_R);
        }
        p.swz_gn = 0;
        p.ntile_n = pl.sk_nb;
        hipLaunchKernelGGL(k, pl.grid, dim3(64 * pl.wn * pl.wk), 0, s, p);
        QL_CHECK_LAUNCH(fn);
        return QL_OK;
    }
    if (pl.split > 1) {
        const size_t need = plan_workspace(pl);
        if (!ws || ws_bytes < need) {
            set_error("%s: workspace of %zu bytes needed for split-K=%d, got %zu", fn, need, pl.split, ws_bytes);
            return QL_EWORKSPACE;
        }
        p.counters = (unsigned*)ws;
        p.partial = (unsigned char*)ws + kCounterBytes;
    }
    // XCD-aware tile order for grids of many tiles per XCD (prefill): the XCD grouping along N that keeps an
    // XCD's weight slice within about half of its 4 MiB L2
    p.swz_gn = 0;
    p.ntile_n = (int)pl.grid.x;
    static const bool no_swizzle = getenv("QL_NO_XCD_SWIZZLE") != nullptr;    // read once, not per launch
    if (pl.split == 1 && pl.grid.x * pl.grid.y >= 512 && !no_swizzle) {
        const uint64_t wbytes = (uint64_t)p.N * p.K;
        for (int gn = 1; gn <= 8; gn *= 2) {
            if (pl.grid.x % gn || pl.grid.y % (8 / gn)) continue;
            p.swz_gn = gn;
            if (wbytes / gn <= (2u << 20)) break;
        }
    }
    dim3 grid = pl.grid;
    if (pl.pair_rows && pl.grid.y > 1 && pl.grid.y <= 8 && pl.split == 1) {
        p.swz_gn = -(int)pl.grid.y;
        grid = dim3((pl.grid.x + 7) / 8 * 8 * pl.grid.y, 1, 1);
    }
    hipLaunchKernelGGL(k, grid, dim3(64 * pl.wn * pl.wk), 0, s, p);
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

}  // namespace ql

using namespace ql;

namespace ql {
// Per-group W8A16 for group sizes that are not multiples of the MFMA kernels' 32-k step (8, 16, 24, 48, ...): a plain
// f32 kernel, 16 x 16 outputs per workgroup, operands through LDS 32 k at a time; a group's products are summed in k
// order and folded into the total with the group's f32 scale, as in the MFMA kernels.  Correctness path for a corner
// of the reference's interface (any divisor of K), not a tuned one.
template <bool FP8W>
__global__ __launch_bounds__(256) void qgemm_anygroup_kernel(const unsigned short* __restrict__ x, const unsigned char* __restrict__ w,
                                                             const float* __restrict__ sw, const unsigned short* __restrict__ bias,
                                                             unsigned short* __restrict__ y, int M, int N, int K, int group,
                                                             int tiled) {
    __shared__ float xs[16][33], ws[16][33];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
    const int ngk = K / group, ksteps = K >> 5;
    // staging: thread t moves x[m0 + t / 16][k0 + 2 (t % 16) .. + 1] and the same element pair of w's 16 rows
    const int sr = threadIdx.x >> 4, sk = (threadIdx.x & 15) * 2;
    const int xm = min((int)blockIdx.y * 16 + sr, M - 1), wn = min((int)blockIdx.x * 16 + sr, N - 1);
    float acc = 0.f, gsum = 0.f;
    int left = group, g = 0;
    const int nc = min(n, N - 1);
    float s = sw[(size_t)nc * ngk];
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k0 + sk + e;
            xs[sr][sk + e] = bf16_bits_to_f32(x[(size_t)xm * K + k]);
            const size_t off = tiled ? ((((size_t)(wn >> 5) * ksteps + (k >> 5)) * 64 + (wn & 31) + 32 * ((k >> 4) & 1)) * 16 + (k & 15))
                                     : (size_t)wn * K + k;
            const unsigned char b = w[off];
            ws[sr][sk + e] = FP8W ? e4m3fn_to_f32(b) : (float)(signed char)b;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < 32; ++kk) {
            gsum = __builtin_fmaf(xs[ty][kk], ws[tx][kk], gsum);
            if (--left == 0) {
                acc = __builtin_fmaf(gsum, s, acc);
                gsum = 0.f;
                left = group;
                g = min(g + 1, ngk - 1);
                s = sw[(size_t)nc * ngk + g];
            }
        }
        __syncthreads();
    }
    if (m < M && n < N) {
        // (the per-group epilogue of the MFMA kernels: round, then the bias)
        float v = acc;
        if (bias) v = __fadd_rn((float)(bf16_t)v, bf16_bits_to_f32(bias[n]));
        y[(size_t)m * N + n] = __builtin_bit_cast(unsigned short, (bf16_t)v);
    }
}
}  // namespace ql

using ql::qgemm_anygroup_kernel;

extern "C" {

size_t ql_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int granularity) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32) return 0;
    // (the qkv entry points with the RoPE epilogue may plan differently, and the W8A8 entry points may hold
    // plan-table entries of their own: cover them all)
    const bool grp = granularity == QL_PER_GROUP;
    size_t need = plan_workspace(make_plan(M, N, K, grp));
    const bool heads = N % 128 == 0;
    if (heads) need = std::max(need, plan_workspace(make_plan(M, N, K, grp, false, true)));
    const bool strips = N % 64 == 0;
    if (strips) need = std::max(need, plan_workspace(make_plan(M, N, K, grp, true)));
    if (!grp) {
        need = std::max(need, plan_workspace(make_plan(M, N, K, false, false, false, true)));
        if (heads) need = std::max(need, plan_workspace(make_plan(M, N, K, false, false, true, true)));
        if (strips) need = std::max(need, plan_workspace(make_plan(M, N, K, false, true, false, true)));
    }
    return need;
}

size_t ql_gemm_workspace_max_bytes(int64_t max_M, int64_t N, int64_t K, int granularity) {
    if (max_M <= 0 || N <= 0 || K <= 0 || K % 32) return 0;
    // every threshold of the tile / split heuristics sits on a multiple of 32 rows and the tiles are multiples
    // of 32 rows tall, so the plan is the same for all M of one 32-row block: its last M stands for the block
    size_t need = 0;
    for (int64_t m = 32;; m += 32) {
        const int64_t mm = m < max_M ? m : max_M;
        const size_t b = ql_gemm_workspace_bytes(mm, N, K, granularity);
        if (b > need) need = b;
        if (mm == max_M) break;
    }
    // ... of the heuristics; a table entry may begin or end anywhere: the ends of every entry for this (N, K) as well
    std::vector<int64_t> ends;
    {
        std::shared_lock<std::shared_mutex> lock(g_plans_mu);
        for (const auto& kv : g_plans)
            if (std::get<1>(kv.first) == N && std::get<2>(kv.first) == K)
                for (const PlanEntry& e : kv.second) { ends.push_back(e.m_lo); ends.push_back(e.m_hi); }
    }
    for (int64_t m : ends)
        if (m >= 1 && m <= max_M) need = std::max(need, ql_gemm_workspace_bytes(m, N, K, granularity));
    return need;
}

int ql_gemm_force_config(int mt, int wn, int wk, int cp, int split_k) {
    const bool any = mt > 0 || wn > 0 || wk > 0 || cp > 0;
    if (any && !tile_built(mt, wn, wk, cp, false) && !pick_head(mt, wn, wk, cp, false)) {
        set_error("ql_gemm_force_config: tile (mt=%d, wn=%d, wk=%d, cp=%d) not built", mt, wn, wk, cp);
        return QL_EINVAL;
    }
    f_mt = any ? mt : 0;
    f_wn = any ? wn : 0;
    f_wk = any ? wk : 0;
    f_cp = any ? cp : 0;
    f_split = split_k;                                       // (< 0: the stream-K schedule on -split_k workgroups)
    return QL_OK;
}

int ql_gemm_plan_set(int form, int64_t m_lo, int64_t m_hi, int64_t N, int64_t K, int mt, int wn, int wk, int cp, int split_k) {
    const char* fn = "ql_gemm_plan_set";
    const int known = QL_PLAN_GROUP | QL_PLAN_SILU | QL_PLAN_ROPE | QL_PLAN_W8A8 | QL_PLAN_PARTIAL;
    const bool group = (form & QL_PLAN_GROUP) != 0, silu = (form & QL_PLAN_SILU) != 0, rope = (form & QL_PLAN_ROPE) != 0;
    const bool partial = (form & QL_PLAN_PARTIAL) != 0;
    if (form < 0 || (form & ~known) || (silu && rope) || (partial && (silu || rope)) || ((form & QL_PLAN_W8A8) && group) ||
        m_lo < 1 || m_hi < m_lo || N <= 0 || K <= 0 || K % 32 || split_k == 0 || split_k > 64 || split_k < -4096 ||
        (split_k < 0 && (partial || (form & QL_PLAN_W8A8)))) {
        set_error("%s: bad argument (form=%d rows %lld..%lld N=%lld K=%lld split=%d)", fn, form, (long long)m_lo,
                  (long long)m_hi, (long long)N, (long long)K, split_k);
        return QL_EINVAL;
    }
    if (!tile_built(mt, wn, wk, cp, group) || (silu && (wn & 1)) || (rope && !rope_tile_built(mt, wn, wk, cp)) ||
        (rope && N % 128) || (split_k < 0 && !sk_tile_built(mt, wn, wk, cp, group, rope))) {
        set_error("%s: tile (mt=%d, wn=%d, wk=%d, cp=%d) with %d slices cannot run form %d", fn, mt, wn, wk, cp, split_k, form);
        return QL_EINVAL;
    }
    // (the self-contained split-K keeps one ticket per output tile of the smallest row count the entry covers)
    const int64_t tiles = ((m_hi + 32 * mt - 1) / (32 * mt)) * ((N + 32 * wn - 1) / (32 * wn));
    if (!partial && split_k > 1 && tiles > kMaxTiles) {
        set_error("%s: %lld output tiles exceed the %d split-K tickets", fn, (long long)tiles, kMaxTiles);
        return QL_EINVAL;
    }
    if (partial && split_k > 16) { set_error("%s: the deferred form takes at most 16 slices", fn); return QL_EINVAL; }
    std::unique_lock<std::shared_mutex> lock(g_plans_mu);
    std::vector<PlanEntry>& v = g_plans[std::make_tuple(form, N, K)];
    for (const PlanEntry& e : v)
        if (m_lo <= e.m_hi && e.m_lo <= m_hi) {
            set_error("%s: rows %lld..%lld overlap an entry of the same form and shape (%lld..%lld)", fn, (long long)m_lo,
                      (long long)m_hi, (long long)e.m_lo, (long long)e.m_hi);
            return QL_EINVAL;
        }
    v.push_back(PlanEntry{m_lo, m_hi, mt, wn, wk, cp, split_k});
    return QL_OK;
}

void ql_gemm_plan_clear(void) {
    std::unique_lock<std::shared_mutex> lock(g_plans_mu);
    g_plans.clear();
}

int ql_gemm_plan_count(void) {
    std::shared_lock<std::shared_mutex> lock(g_plans_mu);
    int n = 0;
    for (const auto& kv : g_plans) n += (int)kv.second.size();
    return n;
}

int ql_gemm_plan_query(int form, int64_t M, int64_t N, int64_t K, int* out) {
    const int known = QL_PLAN_GROUP | QL_PLAN_SILU | QL_PLAN_ROPE | QL_PLAN_W8A8 | QL_PLAN_PARTIAL;
    if (!out || form < 0 || (form & ~known) || M <= 0 || N <= 0 || K <= 0 || K % 32) {
        set_error("ql_gemm_plan_query: bad argument");
        return QL_EINVAL;
    }
    const bool group = (form & QL_PLAN_GROUP) != 0;
    const Plan pl = (form & QL_PLAN_PARTIAL) ? make_plan_partial(M, N, K, group)
                                             : make_plan(M, N, K, group, (form & QL_PLAN_SILU) != 0, (form & QL_PLAN_ROPE) != 0,
                                                         (form & QL_PLAN_W8A8) != 0);
    out[0] = pl.mt; out[1] = pl.wn; out[2] = pl.wk; out[3] = pl.cp; out[4] = pl.split;
    return QL_OK;
}

static int check_rope(const char* fn, const RopeArgs* r, int64_t M, int64_t N) {
    if (!r) return QL_OK;
    if (r->Hq <= 0 || r->Hkv <= 0 || (M > 0 && (!r->cos_sin || !r->q_out)) ||
        (r->slot_mapping && (!r->k_cache || !r->v_cache))) {
        set_error("%s: bad rope argument", fn);
        return QL_EINVAL;
    }
    if (N != (int64_t)(r->Hq + 2 * r->Hkv) * 128) {
        set_error("%s: N=%lld is not (Hq + 2 Hkv) * 128 = %lld", fn, (long long)N, (long long)(r->Hq + 2 * r->Hkv) * 128);
        return QL_ESHAPE;
    }
    return QL_OK;
}

static int gemm_w8a16(const char* fn, bool silu, int wdtype, int granularity, const void* x, const void* qw,
                      const float* sw, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int group,
                      void* workspace, size_t workspace_bytes, void* stream, const RopeArgs* rope = nullptr) {
    const bool tiled = (wdtype & QL_TILED) != 0;
    wdtype &= ~QL_TILED;
    if ((wdtype != QL_INT8 && wdtype != QL_FP8) || granularity < QL_PER_ROW || granularity > QL_PER_GROUP ||
        (M > 0 && (!x || (!y && !rope))) || !qw || !sw) {
        set_error("%s: bad argument (wdtype=%d granularity=%d)", fn, wdtype, granularity);
        return QL_EINVAL;
    }
    int rc = check_common(fn, M, N, K);
    if (rc) return rc;
    if (silu && N % 64) { set_error("%s: N=%lld must be a multiple of 64 (32-row gate/up strips)", fn, (long long)N); return QL_ESHAPE; }
    rc = check_rope(fn, rope, M, N);
    if (rc) return rc;
    const bool grp = granularity == QL_PER_GROUP;
    if (grp) {
        if (group <= 0 || K % group) { set_error("%s: group=%d does not divide K=%lld", fn, group, (long long)K); return QL_ESHAPE; }
        if (group % 32) {
            // the MFMA kernels fold a group's sum at 32-k steps; any other divisor of K (the reference takes them all,
            // quantization.py:205) runs the plain GEMM on the f32 kernel below, and has no fused forms
            if (silu || rope) {
                set_error("%s: group=%d is not a multiple of 32: only the plain GEMM exists for it", fn, group);
                return QL_ESHAPE;
            }
            if (M == 0) return QL_OK;
            const dim3 grid((unsigned)((N + 15) / 16), (unsigned)((M + 15) / 16));
            if (wdtype == QL_FP8)
                hipLaunchKernelGGL(qgemm_anygroup_kernel<true>, grid, dim3(256), 0, as_stream(stream), (const unsigned short*)x,
                                   (const unsigned char*)qw, sw, (const unsigned short*)bias, (unsigned short*)y, (int)M, (int)N,
                                   (int)K, group, tiled ? 1 : 0);
            else
                hipLaunchKernelGGL(qgemm_anygroup_kernel<false>, grid, dim3(256), 0, as_stream(stream), (const unsigned short*)x,
                                   (const unsigned char*)qw, sw, (const unsigned short*)bias, (unsigned short*)y, (int)M, (int)N,
                                   (int)K, group, tiled ? 1 : 0);
            QL_CHECK_LAUNCH(fn);
            return QL_OK;
        }
    }
    if (M == 0) return QL_OK;
    GemmParams p{};
    p.x = (const unsigned char*)x; p.w = (const unsigned char*)qw; p.sw = sw; p.sx = nullptr;
    p.bias = (const unsigned short*)bias; p.y = (unsigned short*)y; p.partial = nullptr; p.counters = nullptr;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group = grp ? group : 0; p.gmagic = grp ? group_magic(group) : 0; p.group_generic = group_generic();
    p.sw_stride = granularity == QL_PER_ROW ? 1 : 0;
    p.bias_after_round = granularity == QL_PER_ROW ? 0 : 1;
    p.silu = silu ? 1 : 0;
    p.wtiled = tiled ? 1 : 0;
    if (rope) { p.rope_on = 1; p.rope = *rope; }
    const Plan pl = make_plan(M, N, K, grp, silu, rope != nullptr);
    kern_t k;
    const bool rp = rope != nullptr;
    if (pl.sk_G > 0) {
        if (wdtype == QL_INT8) k = grp ? pick_sk_bf16_i8_group(pl.mt, pl.wn, pl.wk, pl.cp, rp) : pick_sk_bf16_i8_row(pl.mt, pl.wn, pl.wk, pl.cp, rp);
        else k = grp ? pick_sk_bf16_fp8_group(pl.mt, pl.wn, pl.wk, pl.cp, rp) : pick_sk_bf16_fp8_row(pl.mt, pl.wn, pl.wk, pl.cp, rp);
    } else if (wdtype == QL_INT8) k = grp ? pick_bf16_i8_group(pl.mt, pl.wn, pl.wk, pl.cp, rp) : pick_bf16_i8_row(pl.mt, pl.wn, pl.wk, pl.cp, rp);
    else k = grp ? pick_bf16_fp8_group(pl.mt, pl.wn, pl.wk, pl.cp, rp) : pick_bf16_fp8_row(pl.mt, pl.wn, pl.wk, pl.cp, rp);
    return run(k, p, pl, workspace, workspace_bytes, as_stream(stream), fn);
}

// rowmax_part != null: qx is the bf16 activation matrix and sx is unused (the kernel quantizes while staging)
static int gemm_w8a8(const char* fn, bool silu, int dtype, const void* qx, const float* sx, const void* qw,
                     const float* sw, const void* bias, void* y, int64_t M, int64_t N, int64_t K, void* workspace,
                     size_t workspace_bytes, void* stream, const RopeArgs* rope = nullptr,
                     const float* rowmax_part = nullptr, int nparts = 0, float* rowmax_out = nullptr) {
    const bool tiled = (dtype & QL_TILED) != 0;
    dtype &= ~QL_TILED;
    const bool dynq = rowmax_part != nullptr;
    if ((dtype != QL_INT8 && dtype != QL_FP8) || (M > 0 && (!qx || (!sx && !dynq) || (!y && !rope))) || !qw || !sw ||
        (dynq && (nparts <= 0 || silu || rope))) {
        set_error("%s: bad argument (dtype=%d)", fn, dtype);
        return QL_EINVAL;
    }
    int rc = check_common(fn, M, N, K);
    if (rc) return rc;
    if (silu && N % 64) { set_error("%s: N=%lld must be a multiple of 64 (32-row gate/up strips)", fn, (long long)N); return QL_ESHAPE; }
    rc = check_rope(fn, rope, M, N);
    if (rc) return rc;
    if (M == 0) return QL_OK;
    GemmParams p{};
    p.x = (const unsigned char*)qx; p.w = (const unsigned char*)qw; p.sw = sw; p.sx = sx;
    p.bias = (const unsigned short*)bias; p.y = (unsigned short*)y; p.partial = nullptr; p.counters = nullptr;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group = 0; p.sw_stride = 1; p.bias_after_round = 0;
    p.silu = silu ? 1 : 0;
    p.wtiled = tiled ? 1 : 0;
    if (rope) { p.rope_on = 1; p.rope = *rope; }
    p.rowmax_part = rowmax_part; p.nparts = nparts;
    p.rowmax_out = silu ? rowmax_out : nullptr; p.nparts_out = (int)(N / 64);
    const Plan pl = make_plan(M, N, K, false, silu, rope != nullptr, true);
    kern_t k;
    if (dynq) k = dtype == QL_INT8 ? pick_i8_i8_dynq(pl.mt, pl.wn, pl.wk, pl.cp) : pick_fp8_fp8_dynq(pl.mt, pl.wn, pl.wk, pl.cp);
    else k = dtype == QL_INT8 ? pick_i8_i8(pl.mt, pl.wn, pl.wk, pl.cp, rope != nullptr) : pick_fp8_fp8(pl.mt, pl.wn, pl.wk, pl.cp, rope != nullptr);
    return run(k, p, pl, workspace, workspace_bytes, as_stream(stream), fn);
}

}  // extern "C"

namespace ql {
// for the library's other translation units: the weight-only GEMM on a workspace of the caller's choice, and
// what a problem needs of it (total bytes; *ticket_bytes = the head of the workspace that must be zero)
int w8a16_dispatch(const char* fn, int wdtype, int granularity, const void* x, const void* qw, const float* sw,
                   const void* bias, void* y, int64_t M, int64_t N, int64_t K, int group, void* workspace,
                   size_t workspace_bytes, void* stream) {
    const SkipTable guard;
    return gemm_w8a16(fn, false, wdtype, granularity, x, qw, sw, bias, y, M, N, K, group, workspace, workspace_bytes, stream);
}
size_t w8a16_workspace_need(int64_t M, int64_t N, int64_t K, int granularity, size_t* ticket_bytes) {
    const SkipTable guard;
    *ticket_bytes = 0;
    if (M <= 0 || N <= 0 || K <= 0 || K % 32) return 0;
    const Plan pl = make_plan(M, N, K, granularity == QL_PER_GROUP);
    const size_t need = plan_workspace(pl);
    if (need) *ticket_bytes = ((size_t)pl.grid.x * pl.grid.y * 4 + 15) / 16 * 16;
    return need;
}
}  // namespace ql

extern "C" {

int ql_gemm_w8a16(int wdtype, int granularity, const void* x, const void* qw, const float* sw, const void* bias,
                  void* y, int64_t M, int64_t N, int64_t K, int group, void* workspace, size_t workspace_bytes,
                  void* stream) {
    return gemm_w8a16("ql_gemm_w8a16", false, wdtype, granularity, x, qw, sw, bias, y, M, N, K, group, workspace,
                      workspace_bytes, stream);
}

int ql_gemm_partial_slices(int64_t M, int64_t N, int64_t K, int granularity) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32) return 0;
    return make_plan_partial(M, N, K, granularity == QL_PER_GROUP).split;
}

int ql_gemm_w8a16_partial(int wdtype, int granularity, const void* x, const void* qw, const float* sw, int64_t M, int64_t N,
                          int64_t K, int group, float* partials, int slices, void* stream) {
    const char* fn = "ql_gemm_w8a16_partial";
    const bool tiled = (wdtype & QL_TILED) != 0;
    wdtype &= ~QL_TILED;
    if ((wdtype != QL_INT8 && wdtype != QL_FP8) || granularity < QL_PER_ROW || granularity > QL_PER_GROUP ||
        (M > 0 && (!x || !partials)) || !qw || !sw) {
        set_error("%s: bad argument (wdtype=%d granularity=%d)", fn, wdtype, granularity);
        return QL_EINVAL;
    }
    int rc = check_common(fn, M, N, K);
    if (rc) return rc;
    const bool grp = granularity == QL_PER_GROUP;
    if (grp && !(group >= 32 && group % 32 == 0 && K % group == 0)) {
        set_error("%s: group=%d unsupported for K=%lld", fn, group, (long long)K);
        return QL_ESHAPE;
    }
    if (M == 0) return QL_OK;
    const Plan pl = make_plan_partial(M, N, K, grp);
    if (slices != pl.split) {
        set_error("%s: slices=%d, but the plan for M=%lld N=%lld K=%lld has %d (ql_gemm_partial_slices)", fn, slices,
                  (long long)M, (long long)N, (long long)K, pl.split);
        return QL_EINVAL;
    }
    GemmParams p{};
    p.x = (const unsigned char*)x; p.w = (const unsigned char*)qw; p.sw = sw;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group = grp ? group : 0; p.gmagic = grp ? group_magic(group) : 0; p.group_generic = group_generic();
    p.sw_stride = granularity == QL_PER_ROW ? 1 : 0;
    p.wtiled = tiled ? 1 : 0;
    p.partial_rm = partials;
    p.k_per_split = pl.k_per_split;
    kern_t k;
    if (wdtype == QL_INT8) k = grp ? pick_bf16_i8_group(pl.mt, pl.wn, pl.wk, pl.cp) : pick_bf16_i8_row(pl.mt, pl.wn, pl.wk, pl.cp);
    else k = grp ? pick_bf16_fp8_group(pl.mt, pl.wn, pl.wk, pl.cp) : pick_bf16_fp8_row(pl.mt, pl.wn, pl.wk, pl.cp);
    if (!k) { set_error("%s: tile (%d,%d,%d,%d) not built", fn, pl.mt, pl.wn, pl.wk, pl.cp); return QL_EINVAL; }
    hipLaunchKernelGGL(k, pl.grid, dim3(64 * pl.wn * pl.wk), 0, as_stream(stream), p);
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

int ql_gemm_w8a8_partial(int dtype, const void* qx, const void* qw, int64_t M, int64_t N, int64_t K, void* partials,
                         int slices, void* stream) {
    const char* fn = "ql_gemm_w8a8_partial";
    const bool tiled = (dtype & QL_TILED) != 0;
    dtype &= ~QL_TILED;
    if ((dtype != QL_INT8 && dtype != QL_FP8) || (M > 0 && (!qx || !partials)) || !qw) {
        set_error("%s: bad argument (dtype=%d)", fn, dtype);
        return QL_EINVAL;
    }
    if (int rc = check_common(fn, M, N, K)) return rc;
    if (M == 0) return QL_OK;
    const Plan pl = make_plan_partial(M, N, K, false);
    if (slices != pl.split) {
        set_error("%s: slices=%d, but the plan for M=%lld N=%lld K=%lld has %d (ql_gemm_partial_slices)", fn, slices,
                  (long long)M, (long long)N, (long long)K, pl.split);
        return QL_EINVAL;
    }
    GemmParams p{};
    p.x = (const unsigned char*)qx; p.w = (const unsigned char*)qw;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.wtiled = tiled ? 1 : 0;
    p.partial_rm = (float*)partials;
    p.k_per_split = pl.k_per_split;
    kern_t k = dtype == QL_INT8 ? pick_i8_i8(pl.mt, pl.wn, pl.wk, pl.cp) : pick_fp8_fp8(pl.mt, pl.wn, pl.wk, pl.cp);
    if (!k) { set_error("%s: tile (%d,%d,%d,%d) not built", fn, pl.mt, pl.wn, pl.wk, pl.cp); return QL_EINVAL; }
    hipLaunchKernelGGL(k, pl.grid, dim3(64 * pl.wn * pl.wk), 0, as_stream(stream), p);
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

int ql_gemm_w8a16_silu(int wdtype, int granularity, const void* x, const void* qw, const float* sw, const void* bias,
                       void* y, int64_t M, int64_t N, int64_t K, int group, void* workspace, size_t workspace_bytes,
                       void* stream) {
    return gemm_w8a16("ql_gemm_w8a16_silu", true, wdtype, granularity, x, qw, sw, bias, y, M, N, K, group, workspace,
                      workspace_bytes, stream);
}

int ql_gemm_w8a8(int dtype, const void* qx, const float* sx, const void* qw, const float* sw, const void* bias,
                 void* y, int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_w8a8("ql_gemm_w8a8", false, dtype, qx, sx, qw, sw, bias, y, M, N, K, workspace, workspace_bytes, stream);
}

static RopeArgs rope_args(const int64_t* positions, const float* cos_sin, const void* q_norm_w, const void* k_norm_w,
                          float eps, void* q_out, void* k_out, void* v_out, void* k_cache, void* v_cache,
                          const int* slot_mapping, int Hq, int Hkv) {
    return RopeArgs{positions, cos_sin, (const unsigned short*)q_norm_w, (const unsigned short*)k_norm_w,
                    (unsigned short*)q_out, (unsigned short*)k_out, (unsigned short*)v_out, (unsigned short*)k_cache,
                    (unsigned short*)v_cache, slot_mapping, eps, Hq, Hkv};
}

int ql_gemm_w8a16_rope(int wdtype, int granularity, const void* x, const void* qw, const float* sw, const void* bias,
                       int64_t M, int64_t N, int64_t K, int group, const int64_t* positions, const float* cos_sin,
                       const void* q_norm_w, const void* k_norm_w, float eps, void* q_out, void* k_out, void* v_out,
                       void* k_cache, void* v_cache, const int* slot_mapping, int Hq, int Hkv, int head_dim,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (head_dim != 128) { set_error("ql_gemm_w8a16_rope: head_dim=%d (only 128 is built)", head_dim); return QL_ESHAPE; }
    const RopeArgs r = rope_args(positions, cos_sin, q_norm_w, k_norm_w, eps, q_out, k_out, v_out, k_cache, v_cache,
                                 slot_mapping, Hq, Hkv);
    return gemm_w8a16("ql_gemm_w8a16_rope", false, wdtype, granularity, x, qw, sw, bias, nullptr, M, N, K, group,
                      workspace, workspace_bytes, stream, &r);
}

int ql_gemm_w8a8_rope(int dtype, const void* qx, const float* sx, const void* qw, const float* sw, const void* bias,
                      int64_t M, int64_t N, int64_t K, const int64_t* positions, const float* cos_sin,
                      const void* q_norm_w, const void* k_norm_w, float eps, void* q_out, void* k_out, void* v_out,
                      void* k_cache, void* v_cache, const int* slot_mapping, int Hq, int Hkv, int head_dim,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (head_dim != 128) { set_error("ql_gemm_w8a8_rope: head_dim=%d (only 128 is built)", head_dim); return QL_ESHAPE; }
    const RopeArgs r = rope_args(positions, cos_sin, q_norm_w, k_norm_w, eps, q_out, k_out, v_out, k_cache, v_cache,
                                 slot_mapping, Hq, Hkv);
    return gemm_w8a8("ql_gemm_w8a8_rope", false, dtype, qx, sx, qw, sw, bias, nullptr, M, N, K, workspace, workspace_bytes,
                     stream, &r);
}

int ql_gemm_w8a8_silu(int dtype, const void* qx, const float* sx, const void* qw, const float* sw, const void* bias,
                      void* y, int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_w8a8("ql_gemm_w8a8_silu", true, dtype, qx, sx, qw, sw, bias, y, M, N, K, workspace, workspace_bytes, stream);
}

int ql_gemm_w8a8_silu_rowmax(int dtype, const void* qx, const float* sx, const void* qw, const float* sw, const void* bias,
                             void* y, float* rowmax_part_out, int64_t M, int64_t N, int64_t K, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (M > 0 && !rowmax_part_out) { set_error("ql_gemm_w8a8_silu_rowmax: bad argument"); return QL_EINVAL; }
    return gemm_w8a8("ql_gemm_w8a8_silu_rowmax", true, dtype, qx, sx, qw, sw, bias, y, M, N, K, workspace, workspace_bytes,
                     stream, nullptr, nullptr, 0, rowmax_part_out);
}

int ql_gemm_w8a8_dynq(int dtype, const void* x, const float* rowmax_part, int nparts, const void* qw, const float* sw,
                      const void* bias, void* y, int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (!rowmax_part) { set_error("ql_gemm_w8a8_dynq: bad argument"); return QL_EINVAL; }
    return gemm_w8a8("ql_gemm_w8a8_dynq", false, dtype, x, nullptr, qw, sw, bias, y, M, N, K, workspace, workspace_bytes,
                     stream, nullptr, rowmax_part, nparts);
}

// ---- output projection: bf16 activations x the bf16 embedding table, logits or the greedy sampler's tokens ----------
// One row tile covers the step's sequences wherever it can (the table is then streamed from HBM once): 32, 64 or
// 128 rows x 128 columns, 256 rows x 256 columns (eight waves, 64-k chunks); beyond 256 rows, 128 x 256 tiles whose
// row tiles run side by side on one XCD.  Every workgroup runs the whole K.  Measured with tools/bench_lmhead.py.
static Plan make_plan_head(int64_t M, int64_t V, int64_t K) {
    Plan pl;
    // (129..256 rows: one row tile of as many 32-row fragments as the rows fill, 5..8)
    int mt = M <= 32 ? 1 : M <= 64 ? 2 : M <= 128 ? 4 : M <= 256 ? (int)((M + 31) / 32) : 4, wn = M <= 160 ? 4 : 8, wk = 1,
        cp = (M > 224 && M <= 256) ? 2 : 4;
    pl.pair_rows = true;
    if (f_mt > 0 && f_wn > 0 && f_wk > 0 && f_cp > 0 && pick_head(f_mt, f_wn, f_wk, f_cp, false)) {
        mt = f_mt; wn = f_wn; wk = f_wk; cp = f_cp;
    }
    pl.mt = mt; pl.wn = wn; pl.wk = wk; pl.cp = cp;
    const int kb = 32 * cp * wk;
    pl.split = 1;
    pl.k_per_split = (int)((K + kb - 1) / kb * kb);
    pl.grid = dim3((unsigned)((V + 32 * wn - 1) / (32 * wn)), (unsigned)((M + 32 * mt - 1) / (32 * mt)), 1);
    return pl;
}

static int head_common(const char* fn, const void* x, const void* w, int64_t M, int64_t V, int64_t K) {
    if (M < 0 || V <= 0 || K <= 0 || (M > 0 && (!x || !w))) { set_error("%s: bad argument", fn); return QL_EINVAL; }
    if (K % 32) { set_error("%s: K=%lld must be a multiple of 32", fn, (long long)K); return QL_ESHAPE; }
    // 32-bit byte offsets from the operand bases
    if (V > 0x3FFFFFFF || (M + 255) / 32 > 65535 || (uint64_t)V * (uint64_t)K * 2 >= (1ull << 32) ||
        (uint64_t)(M + 256) * (uint64_t)K * 2 >= (1ull << 32)) {
        set_error("%s: shape too large", fn);
        return QL_ESHAPE;
    }
    return QL_OK;
}

static GemmParams head_params(const void* x, const void* w, int64_t M, int64_t V, int64_t K) {
    GemmParams p{};
    p.x = (const unsigned char*)x;
    p.w = (const unsigned char*)w;
    p.M = (int)M; p.N = (int)V; p.K = (int)K;
    p.sw_stride = 0;
    return p;
}

namespace ql {
// rows of {f32 bits of a maximum, its column} pairs, one pair per column tile -> the row's token
__global__ __launch_bounds__(256) void argmax_finish_kernel(const int2* __restrict__ part, int ntiles, int64_t* __restrict__ out,
                                                            float* __restrict__ max_out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int2* row = part + (size_t)blockIdx.x * ntiles;
    float best = -INFINITY;
    int bi = 0x7FFFFFFF;
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        const int2 e = row[t];
        const float v = __builtin_bit_cast(float, e.x);
        if (v > best || (v == best && e.y < bi)) { best = v; bi = e.y; }
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
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        out[blockIdx.x] = (bi == 0x7FFFFFFF) ? 0 : bi;
        if (max_out) max_out[blockIdx.x] = best;              // -inf when the row has no finite maximum
    }
}
}  // namespace ql

int ql_lmhead_logits(const void* x, const void* w, void* logits, int64_t M, int64_t V, int64_t K, void* stream) {
    const char* fn = "ql_lmhead_logits";
    if (int rc = head_common(fn, x, w, M, V, K)) return rc;
    if (M > 0 && !logits) { set_error("%s: bad argument", fn); return QL_EINVAL; }
    if (M == 0) return QL_OK;
    const Plan pl = make_plan_head(M, V, K);
    GemmParams p = head_params(x, w, M, V, K);
    p.y = (unsigned short*)logits;
    return run(pick_head(pl.mt, pl.wn, pl.wk, pl.cp, false), p, pl, nullptr, 0, as_stream(stream), fn);
}

size_t ql_lmhead_workspace_bytes(int64_t M, int64_t V) {
    if (M <= 0 || V <= 0) return 0;
    return (size_t)M * (size_t)((V + 127) / 128) * sizeof(int2);       // (128 columns: the narrowest tile)
}

int ql_lmhead_argmax(const void* x, const void* w, int64_t* tokens, float* max_out, int64_t M, int64_t V, int64_t K,
                     void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "ql_lmhead_argmax";
    if (int rc = head_common(fn, x, w, M, V, K)) return rc;
    if (M > 0 && !tokens) { set_error("%s: bad argument", fn); return QL_EINVAL; }
    if (M == 0) return QL_OK;
    const Plan pl = make_plan_head(M, V, K);
    const size_t need = (size_t)M * pl.grid.x * sizeof(int2);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: workspace of %zu bytes needed, got %zu", fn, need, workspace_bytes);
        return QL_EWORKSPACE;
    }
    GemmParams p = head_params(x, w, M, V, K);
    p.amax = (int2*)workspace;
    if (int rc = run(pick_head(pl.mt, pl.wn, pl.wk, pl.cp, true), p, pl, nullptr, 0, as_stream(stream), fn)) return rc;
    hipLaunchKernelGGL(argmax_finish_kernel, dim3((unsigned)M), dim3(256), 0, as_stream(stream), (const int2*)workspace,
                       (int)pl.grid.x, tokens, max_out);
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

}  // extern "C"
