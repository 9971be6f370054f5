// One-shot all-reduce of the row-parallel partial outputs over peer-mapped buffers, with the residual add + RMSNorm
// that follows it in the same launch (RowParallelLinear.forward, nanovllm/layers/linear.py:149-153 of the reference:
// F.linear on the K shard, then dist.all_reduce; Qwen3DecoderLayer.forward, nanovllm/models/qwen3.py:153-162: every
// o_proj / down_proj output goes straight into add_rms_forward).
//
// At decode sizes the message is at most a few hundred KiB and a step issues two of them per layer, so the collective
// is latency-bound: a ring pays 2 (P - 1) hops and uses one xGMI link per hop.  Here every rank PUSHES its bf16 partial
// row into a slot of every peer's buffer (P - 1 posted writes, one per link, all links at once), raises a per-row flag
// at every peer, waits for the P - 1 flags raised at its own buffer and sums the P partials from LOCAL memory in rank
// order in f32 - the same values in the same order on every rank, so all ranks hold identical bits and a replay gives
// the same bits again.  One workgroup per token row: nothing in the launch depends on another workgroup of the same
// rank, so no grid-wide hand-off exists.
//
// Buffer of one rank (ql_comm_buffer_bytes; identical layout on every rank, zero-filled at allocation):
//   [0] epoch  [4] ticket  [8] timeouts                                   (256-byte header)
//   flags  [2 parities][world sources][max_rows]            u32           (a source's flag for a row = epoch + 1)
//   data   [2 parities][world sources][max_rows][hidden]    bf16
// The epoch lives in device memory and is advanced by the launch itself (its last workgroup), so a captured hipGraph
// replays correctly; the parity of the epoch selects the half of flags / data, which is all the double buffering a
// stream-ordered sequence of calls needs: a rank can be at most one call ahead of a peer, because the call after that
// waits for the flag the peer raises only once it has finished reading the previous one.
//
// Every wait is a bounded spin.  When the bound is reached (a peer that never arrives) the workgroup counts a timeout in
// the header and carries on with whatever its slots hold: the result is wrong, the process does not hang, and
// ql_comm_status reports it.  Data and flags travel as system-scope relaxed atomics (write-through stores, cache-
// bypassing loads); completed stores (vmcnt(0) + barrier) before the flag and the flag before the loads order them.
#include "ql_common.h"
#include <string.h>

namespace ql {

constexpr int kCommHeader = 256;
constexpr int kMaxWorld = 8;

struct CommArgs {
    unsigned char* buf[kMaxWorld];     // buf[rank] = this rank's buffer, the others peer-mapped
    int rank, world, max_rows, hidden;
};

__device__ __forceinline__ size_t comm_flag_off(const CommArgs& c, int parity, int src, int row) {
    return (size_t)kCommHeader + ((size_t)(parity * c.world + src) * c.max_rows + row) * 4;
}
__device__ __forceinline__ size_t comm_data_off(const CommArgs& c, int parity, int src, int row) {
    const size_t flags = (size_t)2 * c.world * c.max_rows * 4;
    return (size_t)kCommHeader + flags + ((size_t)(parity * c.world + src) * c.max_rows + row) * (size_t)c.hidden * 2;
}

static size_t comm_bytes(int world, int64_t max_rows, int64_t hidden) {
    return (size_t)kCommHeader + (size_t)2 * world * max_rows * 4 + (size_t)2 * world * max_rows * hidden * 2;
}

typedef unsigned long long u64;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// NV = 16-byte vectors per thread (hidden / 8 / 256 rounded up).  weight == null: plain all-reduce (y_out only).
template <int NV>
__global__ __launch_bounds__(256) void allreduce_add_rmsnorm_kernel(const CommArgs c, const unsigned short* __restrict__ x,
                                                                    const unsigned short* __restrict__ res,
                                                                    const unsigned short* __restrict__ w,
                                                                    unsigned short* __restrict__ y_out,
                                                                    unsigned short* __restrict__ out,
                                                                    unsigned short* __restrict__ res_out, int H, float eps,
                                                                    unsigned spin_limit) {
    __shared__ float sh[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    unsigned char* mine = c.buf[c.rank];
    // (written by the previous launch of this stream, or by the allocation's memset)
    const unsigned e = __hip_atomic_load((unsigned*)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int par = (int)(e & 1u);
    const unsigned tag = e + 1u;
    const int nvec = H >> 3;

    // 1. this rank's partial row: keep it, and push it into slot [rank] of every peer
    u16x8 a[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int cc = tid + 256 * j;
        if (cc < nvec) {
            a[j] = *(const u16x8*)(x + (size_t)row * H + cc * 8);
            const i32x4 bits = __builtin_bit_cast(i32x4, a[j]);
            const u64 lo = ((u64)(unsigned)bits[1] << 32) | (unsigned)bits[0], hi = ((u64)(unsigned)bits[3] << 32) | (unsigned)bits[2];
            for (int q = 0; q < c.world; ++q) {
                if (q == c.rank) continue;
                u64* dst = (u64*)(c.buf[q] + comm_data_off(c, par, c.rank, row)) + cc * 2;
                __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // 2. raise this row's flag at every peer; 3. wait for every peer's flag at this rank (bounded)
    if (tid < c.world && tid != c.rank) {
        __hip_atomic_store((unsigned*)(c.buf[tid] + comm_flag_off(c, par, c.rank, row)), tag, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned* f = (const unsigned*)(mine + comm_flag_off(c, par, tid, row));
        unsigned left = spin_limit;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
            if (--left == 0u) {
                __hip_atomic_fetch_add((unsigned*)(mine + 8), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();

    // 4. the sum, in rank order, from local memory (this rank's own term from registers: the same bits it pushed)
    float v[NV][8];
    u16x8 ww[NV];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int cc = tid + 256 * j;
        if (cc < nvec) {
            if (w) ww[j] = *(const u16x8*)(w + cc * 8);
            u16x8 b = {0, 0, 0, 0, 0, 0, 0, 0};
            if (res) b = *(const u16x8*)(res + (size_t)row * H + cc * 8);
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0.f;
            for (int q = 0; q < c.world; ++q) {
                u16x8 t;
                if (q == c.rank) {
                    t = a[j];
                } else {
                    const u64* src = (const u64*)(mine + comm_data_off(c, par, q, row)) + cc * 2;
                    const u64 lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const u64 hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const i32x4 bits = {(int)(unsigned)lo, (int)(unsigned)(lo >> 32), (int)(unsigned)hi, (int)(unsigned)(hi >> 32)};
                    t = __builtin_bit_cast(u16x8, bits);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __fadd_rn(acc[k], bf16_bits_to_f32(t[k]));
            }
            u16x8 y;
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = f32_to_bf16_bits(acc[k]);      // the all-reduced bf16 output
            if (y_out) *(u16x8*)(y_out + (size_t)row * H + cc * 8) = y;
            // 5. from here on: the arithmetic of ql_add_rmsnorm on y, statement for statement
            if (res) {
                u16x8 r;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[j][k] = bf16_bits_to_f32(y[k]) + bf16_bits_to_f32(b[k]);
                    r[k] = f32_to_bf16_bits(v[j][k]);
                }
                if (res_out) *(u16x8*)(res_out + (size_t)row * H + cc * 8) = r;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[j][k] = bf16_bits_to_f32(y[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) ss += v[j][k] * v[j][k];
        }
    }
    if (w) {
        ss = wave_sum_f(ss);
        if ((tid & 63) == 0) sh[tid >> 6] = ss;
        __syncthreads();
        ss = sh[0] + sh[1] + sh[2] + sh[3];
        const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int cc = tid + 256 * j;
            if (cc < nvec) {
                u16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    o[k] = f32_to_bf16_bits(round_to_bf16(v[j][k] * inv) * bf16_bits_to_f32(ww[j][k]));
                *(u16x8*)(out + (size_t)row * H + cc * 8) = o;
            }
        }
    }
    // 6. the launch's last workgroup advances the epoch (every workgroup has read it by then)
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add((unsigned*)(mine + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store((unsigned*)(mine + 4), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((unsigned*)mine, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace ql

using namespace ql;

extern "C" {

size_t ql_comm_buffer_bytes(int world, int64_t max_rows, int64_t hidden) {
    if (world < 1 || world > kMaxWorld || max_rows <= 0 || hidden <= 0 || hidden % 8) return 0;
    return comm_bytes(world, max_rows, hidden);
}

int ql_comm_alloc(size_t bytes, void** buf) {
    const char* fn = "ql_comm_alloc";
    if (!buf || bytes < (size_t)kCommHeader) { set_error("%s: bad argument", fn); return QL_EINVAL; }
    void* p = nullptr;
    // fine-grained device memory: stores of another agent become visible without a kernel boundary
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) { set_error("%s: %zu bytes: %s", fn, bytes, hipGetErrorString(e)); return QL_ELAUNCH; }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); set_error("%s: memset: %s", fn, hipGetErrorString(e)); return QL_ELAUNCH; }
    *buf = p;
    return QL_OK;
}

int ql_comm_free(void* buf) {
    if (buf && hipFree(buf) != hipSuccess) { set_error("ql_comm_free: %s", hipGetErrorString(hipGetLastError())); return QL_ELAUNCH; }
    return QL_OK;
}

int ql_comm_export(void* buf, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == QL_COMM_HANDLE_BYTES, "handle size");
    if (!buf || !handle64) { set_error("ql_comm_export: bad argument"); return QL_EINVAL; }
    const hipError_t e = hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, buf);
    if (e != hipSuccess) { set_error("ql_comm_export: %s", hipGetErrorString(e)); return QL_ELAUNCH; }
    return QL_OK;
}

int ql_comm_import(const void* handle64, void** peer_buf) {
    if (!handle64 || !peer_buf) { set_error("ql_comm_import: bad argument"); return QL_EINVAL; }
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { set_error("ql_comm_import: %s", hipGetErrorString(e)); return QL_ELAUNCH; }
    *peer_buf = p;
    return QL_OK;
}

int ql_comm_unimport(void* peer_buf) {
    if (peer_buf && hipIpcCloseMemHandle(peer_buf) != hipSuccess) {
        set_error("ql_comm_unimport: %s", hipGetErrorString(hipGetLastError()));
        return QL_ELAUNCH;
    }
    return QL_OK;
}

int ql_comm_status(const void* buf, unsigned* epoch, unsigned* timeouts) {
    if (!buf) { set_error("ql_comm_status: bad argument"); return QL_EINVAL; }
    unsigned hdr[3] = {0, 0, 0};
    const hipError_t e = hipMemcpy(hdr, buf, sizeof(hdr), hipMemcpyDeviceToHost);     // (synchronizes)
    if (e != hipSuccess) { set_error("ql_comm_status: %s", hipGetErrorString(e)); return QL_ELAUNCH; }
    if (epoch) *epoch = hdr[0];
    if (timeouts) *timeouts = hdr[2];
    return QL_OK;
}

int ql_allreduce_add_rmsnorm(int rank, int world, void* const* bufs, int64_t max_rows, int64_t hidden, const void* x_partial,
                             const void* residual, const void* weight, float eps, void* y_out, void* out, void* residual_out,
                             int64_t T, unsigned spin_limit, void* stream) {
    const char* fn = "ql_allreduce_add_rmsnorm";
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world || !bufs || T < 0 || max_rows <= 0 ||
        (T > 0 && !x_partial) || (weight && !out) || (!weight && !y_out) || (residual && !weight)) {
        set_error("%s: bad argument", fn);
        return QL_EINVAL;
    }
    if (hidden <= 0 || hidden % 8 || hidden > 8192) { set_error("%s: hidden=%lld must be a multiple of 8, at most 8192", fn, (long long)hidden); return QL_ESHAPE; }
    if (T > max_rows) {
        set_error("%s: %lld rows, but the buffers were sized for %lld (larger messages: the library collective)", fn,
                  (long long)T, (long long)max_rows);
        return QL_ESHAPE;
    }
    CommArgs c{};
    for (int q = 0; q < world; ++q) {
        if (!bufs[q]) { set_error("%s: buffer of rank %d missing", fn, q); return QL_EINVAL; }
        c.buf[q] = (unsigned char*)bufs[q];
    }
    if (T == 0) return QL_OK;
    c.rank = rank; c.world = world; c.max_rows = (int)max_rows; c.hidden = (int)hidden;
    if (spin_limit == 0) spin_limit = 1u << 22;               // a few seconds of polling
    const int nv = (int)((hidden / 8 + 255) / 256);
    dim3 grid((unsigned)T), block(256);
    hipStream_t s = as_stream(stream);
#define QL_AR(NV) hipLaunchKernelGGL((allreduce_add_rmsnorm_kernel<NV>), grid, block, 0, s, c, (const unsigned short*)x_partial, \
        (const unsigned short*)residual, (const unsigned short*)weight, (unsigned short*)y_out, (unsigned short*)out,      \
        (unsigned short*)residual_out, (int)hidden, eps, spin_limit)
    if (nv == 1) QL_AR(1); else if (nv == 2) QL_AR(2); else QL_AR(4);
#undef QL_AR
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

}  // extern "C"
