// ql_gemm_w8a16_noalloc: the weight-only GEMM for callers that cannot allocate device memory (code under hipGraph
// capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels and results as ql_gemm_w8a16; the
// split-K workspace (ticket counters + partial slabs) is a fixed array inside the library's code object.
#include "ql_gemm_kernel.h"
#include <mutex>

#ifndef QL_NOALLOC_SCRATCH_BYTES
#error "include/qlinear.h must define QL_NOALLOC_SCRATCH_BYTES"
#endif

// Held by every GPU the library runs on until the process ends.  In host code this name is only a handle for
// hipGetSymbolAddress: the device address is looked up once per device and that pointer is what the kernels get.
static __device__ __align__(256) unsigned char scratch[QL_NOALLOC_SCRATCH_BYTES];

// clears the ticket counters a call will use (a kernel rather than a memset node: under stream capture it is
// ordered with the GEMM like any other launch of the stream)
__global__ __launch_bounds__(256) void clear_words_kernel(unsigned* __restrict__ p, unsigned nwords) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < nwords) p[i] = 0u;
}

extern "C" int ql_gemm_w8a16_noalloc(int wdtype, int granularity, const void* x, const void* qw, const float* sw,
                                     const void* bias, void* y, int64_t M, int64_t N, int64_t K, int group, void* stream) {
    constexpr int kMaxDev = 64;
    static void* dev_ptr[kMaxDev] = {};
    static std::mutex mu;
    const char* fn = "ql_gemm_w8a16_noalloc";
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        ql::set_error("%s: no current device", fn);
        return QL_ELAUNCH;
    }
    void* ptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!dev_ptr[dev]) {
            // once per device, with that device current: each GPU holds its own copy of the array at its own address
            void* p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !p) {
                ql::set_error("%s: hipGetSymbolAddress failed: %s", fn, hipGetErrorString(e));
                return QL_ELAUNCH;
            }
            dev_ptr[dev] = p;
        }
        ptr = dev_ptr[dev];
    }
    size_t tickets = 0;
    const size_t need = ql::w8a16_workspace_need(M, N, K, granularity, &tickets);
    if (need > QL_NOALLOC_SCRATCH_BYTES) {
        ql::set_error("%s: M=%lld N=%lld K=%lld needs %zu bytes of split-K scratch, the built-in buffer holds %u: "
                      "use ql_gemm_w8a16 with a caller-owned workspace", fn, (long long)M, (long long)N, (long long)K,
                      need, (unsigned)QL_NOALLOC_SCRATCH_BYTES);
        return QL_EWORKSPACE;
    }
    // the array keeps the last call's contents: the tickets this call takes must start at zero
    if (tickets) {
        const unsigned nwords = (unsigned)(tickets / 4);
        hipLaunchKernelGGL(clear_words_kernel, dim3((nwords + 255) / 256), dim3(256), 0, (hipStream_t)stream, (unsigned*)ptr, nwords);
        QL_CHECK_LAUNCH(fn);
    }
    return ql::w8a16_dispatch(fn, wdtype, granularity, x, qw, sw, bias, y, M, N, K, group, ptr, QL_NOALLOC_SCRATCH_BYTES,
                              stream);
}
