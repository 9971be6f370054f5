// What a grid-wide hand-off inside one persistent launch costs on this chip, next to a launch boundary inside a
// hipGraph.  Stand-alone: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
//   1. graph of empty launches                      -> launch boundary
//   2. persistent kernel, counter barrier only      -> barrier
//   3. barrier + hand-off: every workgroup writes 2 KiB (write-through stores), after the barrier reads RD KiB of what
//      the others wrote with cache-bypassing loads
//   4. the same with plain loads behind an agent-scope acquire fence
// Every spin is bounded: a workgroup that does not see the others after SPIN_MAX polls sets an error word and leaves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr unsigned SPIN_MAX = 1u << 20;

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned n = 0, good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++n > SPIN_MAX) { good = 0; *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__global__ void empty_kernel(unsigned* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void clear_kernel(unsigned* p) { if (threadIdx.x == 0) __hip_atomic_store(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void barrier_only(unsigned* ctr, unsigned* err, int rounds) {
    for (int r = 0; r < rounds; ++r)
        if (!grid_barrier(ctr, (unsigned)(r + 1) * gridDim.x, err)) return;
}

// mode 0: bypass loads; 1: plain loads behind an acquire fence; 2: plain loads, no fence (wrong, only for timing)
template <int MODE>
__global__ __launch_bounds__(256) void barrier_handoff(unsigned* ctr, unsigned* err, int rounds, float* buf, int rd_kib, float* sink) {
    const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float* b = buf + (size_t)(r & 1) * nwg * 512;
        // write 2 KiB per workgroup: 512 floats, write-through
        for (int i = tid; i < 512; i += 256)
            __hip_atomic_store(b + (size_t)wg * 512 + i, (float)(r + wg + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(ctr, (unsigned)(r + 1) * nwg, err)) return;
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // read rd_kib KiB starting at another workgroup's part (wraps)
        const int nfl = rd_kib * 256, total = nwg * 512;
        int start = ((wg * 37 + 11) % nwg) * 512;
        for (int i = tid * 4; i < nfl; i += 1024) {
            const int idx = (start + i) % total;
            if (MODE == 0) {
                acc += __hip_atomic_load(b + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += __hip_atomic_load(b + idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += __hip_atomic_load(b + idx + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += __hip_atomic_load(b + idx + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (MODE == 3) {
                float4 v;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(b + idx) : "memory");
                acc += v.x + v.y + v.z + v.w;
            } else {
                const float4 v = *(const float4*)(b + idx);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

static float time_graph(hipStream_t st, hipGraphExec_t ge, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    const int rounds = 200;
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned *ctr, *err; float *buf, *sink;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&buf, (size_t)2 * nwg * 2048 + 4096)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(err, 0, 4)); CK(hipMemset(buf, 0, (size_t)2 * nwg * 2048 + 4096));

    auto capture = [&](auto body) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        body();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        return ge;
    };
    // 1. launch boundary
    {
        auto ge = capture([&] { for (int i = 0; i < rounds; ++i) empty_kernel<<<nwg, 256, 0, st>>>(nullptr); });
        printf("graph of %d empty launches (%d wg): %.2f us per launch\n", rounds, nwg, time_graph(st, ge, 20) / rounds);
    }
    auto check = [&](const char* what) {
        unsigned h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        if (h) { printf("%s: SPIN LIMIT HIT\n", what); exit(2); }
    };
    // 2. barrier only (one launch of `rounds` barriers inside a graph, counter cleared by a memset node in front)
    {
        auto ge = capture([&] {
            clear_kernel<<<1, 64, 0, st>>>(ctr);
            barrier_only<<<nwg, 256, 0, st>>>(ctr, err, rounds);
        });
        const float us = time_graph(st, ge, 20);
        check("barrier_only");
        printf("persistent kernel, %d barriers: %.2f us per barrier (launch included)\n", rounds, us / rounds);
    }
    for (int rd : {8, 32, 128}) {
        auto run = [&](int mode) {
            auto ge = capture([&] {
                clear_kernel<<<1, 64, 0, st>>>(ctr);
                if (mode == 0) barrier_handoff<0><<<nwg, 256, 0, st>>>(ctr, err, rounds, buf, rd, sink);
                if (mode == 1) barrier_handoff<1><<<nwg, 256, 0, st>>>(ctr, err, rounds, buf, rd, sink);
                if (mode == 2) barrier_handoff<2><<<nwg, 256, 0, st>>>(ctr, err, rounds, buf, rd, sink);
                if (mode == 3) barrier_handoff<3><<<nwg, 256, 0, st>>>(ctr, err, rounds, buf, rd, sink);
            });
            const float us = time_graph(st, ge, 10);
            check("barrier_handoff");
            return us / rounds;
        };
        printf("hand-off, 2 KiB written + %3d KiB read per workgroup: bypass loads %.2f us, acquire fence + plain loads %.2f us, plain loads (no fence) %.2f us, 16-B sc1 loads %.2f us per round\n",
               rd, run(0), run(1), run(2), run(3));
    }
    return 0;
}
