// Quantizer kernels: per-row (dynamic, also used on activations), per-group, per-tensor and the
// quantize-dequantize ("fake") forms.  All are HBM-bound byte movers: 16-byte loads per lane,
// one pass for the row/group maxima and one for the conversion (the second pass hits L2).
#include "ql_common.h"

namespace ql {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

template <int DT> struct QMax;
template <> struct QMax<QL_INT8> { static constexpr float v = 127.f; };
template <> struct QMax<QL_FP8> { static constexpr float v = 448.f; };

__device__ __forceinline__ float absmax8(const u16x8& v) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(bf16_bits_to_f32(v[j])));
    return m;
}

// max over the 256 threads of a block (4 waves)
__device__ __forceinline__ float block_max_256(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return v;
}

// ---- per-row: one 256-thread block per row ---------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void quant_per_row_kernel(const unsigned short* __restrict__ x,
                                                            unsigned char* __restrict__ q,
                                                            float* __restrict__ scale, int64_t H) {
    __shared__ float sh[4];
    const int64_t row = blockIdx.x;
    const unsigned short* xr = x + row * H;
    float m = 0.f;
    for (int64_t c = threadIdx.x * 8; c < H; c += 256 * 8) m = fmaxf(m, absmax8(*(const u16x8*)(xr + c)));
    m = block_max_256(m, sh);
    const float row_max = fmaxf(m, 1e-8f);
    const float sc = row_max / QMax<DT>::v;
    const float sc_inv = QMax<DT>::v / row_max;
    if (threadIdx.x == 0) scale[row] = sc;
    unsigned char* qr = q + row * H;
    for (int64_t c = threadIdx.x * 8; c < H; c += 256 * 8) {
        const u16x8 v = *(const u16x8*)(xr + c);
        unsigned long long out = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = bf16_bits_to_f32(v[j]) * sc_inv;
            unsigned b;
            if (DT == QL_INT8) b = (unsigned)(unsigned char)(signed char)(int)roundf(y);
            else b = f32_to_e4m3fn(y);
            out |= (unsigned long long)b << (8 * j);
        }
        *(unsigned long long*)(qr + c) = out;
    }
}

// the same for activations that arrive as fp16 or fp32 (IN = 1 / 2): the reference's kernel converts whatever it loads
// to f32 first (quantization.py:27), so the arithmetic is that of the bf16 form on the widened values
template <int DT, int IN>
__global__ __launch_bounds__(256) void quant_per_row_wide_kernel(const void* __restrict__ xv, unsigned char* __restrict__ q,
                                                                 float* __restrict__ scale, int64_t H) {
    __shared__ float sh[4];
    const int64_t row = blockIdx.x;
    auto load8 = [&](int64_t c, float* f) {
        if constexpr (IN == 1) {
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            const f16x8 v = *(const f16x8*)((const _Float16*)xv + row * H + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
        } else {
            const float4 a = *(const float4*)((const float*)xv + row * H + c), b = *(const float4*)((const float*)xv + row * H + c + 4);
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
        }
    };
    float m = 0.f;
    for (int64_t c = threadIdx.x * 8; c < H; c += 256 * 8) {
        float f[8];
        load8(c, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(f[j]));
    }
    m = block_max_256(m, sh);
    const float row_max = fmaxf(m, 1e-8f);
    const float sc_inv = QMax<DT>::v / row_max;
    if (threadIdx.x == 0) scale[row] = row_max / QMax<DT>::v;
    unsigned char* qr = q + row * H;
    for (int64_t c = threadIdx.x * 8; c < H; c += 256 * 8) {
        float f[8];
        load8(c, f);
        unsigned long long out = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = f[j] * sc_inv;
            unsigned b;
            if (DT == QL_INT8) b = (unsigned)(unsigned char)(signed char)(int)roundf(y);
            else b = f32_to_e4m3fn(y);
            out |= (unsigned long long)b << (8 * j);
        }
        *(unsigned long long*)(qr + c) = out;
    }
}

// ---- partial row maxima: out[row][i] = max |x[row, i*H/parts : (i+1)*H/parts]| (one wave per part) ----
__global__ __launch_bounds__(256) void row_absmax_parts_kernel(const unsigned short* __restrict__ x, float* __restrict__ out,
                                                               int64_t H, int parts) {
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t per = H / parts;
    for (int i = threadIdx.x >> 6; i < parts; i += 4) {
        const unsigned short* xp = x + row * H + (int64_t)i * per;
        float m = 0.f;
        for (int64_t c = lane * 8; c < per; c += 64 * 8) m = fmaxf(m, absmax8(*(const u16x8*)(xp + c)));
        m = wave_max(m);
        if (lane == 0) out[row * parts + i] = m;
    }
}

// ---- per-group: g/8 adjacent lanes share a group ------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void quant_per_group_kernel(const unsigned short* __restrict__ x,
                                                              unsigned char* __restrict__ q,
                                                              float* __restrict__ scales, int64_t total,
                                                              int group) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const bool live = e < total;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) v = *(const u16x8*)(x + e);
    float m = absmax8(v);
    const int lanes = group >> 3;
    for (int o = 1; o < lanes; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float sc = fmaxf(m, 1e-8f) / QMax<DT>::v;
    if (!live) return;
    if ((threadIdx.x & (lanes - 1)) == 0) scales[e / group] = sc;
    unsigned long long out = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float y = bf16_bits_to_f32(v[j]) / sc;
        unsigned b;
        if (DT == QL_INT8) b = (unsigned)(unsigned char)(signed char)(int)fminf(fmaxf(rintf(y), -128.f), 127.f);
        else b = f32_to_e4m3fn(y);
        out |= (unsigned long long)b << (8 * j);
    }
    *(unsigned long long*)(q + e) = out;
}

// ---- per-tensor: global |max| via atomicMax on the (non-negative) f32 bit pattern ---------------
__global__ __launch_bounds__(256) void absmax_kernel(const unsigned short* __restrict__ x, int64_t total,
                                                     unsigned* __restrict__ ws) {
    __shared__ float sh[4];
    float m = 0.f;
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; e < total; e += (int64_t)gridDim.x * 256 * 8)
        m = fmaxf(m, absmax8(*(const u16x8*)(x + e)));
    m = block_max_256(m, sh);
    if (threadIdx.x == 0) atomicMax(ws, __builtin_bit_cast(unsigned, m));
}

template <int DT>
__global__ __launch_bounds__(256) void quant_per_tensor_kernel(const unsigned short* __restrict__ x,
                                                               unsigned char* __restrict__ q,
                                                               float* __restrict__ scale_out,
                                                               const unsigned* __restrict__ ws, int64_t total) {
    // amax, scale and x/scale are bf16 tensors in the reference: round each to bf16
    const float amax = fmaxf(__builtin_bit_cast(float, ws[0]), round_to_bf16(1e-8f));
    const float sc = round_to_bf16(amax / QMax<DT>::v);
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = sc;
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; e < total; e += (int64_t)gridDim.x * 256 * 8) {
        const u16x8 v = *(const u16x8*)(x + e);
        unsigned long long out = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = round_to_bf16(bf16_bits_to_f32(v[j]) / sc);
            unsigned b;
            if (DT == QL_INT8) b = (unsigned)(unsigned char)(signed char)(int)fminf(fmaxf(rintf(y), -128.f), 127.f);
            else b = f32_to_e4m3fn(y);
            out |= (unsigned long long)b << (8 * j);
        }
        *(unsigned long long*)(q + e) = out;
    }
}

// ---- fake quant: quantize-dequantize one value given its block's amax (bf16-valued) -----------
template <int DT>
__device__ __forceinline__ unsigned short fake_one(float x, float amax) {
    float y;
    if (DT == QL_INT8) {
        const float step = round_to_bf16(amax / 127.f);   // bf16 tensor / int stays bf16
        const float t = (x / amax) * 127.f;
        y = rintf(t) * step;
    } else {
        const float t = (x / amax) * 448.f;
        const float d = e4m3fn_to_f32(f32_to_e4m3fn(t));
        y = (d / 448.f) * amax;
    }
    return f32_to_bf16_bits(y);
}

template <int DT>
__device__ __forceinline__ void fake8(const u16x8& v, float amax, unsigned short* out) {
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fake_one<DT>(bf16_bits_to_f32(v[j]), amax);
    *(u16x8*)out = o;
}

template <int DT>
__global__ __launch_bounds__(256) void fake_per_row_kernel(const unsigned short* __restrict__ x,
                                                           unsigned short* __restrict__ out, int64_t H) {
    __shared__ float sh[4];
    const int64_t row = blockIdx.x;
    const unsigned short* xr = x + row * H;
    float m = 0.f;
    for (int64_t c = threadIdx.x * 8; c < H; c += 256 * 8) m = fmaxf(m, absmax8(*(const u16x8*)(xr + c)));
    m = fmaxf(block_max_256(m, sh), round_to_bf16(1e-8f));
    for (int64_t c = threadIdx.x * 8; c < H; c += 256 * 8) fake8<DT>(*(const u16x8*)(xr + c), m, out + row * H + c);
}

template <int DT>
__global__ __launch_bounds__(256) void fake_per_group_kernel(const unsigned short* __restrict__ x,
                                                             unsigned short* __restrict__ out, int64_t total,
                                                             int group) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const bool live = e < total;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) v = *(const u16x8*)(x + e);
    float m = absmax8(v);
    const int lanes = group >> 3;
    for (int o = 1; o < lanes; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    m = fmaxf(m, round_to_bf16(1e-8f));
    if (live) fake8<DT>(v, m, out + e);
}

template <int DT>
__global__ __launch_bounds__(256) void fake_per_tensor_kernel(const unsigned short* __restrict__ x,
                                                              unsigned short* __restrict__ out,
                                                              const unsigned* __restrict__ ws, int64_t total) {
    const float amax = fmaxf(__builtin_bit_cast(float, ws[0]), round_to_bf16(1e-8f));
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; e < total; e += (int64_t)gridDim.x * 256 * 8)
        fake8<DT>(*(const u16x8*)(x + e), amax, out + e);
}

// any other group size that is a multiple of 8 (96, 384, 1024, ...): one wave per group, its lanes stride over
// the group's 16-byte pieces; same arithmetic as the kernels above.  FAKE: quantize-dequantize to bf16.
template <int DT, bool FAKE>
__global__ __launch_bounds__(256) void per_group_wave_kernel(const unsigned short* __restrict__ x,
                                                             unsigned char* __restrict__ q, float* __restrict__ scales,
                                                             unsigned short* __restrict__ out, int64_t ngroups, int group) {
    const int lane = threadIdx.x & 63;
    const int pieces = group >> 3;
    for (int64_t gi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); gi < ngroups; gi += (int64_t)gridDim.x * 4) {
        const unsigned short* xg = x + gi * group;
        float m = 0.f;
        for (int c = lane; c < pieces; c += 64) m = fmaxf(m, absmax8(*(const u16x8*)(xg + c * 8)));
        m = wave_max(m);
        if (FAKE) {
            m = fmaxf(m, round_to_bf16(1e-8f));
            for (int c = lane; c < pieces; c += 64) fake8<DT>(*(const u16x8*)(xg + c * 8), m, out + gi * group + c * 8);
        } else {
            const float sc = fmaxf(m, 1e-8f) / QMax<DT>::v;
            if (lane == 0) scales[gi] = sc;
            for (int c = lane; c < pieces; c += 64) {
                const u16x8 v = *(const u16x8*)(xg + c * 8);
                unsigned long long o8 = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = bf16_bits_to_f32(v[j]) / sc;
                    unsigned b;
                    if (DT == QL_INT8) b = (unsigned)(unsigned char)(signed char)(int)fminf(fmaxf(rintf(y), -128.f), 127.f);
                    else b = f32_to_e4m3fn(y);
                    o8 |= (unsigned long long)b << (8 * j);
                }
                *(unsigned long long*)(q + gi * group + c * 8) = o8;
            }
        }
    }
}

// group sizes that are not multiples of 8 (the reference takes any divisor of K, quantization.py:205): one thread per
// group, element by element; the same arithmetic once more
template <int DT, bool FAKE>
__global__ __launch_bounds__(256) void per_group_scalar_kernel(const unsigned short* __restrict__ x,
                                                               unsigned char* __restrict__ q, float* __restrict__ scales,
                                                               unsigned short* __restrict__ out, int64_t ngroups, int group) {
    for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * 256) {
        const unsigned short* xg = x + gi * group;
        float m = 0.f;
        for (int j = 0; j < group; ++j) m = fmaxf(m, fabsf(bf16_bits_to_f32(xg[j])));
        if (FAKE) {
            m = fmaxf(m, round_to_bf16(1e-8f));
            for (int j = 0; j < group; ++j) out[gi * group + j] = fake_one<DT>(bf16_bits_to_f32(xg[j]), m);
        } else {
            const float sc = fmaxf(m, 1e-8f) / QMax<DT>::v;
            scales[gi] = sc;
            for (int j = 0; j < group; ++j) {
                const float y = bf16_bits_to_f32(xg[j]) / sc;
                unsigned b;
                if (DT == QL_INT8) b = (unsigned)(unsigned char)(signed char)(int)fminf(fmaxf(rintf(y), -128.f), 127.f);
                else b = f32_to_e4m3fn(y);
                q[gi * group + j] = (unsigned char)b;
            }
        }
    }
}

static bool group_pow2(int g) { return g >= 8 && g <= 512 && (g % 8) == 0 && (512 % g) == 0; }   // shuffle kernels
static bool group_ok(int g) { return g >= 1; }
static int stream_grid(int64_t total) {
    int64_t b = (total / 8 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// one thread per 16-byte piece of the tiled layout: piece = (strip * K/32 + kstep) * 64 + lane,
// lane = 32 * (k half) + column
__global__ __launch_bounds__(256) void tile_weight_kernel(const unsigned char* __restrict__ qw, unsigned char* __restrict__ out,
                                                          int64_t N, int64_t K, int64_t pieces) {
    const int64_t pc = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pc >= pieces) return;
    const int lane = (int)(pc & 63);
    const int64_t blk = pc >> 6, ksteps = K / 32;
    const int64_t strip = blk / ksteps, kstep = blk % ksteps;
    const int64_t n = strip * 32 + (lane & 31), k = kstep * 32 + (lane >> 5) * 16;
    i32x4 v = {0, 0, 0, 0};
    if (n < N) v = *(const i32x4*)(qw + n * K + k);
    *(i32x4*)(out + pc * 16) = v;
}

}  // namespace ql

using namespace ql;

extern "C" {

int ql_version(void) { return 1; }
const char* ql_last_error(void) { return ql::g_err; }

int ql_quant_per_row_wide(int dtype, int in_dtype, const void* x, int64_t T, int64_t H, void* q, float* scale, void* stream) {
    const char* fn = "ql_quant_per_row_wide";
    if (in_dtype == QL_IN_BF16) return ql_quant_per_row(dtype, x, T, H, q, scale, stream);
    if ((dtype != QL_INT8 && dtype != QL_FP8) || (in_dtype != QL_IN_F16 && in_dtype != QL_IN_F32) || T < 0 || H <= 0 ||
        (T > 0 && (!x || !q || !scale))) {
        set_error("%s: bad argument (dtype=%d in_dtype=%d T=%lld H=%lld)", fn, dtype, in_dtype, (long long)T, (long long)H);
        return QL_EINVAL;
    }
    if (H % 8) { set_error("%s: H=%lld must be a multiple of 8", fn, (long long)H); return QL_ESHAPE; }
    if (T == 0) return QL_OK;
    if (T > 0x7FFFFFFF) { set_error("%s: T too large", fn); return QL_ESHAPE; }
    dim3 grid((unsigned)T), block(256);
    hipStream_t s = as_stream(stream);
#define QL_QW(DT_, IN_) hipLaunchKernelGGL((quant_per_row_wide_kernel<DT_, IN_>), grid, block, 0, s, x, (unsigned char*)q, scale, H)
    if (dtype == QL_INT8) { if (in_dtype == QL_IN_F16) QL_QW(QL_INT8, 1); else QL_QW(QL_INT8, 2); }
    else { if (in_dtype == QL_IN_F16) QL_QW(QL_FP8, 1); else QL_QW(QL_FP8, 2); }
#undef QL_QW
    QL_CHECK_LAUNCH(fn);
    return QL_OK;
}

int ql_quant_per_row(int dtype, const void* x, int64_t T, int64_t H, void* q, float* scale, void* stream) {
    if ((dtype != QL_INT8 && dtype != QL_FP8) || (T > 0 && (!x || !q || !scale)) || T < 0 || H <= 0) {
        set_error("ql_quant_per_row: bad argument (dtype=%d T=%lld H=%lld)", dtype, (long long)T, (long long)H);
        return QL_EINVAL;
    }
    if (H % 8) { set_error("ql_quant_per_row: H=%lld must be a multiple of 8", (long long)H); return QL_ESHAPE; }
    if (T == 0) return QL_OK;
    if (T > 0x7FFFFFFF) { set_error("ql_quant_per_row: T too large"); return QL_ESHAPE; }
    dim3 grid((unsigned)T), block(256);
    if (dtype == QL_INT8)
        hipLaunchKernelGGL(quant_per_row_kernel<QL_INT8>, grid, block, 0, as_stream(stream),
                           (const unsigned short*)x, (unsigned char*)q, scale, H);
    else
        hipLaunchKernelGGL(quant_per_row_kernel<QL_FP8>, grid, block, 0, as_stream(stream),
                           (const unsigned short*)x, (unsigned char*)q, scale, H);
    QL_CHECK_LAUNCH("ql_quant_per_row");
    return QL_OK;
}

int ql_row_absmax_parts(const void* x, int64_t T, int64_t H, int parts, float* out, void* stream) {
    if (T < 0 || H <= 0 || parts <= 0 || (T > 0 && (!x || !out))) { set_error("ql_row_absmax_parts: bad argument"); return QL_EINVAL; }
    if (H % ((int64_t)parts * 8)) {
        set_error("ql_row_absmax_parts: H=%lld must be a multiple of 8 * parts (parts=%d)", (long long)H, parts);
        return QL_ESHAPE;
    }
    if (T == 0) return QL_OK;
    if (T > 0x7FFFFFFF) { set_error("ql_row_absmax_parts: T too large"); return QL_ESHAPE; }
    hipLaunchKernelGGL(row_absmax_parts_kernel, dim3((unsigned)T), dim3(256), 0, as_stream(stream), (const unsigned short*)x,
                       out, H, parts);
    QL_CHECK_LAUNCH("ql_row_absmax_parts");
    return QL_OK;
}

int ql_quant_per_group(int dtype, const void* w, int64_t N, int64_t K, int group, void* q, float* scales,
                       void* stream) {
    if ((dtype != QL_INT8 && dtype != QL_FP8) || !w || !q || !scales || N < 0 || K <= 0) {
        set_error("ql_quant_per_group: bad argument");
        return QL_EINVAL;
    }
    if (!group_ok(group) || K % group) {
        set_error("ql_quant_per_group: K=%lld, group=%d unsupported (need group >= 1 and K%%group==0)",
                  (long long)K, group);
        return QL_ESHAPE;
    }
    const int64_t total = N * K;
    if (total == 0) return QL_OK;
    if (group % 8) {
        const int64_t ngroups = total / group;
        dim3 grid((unsigned)((ngroups + 255) / 256 > 4096 ? 4096 : (ngroups + 255) / 256)), block(256);
        if (dtype == QL_INT8)
            hipLaunchKernelGGL((per_group_scalar_kernel<QL_INT8, false>), grid, block, 0, as_stream(stream),
                               (const unsigned short*)w, (unsigned char*)q, scales, (unsigned short*)nullptr, ngroups, group);
        else
            hipLaunchKernelGGL((per_group_scalar_kernel<QL_FP8, false>), grid, block, 0, as_stream(stream),
                               (const unsigned short*)w, (unsigned char*)q, scales, (unsigned short*)nullptr, ngroups, group);
        QL_CHECK_LAUNCH("ql_quant_per_group");
        return QL_OK;
    }
    if (!group_pow2(group)) {
        const int64_t ngroups = total / group;
        dim3 grid((unsigned)((ngroups + 3) / 4 > 8192 ? 8192 : (ngroups + 3) / 4)), block(256);
        if (dtype == QL_INT8)
            hipLaunchKernelGGL((per_group_wave_kernel<QL_INT8, false>), grid, block, 0, as_stream(stream),
                               (const unsigned short*)w, (unsigned char*)q, scales, (unsigned short*)nullptr, ngroups, group);
        else
            hipLaunchKernelGGL((per_group_wave_kernel<QL_FP8, false>), grid, block, 0, as_stream(stream),
                               (const unsigned short*)w, (unsigned char*)q, scales, (unsigned short*)nullptr, ngroups, group);
        QL_CHECK_LAUNCH("ql_quant_per_group");
        return QL_OK;
    }
    const int64_t blocks = (total / 8 + 255) / 256;
    if (blocks > 0x7FFFFFFF) { set_error("ql_quant_per_group: tensor too large"); return QL_ESHAPE; }
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == QL_INT8)
        hipLaunchKernelGGL(quant_per_group_kernel<QL_INT8>, grid, block, 0, as_stream(stream),
                           (const unsigned short*)w, (unsigned char*)q, scales, total, group);
    else
        hipLaunchKernelGGL(quant_per_group_kernel<QL_FP8>, grid, block, 0, as_stream(stream),
                           (const unsigned short*)w, (unsigned char*)q, scales, total, group);
    QL_CHECK_LAUNCH("ql_quant_per_group");
    return QL_OK;
}

int ql_quant_per_tensor(int dtype, const void* w, int64_t N, int64_t K, void* q, float* scale, void* workspace,
                        void* stream) {
    if ((dtype != QL_INT8 && dtype != QL_FP8) || !w || !q || !scale || !workspace || N <= 0 || K <= 0) {
        set_error("ql_quant_per_tensor: bad argument");
        return QL_EINVAL;
    }
    const int64_t total = N * K;
    if (total % 8) { set_error("ql_quant_per_tensor: N*K must be a multiple of 8"); return QL_ESHAPE; }
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(workspace, 0, 16, s) != hipSuccess) { set_error("ql_quant_per_tensor: memset failed"); return QL_ELAUNCH; }
    dim3 grid(stream_grid(total)), block(256);
    hipLaunchKernelGGL(absmax_kernel, grid, block, 0, s, (const unsigned short*)w, total, (unsigned*)workspace);
    if (dtype == QL_INT8)
        hipLaunchKernelGGL(quant_per_tensor_kernel<QL_INT8>, grid, block, 0, s, (const unsigned short*)w,
                           (unsigned char*)q, scale, (const unsigned*)workspace, total);
    else
        hipLaunchKernelGGL(quant_per_tensor_kernel<QL_FP8>, grid, block, 0, s, (const unsigned short*)w,
                           (unsigned char*)q, scale, (const unsigned*)workspace, total);
    QL_CHECK_LAUNCH("ql_quant_per_tensor");
    return QL_OK;
}

int ql_fake_quant(int dtype, int granularity, const void* w, int64_t N, int64_t K, int group, void* out,
                  void* workspace, void* stream) {
    if ((dtype != QL_INT8 && dtype != QL_FP8) || !w || !out || N <= 0 || K <= 0) {
        set_error("ql_fake_quant: bad argument");
        return QL_EINVAL;
    }
    if (K % 8) { set_error("ql_fake_quant: K must be a multiple of 8"); return QL_ESHAPE; }
    hipStream_t s = as_stream(stream);
    const int64_t total = N * K;
    const unsigned short* x = (const unsigned short*)w;
    unsigned short* o = (unsigned short*)out;
    if (granularity == QL_PER_ROW) {
        if (N > 0x7FFFFFFF) { set_error("ql_fake_quant: N too large"); return QL_ESHAPE; }
        dim3 grid((unsigned)N), block(256);
        if (dtype == QL_INT8) hipLaunchKernelGGL(fake_per_row_kernel<QL_INT8>, grid, block, 0, s, x, o, K);
        else hipLaunchKernelGGL(fake_per_row_kernel<QL_FP8>, grid, block, 0, s, x, o, K);
    } else if (granularity == QL_PER_GROUP) {
        if (!group_ok(group) || K % group) { set_error("ql_fake_quant: group=%d unsupported for K=%lld", group, (long long)K); return QL_ESHAPE; }
        const int64_t blocks = (total / 8 + 255) / 256;
        if (blocks > 0x7FFFFFFF) { set_error("ql_fake_quant: tensor too large"); return QL_ESHAPE; }
        dim3 grid((unsigned)blocks), block(256);
        if (group % 8) {
            const int64_t ngroups = total / group;
            dim3 sgrid((unsigned)((ngroups + 255) / 256 > 4096 ? 4096 : (ngroups + 255) / 256));
            if (dtype == QL_INT8) hipLaunchKernelGGL((per_group_scalar_kernel<QL_INT8, true>), sgrid, block, 0, s, x, (unsigned char*)nullptr, (float*)nullptr, o, ngroups, group);
            else hipLaunchKernelGGL((per_group_scalar_kernel<QL_FP8, true>), sgrid, block, 0, s, x, (unsigned char*)nullptr, (float*)nullptr, o, ngroups, group);
        }
        else if (!group_pow2(group)) {
            const int64_t ngroups = total / group;
            dim3 wgrid((unsigned)((ngroups + 3) / 4 > 8192 ? 8192 : (ngroups + 3) / 4));
            if (dtype == QL_INT8) hipLaunchKernelGGL((per_group_wave_kernel<QL_INT8, true>), wgrid, block, 0, s, x, (unsigned char*)nullptr, (float*)nullptr, o, ngroups, group);
            else hipLaunchKernelGGL((per_group_wave_kernel<QL_FP8, true>), wgrid, block, 0, s, x, (unsigned char*)nullptr, (float*)nullptr, o, ngroups, group);
        }
        else if (dtype == QL_INT8) hipLaunchKernelGGL(fake_per_group_kernel<QL_INT8>, grid, block, 0, s, x, o, total, group);
        else hipLaunchKernelGGL(fake_per_group_kernel<QL_FP8>, grid, block, 0, s, x, o, total, group);
    } else if (granularity == QL_PER_TENSOR) {
        if (!workspace) { set_error("ql_fake_quant: per-tensor needs a workspace"); return QL_EINVAL; }
        if (hipMemsetAsync(workspace, 0, 16, s) != hipSuccess) { set_error("ql_fake_quant: memset failed"); return QL_ELAUNCH; }
        dim3 grid(stream_grid(total)), block(256);
        hipLaunchKernelGGL(absmax_kernel, grid, block, 0, s, x, total, (unsigned*)workspace);
        if (dtype == QL_INT8) hipLaunchKernelGGL(fake_per_tensor_kernel<QL_INT8>, grid, block, 0, s, x, o, (const unsigned*)workspace, total);
        else hipLaunchKernelGGL(fake_per_tensor_kernel<QL_FP8>, grid, block, 0, s, x, o, (const unsigned*)workspace, total);
    } else {
        set_error("ql_fake_quant: unknown granularity %d", granularity);
        return QL_EINVAL;
    }
    QL_CHECK_LAUNCH("ql_fake_quant");
    return QL_OK;
}

int ql_tile_weight(const void* qw, void* out, int64_t N, int64_t K, void* stream) {
    if (!qw || !out || N <= 0 || K <= 0 || qw == out) { set_error("ql_tile_weight: bad argument"); return QL_EINVAL; }
    if (K % 32) { set_error("ql_tile_weight: K=%lld must be a multiple of 32", (long long)K); return QL_ESHAPE; }
    const int64_t pieces = ((N + 31) / 32) * (K / 32) * 64;          // 16-byte pieces of the output
    if (pieces > 0x7FFFFFFFll * 256) { set_error("ql_tile_weight: too large"); return QL_ESHAPE; }
    hipLaunchKernelGGL(tile_weight_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (const unsigned char*)qw, (unsigned char*)out, N, K, pieces);
    QL_CHECK_LAUNCH("ql_tile_weight");
    return QL_OK;
}

}  // extern "C"
