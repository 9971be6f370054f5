/*
 * libqlinear - MI355X (gfx950) quantized-linear kernels behind a plain C ABI.
 *
 * Every entry point replaces one piece of the reference's Python/Triton quantized-linear path
 * (citations are file:line in the reference checkout, nanovllm/utils/quantization.py unless a
 * file is named).  Conventions:
 *   - all pointers are DEVICE pointers owned by the caller; matrices are row-major and dense;
 *   - bf16 = 16-bit brain float bit patterns, fp8 = OCP e4m3fn bit patterns, scales = float32;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); nothing here
 *     synchronises the host, allocates or frees device memory, so every call may be captured into
 *     a hipGraph;
 *   - return value 0 = launched; a negative QL_E* code = rejected before any launch, with a
 *     message available from ql_last_error() (thread-local).  The reference asserts instead.
 */
#ifndef QLINEAR_H
#define QLINEAR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* storage type of quantized values: torch.int8 / torch.float8_e4m3fn (quantization.py:151) */
#define QL_INT8 0
#define QL_FP8 1
/* OR-ed into the weight dtype of the GEMM entry points: qw is in the tiled layout of ql_tile_weight */
#define QL_TILED 16

/* granularity of the weight scales */
#define QL_PER_ROW 0    /* sw[N]            (QLinear,       :349-364) */
#define QL_PER_TENSOR 1 /* sw[1]            (QTensorLinear, :367-382) */
#define QL_PER_GROUP 2  /* sw[N, K/group]   (QGroupLinear,  :385-401) */

#define QL_OK 0
#define QL_EINVAL (-1)      /* bad enum / null pointer / negative size */
#define QL_ESHAPE (-2)      /* shape not supported by the kernels (K % 32, group size, ...) */
#define QL_EWORKSPACE (-3)  /* workspace too small */
#define QL_ELAUNCH (-4)     /* HIP reported a launch error */

int ql_version(void);
const char* ql_last_error(void);

/* ---- quantizers ------------------------------------------------------------------------- */

/* per_row_quant + _per_row_quant_kernel (:13-51, :147-163).  x[T,H] bf16 -> q[T,H] (1 byte each),
 * scale[T] f32.  row_max = max(max|x|,1e-8); scale = row_max/Q; q = cast(x * (Q/row_max)) with
 * roundf (half away from zero) for int8 and RTNE for fp8; Q = 127 | 448.  Used on weights at
 * conversion time and on activations at every W8A8 forward.  Requires H % 8 == 0. */
int ql_quant_per_row(int dtype, const void* x_bf16, int64_t T, int64_t H, void* q, float* scale,
                     void* stream);

/* per_row_quant on activations that are not bf16: the reference's kernel converts whatever it loads to f32
 * (`.to(tl.float32)`, quantization.py:27), so fp16 and fp32 inputs go through the same arithmetic on the widened
 * values.  in_dtype: QL_IN_BF16 (= ql_quant_per_row), QL_IN_F16, QL_IN_F32. */
#define QL_IN_BF16 0
#define QL_IN_F16 1
#define QL_IN_F32 2
int ql_quant_per_row_wide(int dtype, int in_dtype, const void* x, int64_t T, int64_t H, void* q, float* scale,
                          void* stream);

/* per_tensor_quant (:166-191).  One scale for the whole [N,K] bf16 tensor; amax, scale and x/scale
 * are rounded to bf16 as the reference's bf16 tensor arithmetic does; int8 uses round-half-even +
 * clamp(-128,127).  `scale` receives the (bf16-valued) scale as f32.  `workspace` >= 16 bytes. */
int ql_quant_per_tensor(int dtype, const void* w_bf16, int64_t N, int64_t K, void* q, float* scale,
                        void* workspace, void* stream);

/* per_group_quant (:194-234).  scales[N, K/group] f32 = max(amax_group,1e-8)/Q; q = x/scale with
 * round-half-even + clamp (int8) or RTNE cast (fp8).  Any group size that divides K, as the reference (:205):
 * multiples of 8 on the vector kernels, other sizes one thread per group.  The GEMM below likewise takes any divisor
 * of K (fused forms: multiples of 32). */
int ql_quant_per_group(int dtype, const void* w_bf16, int64_t N, int64_t K, int group, void* q,
                       float* scales, void* stream);

/* _fake_per_block_quant (:404-431) and its wrappers (:434-445): quantize-dequantize, bf16 in,
 * bf16 out (out may alias w).  granularity QL_PER_TENSOR (blocks -1,-1), QL_PER_ROW (1,-1) or
 * QL_PER_GROUP (1,group).  `workspace` >= 16 bytes (used by QL_PER_TENSOR only). */
int ql_fake_quant(int dtype, int granularity, const void* w_bf16, int64_t N, int64_t K, int group,
                  void* out_bf16, void* workspace, void* stream);

/* Re-arrange a quantized weight qw[N,K] (1 byte per element, row-major) for the GEMM kernels: the 32 columns x
 * 32 k that one wave loads at a time become one contiguous KiB in lane order,
 *   out[((n/32) * (K/32) + k/32) * 1024 + ((k%32)/16 * 32 + n%32) * 16 + k%16] = qw[n][k],
 * rows n >= N of the last strip zero.  `out` holds ceil(N/32)*32*K bytes and must not alias qw.  Done once after
 * quantizing; the GEMMs then take the buffer with QL_TILED OR-ed into their dtype argument.  A wave load touches
 * 8 full 128-byte lines instead of 32 quarter-used ones, which is what bounds these kernels.  K % 32 == 0. */
int ql_tile_weight(const void* qw, void* out, int64_t N, int64_t K, void* stream);

/* ---- fused dequant GEMMs ---------------------------------------------------------------- */

/* Bytes of scratch the GEMM entry points need for an [M,N,K] problem with the configuration the
 * library would pick right now (64 KiB of per-tile ticket counters + split-K partial slabs; 0 when
 * K is not split).  `granularity` is the value that will be passed to ql_gemm_w8a16 (use
 * QL_PER_ROW for ql_gemm_w8a8).  The workspace must be ZERO-FILLED once when it is allocated: the
 * first 64 KiB hold the counters, which every call leaves at zero again.  Calls that share a
 * workspace must be ordered on one stream. */
size_t ql_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int granularity);
/* The largest ql_gemm_workspace_bytes(M, N, K, granularity) over 1 <= M <= max_M: what to reserve once for a
 * weight whose calls may bring any number of rows up to max_M (e.g. the scheduler's token budget) when the
 * workspace must never be re-allocated afterwards (pointers baked into captured hipGraphs). */
size_t ql_gemm_workspace_max_bytes(int64_t max_M, int64_t N, int64_t K, int granularity);

/* Weight-only forward: y[M,N] bf16 = x[M,K] bf16 . dequant(qw[N,K])^T (+ bias[N] bf16).
 *   QL_PER_ROW    y = bf16( (sum_k x*q) * sw[n] + bias )                 - the W8A16 counterpart of
 *                 QLinear; numerics of F.linear(x, fake_per_row_quant(W)) (:438-439) without the
 *                 extra bf16 rounding of the dequantised weight.
 *   QL_PER_TENSOR y = bf16( bf16(sum_k x*(q*sw)) + bias )                - per_tensor_matmul (:237-262)
 *   QL_PER_GROUP  y = bf16( bf16(sum_g sw[n,g] * sum_{k in g} x*q) + bias ) - per_group_matmul (:265-307)
 * int8/fp8 weights are widened to bf16 in registers (exact) and contracted on bf16 MFMA with f32
 * accumulation; nothing is dequantised to memory.  Requires K % 32 == 0; per-group additionally
 * K % group == 0.  A group's partial sum is folded with its scale at 32-k steps, so group sizes that are multiples
 * of 32 run on the MFMA kernels; any other divisor of K (the reference takes them all, :205) runs this plain form
 * on an f32 kernel - exact per-group arithmetic, not tuned - and has none of the fused forms below (QL_ESHAPE). */
int ql_gemm_w8a16(int wdtype, int granularity, const void* x_bf16, const void* qw, const float* sw,
                  const void* bias_bf16, void* y_bf16, int64_t M, int64_t N, int64_t K, int group,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ql_gemm_w8a16 for callers that cannot allocate device memory: code under hipGraph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  (No counterpart in the reference, which allocates its outputs and
 * Triton scratch freely.)  Same arguments, kernels and results, but the split-K workspace is a fixed
 * QL_NOALLOC_SCRATCH_BYTES array inside the library's code object, one copy per GPU, held until the process
 * ends; the call neither allocates nor frees anything.  A problem whose plan needs more than that is rejected
 * with QL_EWORKSPACE (ql_gemm_workspace_bytes tells in advance).  The array keeps the previous call's contents;
 * the call clears the ticket counters it will use (a small launch on `stream` in front of the GEMM).  All calls on one GPU share
 * the array: calls that overlap in time on one GPU (different streams, different threads) race.  The first call
 * on each device looks up the array's address; make it outside stream capture.  The call plans with the built-in
 * heuristics only (entries of the plan table, ql_gemm_plan_set, may split K further than the array holds). */
#define QL_NOALLOC_SCRATCH_BYTES (16u << 20)
int ql_gemm_w8a16_noalloc(int wdtype, int granularity, const void* x_bf16, const void* qw, const float* sw,
                          const void* bias_bf16, void* y_bf16, int64_t M, int64_t N, int64_t K, int group,
                          void* stream);

/* _per_row_post_process_mm as launched by per_row_matmul (:54-144, :324-345):
 * y[M,N] bf16 = bf16( (sum_k qx[m,k]*qw[n,k]) * sx[m] * sw[n] + bias[n] ), both operands int8
 * (exact int32 accumulation on i8 MFMA) or both fp8-e4m3fn (f32 accumulation on fp8 MFMA).
 * The bias is added per output column (the reference's kernel indexes it by tile row, :133).
 * Requires K % 32 == 0. */
int ql_gemm_w8a8(int dtype, const void* qx, const float* sx, const void* qw, const float* sw,
                 const void* bias_bf16, void* y_bf16, int64_t M, int64_t N, int64_t K,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Gate/up projection with the SiLU-and-multiply of the MLP fused into the GEMM epilogue
 * (Qwen3MLP.forward, nanovllm/models/qwen3.py:116-118 = gate_up_proj then SiluAndMul,
 * nanovllm/layers/activation.py:11-14):  y[M, N/2] bf16 = silu(g) * u with g, u the bf16-rounded
 * outputs ql_gemm_w8a16 / ql_gemm_w8a8 would have stored, so the result is bit-identical to the
 * GEMM followed by ql_silu_mul.  The weight must be PRE-ARRANGED by the caller: with I = N/2 and the
 * usual packed rows (gate rows [0,I), up rows [I,2I)), row 64*s + j of qw (and of sw / the group
 * scales / bias) holds gate row 32*s + j for j < 32 and up row 32*s + j - 32 for j >= 32.
 * Requires N % 64 == 0.  Workspace as for the plain GEMMs (ql_gemm_workspace_bytes covers this form): when K is
 * split across workgroups, gate and up meet in the epilogue of the tile's last workgroup, which holds the whole
 * reduced tile. */
int ql_gemm_w8a16_silu(int wdtype, int granularity, const void* x_bf16, const void* qw, const float* sw,
                       const void* bias_bf16, void* y_bf16, int64_t M, int64_t N, int64_t K, int group,
                       void* workspace, size_t workspace_bytes, void* stream);
int ql_gemm_w8a8_silu(int dtype, const void* qx, const float* sx, const void* qw, const float* sw,
                      const void* bias_bf16, void* y_bf16, int64_t M, int64_t N, int64_t K, void* workspace,
                      size_t workspace_bytes, void* stream);

/* per_row_matmul (nanovllm/utils/quantization.py:310-346) with its per_row_quant(x) (:323) inside the GEMM: x is the
 * bf16 activation matrix, rowmax_part f32 [M, nparts] holds partial |max| values of every row over ANY split of its
 * K columns (what the workgroups of the producing kernel each see: ql_gemm_w8a8_silu and the attention kernels can
 * emit them; ql_row_absmax_parts computes them from x).  The kernel takes their maximum, clamps it at 1e-8 and
 * quantizes the activations while it stages them, with exactly the arithmetic of ql_quant_per_row; the result is
 * bit-identical to ql_quant_per_row followed by ql_gemm_w8a8.  Workspace as for ql_gemm_w8a8.
 * ql_row_absmax_parts: out f32 [T, parts], out[t][i] = max |x[t, i*H/parts : (i+1)*H/parts]|; H % (8 * parts) == 0. */
int ql_row_absmax_parts(const void* x_bf16, int64_t T, int64_t H, int parts, float* out, void* stream);
/* ql_gemm_w8a8_silu that also emits what a ql_gemm_w8a8_dynq consumer of y needs: rowmax_part_out f32 [M, N/64],
 * the |max| of every row of y over each of its 32-column strips (Qwen3MLP.forward, nanovllm/models/qwen3.py:116-118:
 * gate_up_proj -> SiluAndMul -> down_proj, whose per_row_quant(x) of quantization.py:323 then needs no launch). */
int ql_gemm_w8a8_silu_rowmax(int dtype, const void* qx, const float* sx, const void* qw, const float* sw,
                             const void* bias_bf16, void* y_bf16, float* rowmax_part_out, int64_t M, int64_t N, int64_t K,
                             void* workspace, size_t workspace_bytes, void* stream);
int ql_gemm_w8a8_dynq(int dtype, const void* x_bf16, const float* rowmax_part, int nparts, const void* qw,
                      const float* sw, const void* bias_bf16, void* y_bf16, int64_t M, int64_t N, int64_t K,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Weight-only GEMM whose split-K reduction is left to the NEXT launch.  For the projections that feed a residual
 * add + RMSNorm (o_proj, down_proj: Qwen3DecoderLayer.forward, nanovllm/models/qwen3.py:153-162) the K dimension is
 * cut into ql_gemm_partial_slices(M, N, K, granularity) slices, slice s stores its UNSCALED f32 partial sums at
 * partials[s][M][N], and ql_add_rmsnorm_partial adds the slices in slice order, applies the GEMM's own epilogue
 * (x = bf16(sum * sw[n] + bias), formulas of ql_gemm_w8a16) and continues as ql_add_rmsnorm(x, residual, ...).
 * The pair is bit-identical to ql_gemm_w8a16 (same tile, same slices) followed by ql_add_rmsnorm; it removes the
 * ticket + second pass from the GEMM's tail, so K can be cut finely enough to occupy the chip at decode sizes.
 * `slices` must be the value ql_gemm_partial_slices returns for the same arguments; no workspace.  sw / bias of
 * ql_add_rmsnorm_partial are the GEMM's (sw ignored for QL_PER_GROUP: group scales are applied inside the GEMM);
 * residual and residual_out are required. */
int ql_gemm_partial_slices(int64_t M, int64_t N, int64_t K, int granularity);
int ql_gemm_w8a16_partial(int wdtype, int granularity, const void* x_bf16, const void* qw, const float* sw, int64_t M,
                          int64_t N, int64_t K, int group, float* partials, int slices, void* stream);
int ql_add_rmsnorm_partial(const float* partials, int slices, int granularity, const float* sw, const void* bias_bf16,
                           const void* residual_bf16, const void* weight_bf16, float eps, void* out_bf16,
                           void* residual_out_bf16, int64_t T, int64_t H, void* stream);

/* The same pair for the W8A8 form (per_row_matmul, nanovllm/utils/quantization.py:310-346): the GEMM stores the raw
 * slices (int32 sums for int8 x int8, f32 for fp8 x fp8; [slices][M][N], 4 bytes per element) and the norm launch
 * adds them (integers exactly), applies (sum * sx[m]) * sw[n] (+ bias) as ql_gemm_w8a8's epilogue does, and continues
 * as ql_add_rmsnorm (out_dtype < 0: bf16 `out`) or ql_add_rmsnorm_quant (out_dtype = QL_INT8 / QL_FP8: q_out +
 * scale_out for the next W8A8 linear).  Bit-identical to ql_gemm_w8a8 followed by the plain norm launch. */
int ql_gemm_w8a8_partial(int dtype, const void* qx, const void* qw, int64_t M, int64_t N, int64_t K, void* partials,
                         int slices, void* stream);
int ql_add_rmsnorm_partial_w8a8(int out_dtype, const void* partials, int slices, int acc_is_int, const float* sx,
                                const float* sw, const void* bias_bf16, const void* residual_bf16,
                                const void* weight_bf16, float eps, void* out_bf16, void* residual_out_bf16,
                                void* q_out, float* scale_out, int64_t T, int64_t H, void* stream);

/* The qkv projection with everything between it and attention fused into the GEMM epilogue
 * (Qwen3Attention.forward, nanovllm/models/qwen3.py:80-88: qkv_proj, split, q_norm / k_norm, rotary_emb, and the
 * store_kvcache of nanovllm/layers/attention.py:10-40, 52-58): the [M, N] projection is never written; every
 * 128-column head of it goes through exactly the arithmetic of ql_qknorm_rope_store (whose arguments these are)
 * on the bf16 values ql_gemm_w8a16 / ql_gemm_w8a8 would have stored, so q_out, k_out, v_out and the cache rows are
 * bit-identical to the GEMM followed by that operator when both run the same tile.  N must equal
 * (Hq + 2 Hkv) * 128.  Workspace as for the plain GEMMs (ql_gemm_workspace_bytes covers both). */
int ql_gemm_w8a16_rope(int wdtype, int granularity, const void* x_bf16, const void* qw, const float* sw,
                       const void* bias_bf16, int64_t M, int64_t N, int64_t K, int group, const int64_t* positions,
                       const float* cos_sin, const void* q_norm_w, const void* k_norm_w, float eps, void* q_out,
                       void* k_out, void* v_out, void* k_cache, void* v_cache, const int* slot_mapping, int Hq,
                       int Hkv, int head_dim, void* workspace, size_t workspace_bytes, void* stream);
int ql_gemm_w8a8_rope(int dtype, const void* qx, const float* sx, const void* qw, const float* sw,
                      const void* bias_bf16, int64_t M, int64_t N, int64_t K, const int64_t* positions,
                      const float* cos_sin, const void* q_norm_w, const void* k_norm_w, float eps, void* q_out,
                      void* k_out, void* v_out, void* k_cache, void* v_cache, const int* slot_mapping, int Hq,
                      int Hkv, int head_dim, void* workspace, size_t workspace_bytes, void* stream);

/* ---- row-parallel all-reduce over peer-mapped buffers (one node, up to 8 ranks) ------------------------
 * RowParallelLinear.forward under tensor parallelism (nanovllm/layers/linear.py:149-153: F.linear on the K shard,
 * then dist.all_reduce(y)) followed by the residual add + RMSNorm that consumes every o_proj / down_proj output
 * (Qwen3DecoderLayer.forward, nanovllm/models/qwen3.py:153-162; RMSNorm.add_rms_forward, layernorm.py:28-40), in ONE
 * launch and without the collective library: each rank pushes its bf16 partial rows into a slot of every peer's
 * buffer, raises a per-row flag there, waits (bounded) for its peers' flags and sums the `world` partials in rank
 * order in f32 from local memory:  y = bf16(sum_r partial_r)  - identical bits on every rank and on every replay -
 * then exactly ql_add_rmsnorm(y, residual, weight).  weight == NULL: the plain all-reduce, y_out only.
 *
 * Set-up, once per process and communicator: ql_comm_alloc a buffer of ql_comm_buffer_bytes(world, max_rows, hidden)
 * (zero-filled; device memory other agents may write), ql_comm_export its 64-byte handle, exchange the handles
 * through any channel (the process group that already exists), ql_comm_import every peer's, and pass the `world`
 * device pointers (own buffer at index rank) to every call.  Calls of one communicator must be issued in the same
 * order on every rank and on one stream per rank; the sequence number lives in the buffer and is advanced by the
 * launches themselves, so calls captured into a hipGraph replay correctly.  T <= max_rows (decode-sized messages;
 * larger ones belong to the collective library).  spin_limit = polls a wait may make before it gives up (0: a few
 * seconds' worth): a peer that never arrives makes the result wrong, never hangs the device; ql_comm_status
 * (synchronous) returns how many waits have timed out since allocation. */
#define QL_COMM_HANDLE_BYTES 64
size_t ql_comm_buffer_bytes(int world, int64_t max_rows, int64_t hidden);
int ql_comm_alloc(size_t bytes, void** buf);
int ql_comm_free(void* buf);
int ql_comm_export(void* buf, void* handle64);
int ql_comm_import(const void* handle64, void** peer_buf);
int ql_comm_unimport(void* peer_buf);
int ql_comm_status(const void* buf, unsigned* epoch, unsigned* timeouts);
int ql_allreduce_add_rmsnorm(int rank, int world, void* const* bufs, int64_t max_rows, int64_t hidden,
                             const void* x_partial_bf16, const void* residual_bf16, const void* weight_bf16, float eps,
                             void* y_out_bf16, void* out_bf16, void* residual_out_bf16, int64_t T, unsigned spin_limit,
                             void* stream);

/* Tuning hook for benchmarks/tests: force the tile configuration of the next GEMM calls made by
 * this thread.  mt = 32-row fragments per wave (1,2,4,8), wn x wk = how the block's 4 waves split
 * N and K ((4,1),(2,2),(1,4)), cp = 32-k steps a wave loads per chunk (2,4,8), split_k = K slices
 * across workgroups, or, negative, the stream-K schedule on -split_k workgroups (weight-only GEMMs on the
 * tiles built for it; see ql_gemm_plan_set).  mt = wn = wk = cp = 0 restores the built-in tile heuristic,
 * split_k = 0 the split heuristic.  Returns QL_EINVAL for a tile that is not built.  A forced split needs the
 * workspace size that ql_gemm_workspace_bytes reports while the configuration is forced. */
int ql_gemm_force_config(int mt, int wn, int wk, int cp, int split_k);

/* Plan table (no counterpart in the reference: its Triton kernel runs one fixed 128 x 128 x 128 block
 * configuration, nanovllm/utils/quantization.py:324-330): tile and split choices measured on the
 * device for a given launch form, (N, K) and range of row counts m_lo..m_hi take precedence over the
 * built-in heuristics for every later GEMM call of the process (ql_gemm_force_config still wins).
 * `form` is a combination of the bits below: the scale granularity, the epilogue and the arithmetic of
 * the launch (0 = plain per-row / per-tensor W8A16).  With QL_PLAN_PARTIAL the entry is for the deferred
 * split-K form (ql_gemm_*_partial; split_k = 1 there means "do not defer").  Entries of one form and
 * shape must not overlap; a tile that is not built for the form, or a split the form cannot run, is
 * QL_EINVAL.  split_k = -G (8 <= G <= 4096) selects the STREAM-K schedule for the weight-only forms: a 1-D grid
 * of G workgroups (normally one or two per CU) shares the (output tile, k chunk) units of the problem evenly,
 * each workgroup taking one contiguous range, and tiles that several workgroups share are combined like split-K
 * tiles (ascending k order, bit-reproducible); it keeps every CU busy when the tile count does not fill whole
 * rounds of the chip (152 tiles of 128 x 256 for a Qwen3-4B gate/up projection at 256 rows).  ql_gemm_workspace_bytes / _max_bytes see the table, so set entries before sizing a
 * workspace.  ql_gemm_plan_query writes {mt, wn, wk, cp, split_k} of the plan a call would run to out[5]. */
#define QL_PLAN_GROUP 1    /* per-group scales */
#define QL_PLAN_SILU 2     /* ql_gemm_*_silu */
#define QL_PLAN_ROPE 4     /* ql_gemm_*_rope */
#define QL_PLAN_W8A8 8     /* ql_gemm_w8a8* (else ql_gemm_w8a16*) */
#define QL_PLAN_PARTIAL 16 /* ql_gemm_*_partial */
int ql_gemm_plan_set(int form, int64_t m_lo, int64_t m_hi, int64_t N, int64_t K, int mt, int wn, int wk, int cp,
                     int split_k);
void ql_gemm_plan_clear(void);
int ql_gemm_plan_count(void);
int ql_gemm_plan_query(int form, int64_t M, int64_t N, int64_t K, int* out);

/* ---- operators between the quantized GEMMs of a decoder layer (one launch each) ------------- */

/* RMSNorm.rms_forward / add_rms_forward (nanovllm/layers/layernorm.py:16-40): with `residual`,
 * s = x + residual in f32, residual_out = bf16(s); out = bf16( bf16(s * rsqrt(mean(s^2)+eps)) * w ).
 * residual may be NULL (plain norm; residual_out ignored).  H % 8 == 0, H <= 8192. */
int ql_add_rmsnorm(const void* x_bf16, const void* residual_bf16, const void* weight_bf16, float eps,
                   void* out_bf16, void* residual_out_bf16, int64_t T, int64_t H, void* stream);

/* ql_add_rmsnorm with the dynamic per-row activation quantization of the W8A8 linear that consumes the
 * normalised row folded in (per_row_quant on x at nanovllm/utils/quantization.py:323, kernel :13-51):
 * q[T,H] (dtype QL_INT8 / QL_FP8) and scale[T] are bit-identical to ql_add_rmsnorm followed by
 * ql_quant_per_row; the bf16 normalised row is never written.  residual as above. */
int ql_add_rmsnorm_quant(int dtype, const void* x_bf16, const void* residual_bf16, const void* weight_bf16, float eps,
                         void* q, float* scale, void* residual_out_bf16, int64_t T, int64_t H, void* stream);

/* SiluAndMul.forward (nanovllm/layers/activation.py:11-14): out[T,I] = silu(x[:, :I]) * x[:, I:]. */
int ql_silu_mul(const void* x_bf16, void* out_bf16, int64_t T, int64_t I, void* stream);

/* The q/k path of Qwen3Attention.forward between qkv_proj and attention
 * (nanovllm/models/qwen3.py:81-88, layers/rotary_embedding.py:7-48, layers/attention.py:10-40):
 * split qkv[T, (Hq+2Hkv)*128], per-head RMSNorm of q and k (weights may be NULL = no norm),
 * rotate-half RoPE from cos_sin[pos, 0:64 | 64:128] (f32), q -> q_out[T,Hq,128]; k, v -> k_out/v_out
 * (optional, contiguous, for prefill attention) and scattered to k_cache/v_cache at row
 * slot_mapping[t] (skipped for -1 or a NULL slot_mapping).  head_dim must be 128.  positions may be NULL:
 * cos_sin then holds one row per token, [T, 128], already gathered (one gather per step serves every layer and
 * takes the position -> table-row dependency out of each call). */
int ql_qknorm_rope_store(const void* qkv_bf16, const int64_t* positions, const float* cos_sin,
                         const void* q_norm_w, const void* k_norm_w, float eps, void* q_out, void* k_out,
                         void* v_out, void* k_cache, void* v_cache, const int* slot_mapping, int64_t T,
                         int Hq, int Hkv, int head_dim, void* stream);

/* Greedy sampler (nanovllm/layers/sampler.py:10-12): out[t] = argmax_v logits[t, v] over bf16 logits [T, V],
 * the smallest index among equal maxima.  V % 8 == 0. */
int ql_argmax_rows(const void* logits_bf16, int64_t* out, int64_t T, int64_t V, void* stream);

/* Output projection (ParallelLMHead.forward, nanovllm/layers/embed_head.py:56-66 without the gather: the caller
 * passes the rows it wants logits for): logits[M, V] = bf16(x[M, K] @ w[V, K]^T), bf16 operands (the table is the
 * unquantized embedding matrix, as in the reference), f32 accumulation.  K % 32 == 0. */
int ql_lmhead_logits(const void* x_bf16, const void* w_bf16, void* logits_bf16, int64_t M, int64_t V, int64_t K,
                     void* stream);

/* The same projection with the greedy sampler (nanovllm/layers/sampler.py:10-12) in its epilogue:
 * tokens[m] = ql_argmax_rows(ql_lmhead_logits(x, w))[m], bit for bit (same kernel arithmetic, same tie and NaN
 * rules), without the logits ever reaching memory.  max_out (f32 [M], may be NULL) receives the winning logit
 * (-inf for a row without a finite maximum): what vocabulary-parallel ranks exchange instead of the logits the
 * reference gathers (embed_head.py:62-65).  workspace: at least ql_lmhead_workspace_bytes(M, V) bytes of device
 * memory, need not be cleared. */
size_t ql_lmhead_workspace_bytes(int64_t M, int64_t V);
int ql_lmhead_argmax(const void* x_bf16, const void* w_bf16, int64_t* tokens, float* max_out, int64_t M, int64_t V,
                     int64_t K, void* workspace, size_t workspace_bytes, void* stream);

/* Prefill attention: causal self-attention over packed variable-length sequences, the
 * flash_attn_varlen_func call of Attention.forward's prefill branch without a prefix cache
 * (nanovllm/layers/attention.py:64-69).  q [T,Hq,128], k/v [T,Hkv,128], out [T,Hq,128] bf16;
 * cu_seqlens int32 [num_seqs+1] in device memory, sequence s owns rows cu[s] .. cu[s+1]-1;
 * max_seqlen >= the longest sequence (sizes the grid).  out = softmax(scale * q k^T + causal) v with
 * f32 accumulation, probabilities rounded to bf16 before the second product.  Any Hq % Hkv == 0 (a kv head's query heads share
 * a workgroup for the ratios 1, 2, 4; other ratios are dealt to workgroups of 4, 2 or 1 heads). */
int ql_attn_prefill(const void* q, const void* k, const void* v, const int* cu_seqlens, void* out,
                    float scale, int num_seqs, int max_seqlen, int Hq, int Hkv, int head_dim, void* stream);

/* Single-query attention over a paged KV cache: the decode half of Attention.forward
 * (nanovllm/layers/attention.py:59-75, flash_attn_with_kvcache there).  q/out [B,Hq,128];
 * caches [num_blocks, block_size, Hkv, 128]; block_tables [B, max_blocks]; context_lens [B]
 * (0 = padded row, output 0).  Any Hq % Hkv == 0 (ratios other than 1, 2, 4 re-read a kv head's K/V once per
 * workgroup of 4, 2 or 1 query heads).  One workgroup per (sequence, kv head, 512-token
 * chunk); sequences longer than one chunk are merged by a second small launch through `workspace`
 * (ql_paged_attn_workspace_bytes bytes, no initialisation needed; calls sharing it must be
 * stream-ordered). */
size_t ql_paged_attn_workspace_bytes(int64_t B, int Hq, int Hkv, int block_size, int max_blocks);
int ql_paged_attn_decode(const void* q_bf16, const void* k_cache, const void* v_cache,
                         const int* block_tables, const int* context_lens, void* out_bf16, float scale,
                         int64_t B, int Hq, int Hkv, int head_dim, int block_size, int max_blocks,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same for a W8A8 o_proj (nanovllm/models/qwen3.py:89-90: attn -> o_proj): the merge launch also takes over the
 * per_row_quant(x) that per_row_matmul runs on its input (nanovllm/utils/quantization.py:323).  q_out [B, Hq*128] in
 * `dtype` and scale_out f32 [B] are bit-identical to ql_quant_per_row(out); out is written as well. */
int ql_paged_attn_decode_quant(int dtype, const void* q_bf16, const void* k_cache, const void* v_cache,
                               const int* block_tables, const int* context_lens, void* out_bf16, void* q_out,
                               float* scale_out, float scale, int64_t B, int Hq, int Hkv, int head_dim, int block_size,
                               int max_blocks, void* workspace, size_t workspace_bytes, void* stream);
/* The same, also emitting rowmax_part_out f32 [B, Hkv]: the |max| of every row of out over each kv head's
 * (Hq/Hkv) x 128 outputs, i.e. what ql_gemm_w8a8_dynq takes as rowmax_part (nparts = Hkv) when a W8A8 o_proj
 * consumes out (nanovllm/models/qwen3.py:89-90: attn -> o_proj, whose per_row_quant(x) then needs no launch). */
int ql_paged_attn_decode_rowmax(const void* q_bf16, const void* k_cache, const void* v_cache,
                                const int* block_tables, const int* context_lens, void* out_bf16,
                                float* rowmax_part_out, float scale, int64_t B, int Hq, int Hkv, int head_dim,
                                int block_size, int max_blocks, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QLINEAR_H */
