"""CPU oracle for the quantized-linear path (TEST INFRASTRUCTURE - never shipped, never measured
as the product).

A numpy restatement of the arithmetic of the reference's quantized linear layers.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product path (the HIP library behind ``include/qlinear.h``) never does.

Parity pinning
--------------
* per-tensor / per-group quantizers and matmuls, the fake quantizers and the TP weight loaders are
  pinned against outputs of the reference itself (``tests/golden/quant_golden.npz``, produced by
  ``tests/golden/make_golden.py`` importing the reference's torch code on CPU).
* the per-row W8A8 path is Triton-only in the reference.  Its restatement below follows the kernel
  text line by line and is pinned against the reference's own ``@triton.jit`` kernels executed by
  Triton's CPU interpreter (``tests/golden/triton_golden.npz``, generator
  ``tests/golden/make_golden_triton.py``): the f32 scales, the scaled values ``x * (Q / row_max)``
  and the GEMM's ``acc * sa * sb`` are reproduced BIT-EXACTLY up to the final narrowing cast.  The
  interpreter narrows f32 -> bf16 by truncation and f32 -> e4m3 by a carry-less round-half-up, which
  is not what compiled Triton does (round-to-nearest-even, as torch); the tests therefore apply the
  interpreter's narrowing to this oracle's pre-cast values (``return_scaled`` / ``return_f32``) and
  compare those, while the shipped semantics of the cast stay RTNE.  One call has no
  reference-produced vector: ``libdevice.round`` of the int8 activation/weight quantizer
  (quantization.py:41) cannot be interpreted; it is restated as C ``roundf`` and cross-checked
  against ``fake_per_row_quant`` golden outputs on the weight side.

Conventions: bf16 tensors travel as ``uint16`` bit patterns, fp8-e4m3fn and int8 as ``uint8`` /
``int8`` arrays, scales as ``float32``.  dtype names are the strings ``"int8"`` and ``"fp8"``.
"""
import numpy as np

INT8 = "int8"
FP8 = "fp8"
QMAX = {INT8: 127.0, FP8: 448.0}


# --------------------------------------------------------------------------- number formats
def bf16_to_f32(u16):
    """bf16 bit pattern -> float32 (exact)."""
    u16 = np.asarray(u16, dtype=np.uint16)
    return (u16.astype(np.uint32) << np.uint32(16)).view(np.float32)


def f32_to_bf16(f):
    """float32 -> bf16 bit pattern, round-to-nearest-even (what torch's ``.to(bfloat16)`` does)."""
    f = np.ascontiguousarray(f, dtype=np.float32)
    u = f.view(np.uint32)
    lsb = (u >> np.uint32(16)) & np.uint32(1)
    r = (u + np.uint32(0x7FFF) + lsb) >> np.uint32(16)
    r = np.where(np.isnan(f), np.uint32(0x7FC0), r)
    return r.astype(np.uint16)


def _fp8_table():
    t = np.zeros(256, dtype=np.float32)
    for b in range(256):
        s = -1.0 if b & 0x80 else 1.0
        e = (b >> 3) & 0xF
        m = b & 7
        if e == 0xF and m == 7:
            v = np.nan
        elif e == 0:
            v = s * (m / 8.0) * 2.0 ** (-6)
        else:
            v = s * (1.0 + m / 8.0) * 2.0 ** (e - 7)
        t[b] = v
    return t


_FP8_TABLE = _fp8_table()


def fp8_to_f32(u8):
    """OCP float8 e4m3fn bit pattern -> float32 (exact; 0x7F/0xFF are NaN, no infinities)."""
    return _FP8_TABLE[np.asarray(u8).view(np.uint8)]


def f32_to_fp8(f):
    """float32 -> e4m3fn bit pattern, round-to-nearest-even, no saturation: anything that rounds
    above 448 becomes the NaN pattern 0x7F.  This is the conversion torch applies for
    ``.to(torch.float8_e4m3fn)`` (used at quantization.py:189, 228, 426)."""
    f = np.ascontiguousarray(f, dtype=np.float32)
    bits = f.view(np.uint32).copy()
    sign = bits & np.uint32(0x80000000)
    a = bits ^ sign
    res = np.zeros(a.shape, dtype=np.uint32)
    is_nan = a >= np.uint32(1087 << 20)             # >= 480.0, inf, NaN
    small = a < np.uint32(121 << 23)                # < 2**-6: result is an fp8 subnormal
    denorm_mask = np.uint32(141 << 23)
    with np.errstate(all="ignore"):
        sm = (a.view(np.float32) + denorm_mask.view(np.float32)).astype(np.float32)
    sm = (sm.view(np.uint32) - denorm_mask) & np.uint32(0xFF)
    mant_odd = (a >> np.uint32(20)) & np.uint32(1)
    nm = a + np.uint32(((7 - 127) << 23) & 0xFFFFFFFF) + np.uint32(0x7FFFF) + mant_odd
    nm = (nm >> np.uint32(20)) & np.uint32(0xFF)
    res = np.where(small, sm, nm)
    res = np.where(is_nan, np.uint32(0x7F), res)
    res = res | (sign >> np.uint32(24))
    return res.astype(np.uint8)


def round_half_away(v):
    """C ``roundf`` (what ``libdevice.round`` is, quantization.py:41)."""
    v64 = np.asarray(v, dtype=np.float64)
    return np.trunc(v64 + np.copysign(0.5, v64)).astype(np.float32)


def _decode_q(q, dtype):
    """quantized storage -> float32 values."""
    if dtype == INT8:
        return np.asarray(q).view(np.int8).astype(np.float32)
    return fp8_to_f32(q)


# --------------------------------------------------------------------------- quantizers
def per_row_quant(x_bf16, dtype, return_scaled=False):
    """Dynamic per-row quantization of a bf16 matrix [T, H] (weights at init, activations at every
    forward).  Follows ``_per_row_quant_kernel`` (quantization.py:26-41) as launched by
    ``per_row_quant`` (quantization.py:147-163):
      row_max = max(max|x|, 1e-8) in f32; scale = row_max / Q; scale_inv = Q / row_max;
      y = x * scale_inv; int8: roundf(y) -> int8 ; fp8: cast(y) to e4m3fn.
    Returns (q [T,H] int8|uint8, scale [T] f32); with ``return_scaled`` the f32 values y in front of
    the narrowing cast instead of q."""
    # (bf16 bit patterns, or - as the kernel takes any float type and widens it to f32, quantization.py:27 - a float
    # array: fp16 / fp32 activations)
    x = np.asarray(x_bf16)
    x = x.astype(np.float32) if x.dtype.kind == "f" else bf16_to_f32(x)
    assert x.ndim == 2
    Q = np.float32(QMAX[dtype])
    row_max = np.maximum(np.abs(x).max(axis=1), np.float32(1e-8)).astype(np.float32)
    scale = (row_max / Q).astype(np.float32)
    scale_inv = (Q / row_max).astype(np.float32)
    y = (x * scale_inv[:, None]).astype(np.float32)
    if return_scaled:
        return y, scale
    if dtype == INT8:
        q = round_half_away(y).astype(np.int32).astype(np.int8)
    else:
        q = f32_to_fp8(y)
    return q, scale


def per_tensor_quant(w_bf16, dtype):
    """Per-tensor quantization (quantization.py:166-191).  The reference keeps everything in the
    input dtype: amax, ``scale = amax / Q`` and ``x / scale`` are bf16 tensors, i.e. each is the
    f32 result rounded to bf16, before round()/clamp()/cast.  Returns (q, scale) with scale a
    float32 scalar holding the bf16-valued scale."""
    x = bf16_to_f32(w_bf16)
    Q = np.float32(QMAX[dtype])
    amax = np.maximum(np.abs(x).max(), bf16_to_f32(f32_to_bf16(np.float32(1e-8))))
    scale = bf16_to_f32(f32_to_bf16(np.float32(amax) / Q))
    y = bf16_to_f32(f32_to_bf16((x / scale).astype(np.float32)))
    if dtype == INT8:
        q = np.clip(np.rint(y), -128, 127).astype(np.int8)
    else:
        q = f32_to_fp8(y)
    return q, np.float32(scale)


def per_group_quant(w_bf16, group_size, dtype):
    """Per-group quantization along K (quantization.py:194-234): f32 math, amax per group floored
    at 1e-8, ``scales = amax / Q``, ``x / scales`` then round-half-even + clamp (int8) or cast
    (fp8).  Returns (q [N,K], scales [N, K/g] f32)."""
    x = bf16_to_f32(w_bf16)
    N, K = x.shape
    assert K % group_size == 0
    Q = np.float32(QMAX[dtype])
    xg = x.reshape(N, K // group_size, group_size)
    amax = np.maximum(np.abs(xg).max(axis=2, keepdims=True), np.float32(1e-8)).astype(np.float32)
    scales = (amax / Q).astype(np.float32)
    y = (xg / scales).astype(np.float32)
    if dtype == INT8:
        q = np.clip(np.rint(y), -128, 127).astype(np.int8)
    else:
        q = f32_to_fp8(y)
    return q.reshape(N, K), scales[..., 0]


def fake_per_block_quant(w_bf16, blk_m, blk_n, dtype):
    """Quantize-dequantize in place, result in bf16 (quantization.py:404-431).  amax stays in the
    input dtype, so ``amax / 127`` is a bf16 value (rounded); int8: ``round(x / amax * 127) * (amax / 127)``;
    fp8: ``cast(x / amax * 448) / 448 * amax``; all in f32, then one rounding to bf16."""
    x = bf16_to_f32(w_bf16)
    m, n = x.shape
    bm = m if blk_m == -1 else blk_m
    bn = n if blk_n == -1 else blk_n
    assert m % bm == 0 and n % bn == 0
    xb = x.reshape(m // bm, bm, n // bn, bn)
    amax = np.abs(xb).max(axis=3, keepdims=True).max(axis=1, keepdims=True)
    # torch.clamp(bf16 tensor, min=1e-8): the bound is rounded to bf16
    amax = np.maximum(amax, bf16_to_f32(f32_to_bf16(np.float32(1e-8)))).astype(np.float32)
    if dtype == INT8:
        t = ((xb / amax).astype(np.float32) * np.float32(127)).astype(np.float32)
        step = bf16_to_f32(f32_to_bf16((amax / np.float32(127)).astype(np.float32)))  # bf16 / int stays bf16
        y = (np.rint(t).astype(np.float32) * step).astype(np.float32)
    else:
        t = ((xb / amax).astype(np.float32) * np.float32(448)).astype(np.float32)
        d = fp8_to_f32(f32_to_fp8(t))
        y = ((d / np.float32(448)).astype(np.float32) * amax).astype(np.float32)
    return f32_to_bf16(y.reshape(m, n))


def fake_per_tensor_quant(w, dtype):   # quantization.py:434-435
    return fake_per_block_quant(w, -1, -1, dtype)


def fake_per_row_quant(w, dtype):      # quantization.py:438-439
    return fake_per_block_quant(w, 1, -1, dtype)


def fake_per_group_quant(w, g, dtype):  # quantization.py:442-445
    return fake_per_block_quant(w, 1, g, dtype)


# --------------------------------------------------------------------------- matmuls
def _bf16_add_bias(y_bits, bias_bf16):
    """``y = y + bias`` on bf16 tensors: f32 add, one rounding (quantization.py:260-261, 304-305)."""
    if bias_bf16 is None:
        return y_bits
    return f32_to_bf16(bf16_to_f32(y_bits) + bf16_to_f32(bias_bf16)[None, :])


def per_tensor_matmul(x_bf16, qw, sw, bias_bf16, dtype, return_f64=False):
    """quantization.py:237-262: ``w = qw.float() * sw``; ``y = (x.float() @ w.T).to(bf16)``; then
    the bf16 bias add.  The contraction is evaluated in f64 here (the reference's f32 BLAS sums in
    an unspecified order), then rounded once to bf16."""
    x = bf16_to_f32(x_bf16).astype(np.float64)
    w = (_decode_q(qw, dtype) * np.float32(sw)).astype(np.float32)
    acc = x @ w.astype(np.float64).T
    if return_f64:
        return acc
    return _bf16_add_bias(f32_to_bf16(acc.astype(np.float32)), bias_bf16)


def per_group_matmul(x_bf16, qw, sw, bias_bf16, group_size, dtype, return_f64=False):
    """quantization.py:265-307: fp32 dequant ``qw.float() * sw[..., None]`` per group, fp32 GEMM,
    cast to bf16, then the bf16 bias add."""
    x = bf16_to_f32(x_bf16).astype(np.float64)
    N, K = np.asarray(qw).shape
    w = (_decode_q(qw, dtype).reshape(N, K // group_size, group_size)
         * np.asarray(sw, dtype=np.float32)[..., None]).astype(np.float32).reshape(N, K)
    acc = x @ w.astype(np.float64).T
    if return_f64:
        return acc
    return _bf16_add_bias(f32_to_bf16(acc.astype(np.float32)), bias_bf16)


def w8a8_gemm(qx, sx, qw, sw, bias_bf16, dtype, return_acc=False, return_f32=False):
    """``_per_row_post_process_mm`` (quantization.py:93-144): acc = sum_k qx[m,k]*qw[n,k] (exact
    int32 for int8; f32 with unspecified order for fp8 - evaluated exactly in f64 here), then in
    f32 ``c = acc * sa[m] * sb[n]`` in that order, ``+ bias[n]`` in f32, one cast to bf16.
    The reference adds ``bias[:, None]`` (quantization.py:133), which indexes the bias by the tile
    row; the intended per-output-column bias is implemented here and the quantized Qwen3 linears
    carry no bias (qwen3.py:131), so no reference-visible result depends on it."""
    if dtype == INT8:
        a = np.asarray(qx).view(np.int8).astype(np.int64)
        b = np.asarray(qw).view(np.int8).astype(np.int64)
        acc = a @ b.T
        accf = acc.astype(np.int32).astype(np.float32)
    else:
        a = fp8_to_f32(qx).astype(np.float64)
        b = fp8_to_f32(qw).astype(np.float64)
        acc = a @ b.T
        accf = acc.astype(np.float32)
    if return_acc:
        return acc
    c = (accf * np.asarray(sx, np.float32)[:, None]).astype(np.float32)
    c = (c * np.asarray(sw, np.float32)[None, :]).astype(np.float32)
    if bias_bf16 is not None:
        c = (c + bf16_to_f32(bias_bf16)[None, :]).astype(np.float32)
    if return_f32:                  # the f32 value in front of the cast to the output dtype (:144)
        return c
    return f32_to_bf16(c)


def per_row_matmul(x_bf16, qw, sw, bias_bf16, dtype):
    """quantization.py:310-346: quantize the activation rows dynamically, then the dual-scale GEMM."""
    qx, sx = per_row_quant(x_bf16, dtype)
    return w8a8_gemm(qx, sx, qw, sw, bias_bf16, dtype)


def w8a16_rowscale_matmul(x_bf16, qw, sw, bias_bf16, dtype, return_f64=False):
    """Weight-only per-row path: ``y[m,n] = bf16( (sum_k x[m,k]*q[n,k]) * sw[n] + bias[n] )`` with
    bf16 activations untouched.  The reference reaches these numerics through its fake-quant route
    (``F.linear(x, fake_per_row_quant(W))``, quantization.py:438-439 + linear.py:72-73), where the
    dequantised weight is additionally rounded to bf16 before the GEMM; this form keeps the integer
    weights exact and applies the f32 scale once, so it differs from that route by at most the
    bf16 rounding of the weights (relative 2**-9 per weight)."""
    x = bf16_to_f32(x_bf16).astype(np.float64)
    q = _decode_q(qw, dtype).astype(np.float64)
    acc = x @ q.T
    if return_f64:
        return acc * np.asarray(sw, np.float64)[None, :]
    c = (acc.astype(np.float32) * np.asarray(sw, np.float32)[None, :]).astype(np.float32)
    if bias_bf16 is not None:
        c = (c + bf16_to_f32(bias_bf16)[None, :]).astype(np.float32)
    return f32_to_bf16(c)


def abs_product_sum(x_bf16, w_f32):
    """L1[m,n] = sum_k |x[m,k] * w[n,k]| - the scale of f32 accumulation error, used by the tests to
    turn 'f32 accumulation order is unspecified' into an explicit per-element tolerance."""
    x = np.abs(bf16_to_f32(x_bf16)).astype(np.float64)
    return x @ np.abs(np.asarray(w_f32, dtype=np.float64)).T


# --------------------------------------------------------------------------- tensor parallel
def shard_rows(w, tp, rank):
    """Column-parallel shard: rows [rank*N/tp, (rank+1)*N/tp) (linear.py:65-70)."""
    n = w.shape[0] // tp
    return w[rank * n:(rank + 1) * n]


def shard_cols(w, tp, rank):
    """Row-parallel shard: input columns [rank*K/tp, (rank+1)*K/tp) (linear.py:142-147)."""
    k = w.shape[1] // tp
    return w[:, rank * k:(rank + 1) * k]


def pack_qkv(q, k, v, tp, rank):
    """QKVParallelLinear layout on one rank: this rank's q heads, then its k heads, then its v
    heads (linear.py:114-128)."""
    return np.concatenate([shard_rows(q, tp, rank), shard_rows(k, tp, rank), shard_rows(v, tp, rank)], 0)


def pack_gate_up(gate, up, tp, rank):
    """MergedColumnParallelLinear layout on one rank: gate shard then up shard (linear.py:87-93)."""
    return np.concatenate([shard_rows(gate, tp, rank), shard_rows(up, tp, rank)], 0)


def row_parallel_reduce(partials_bf16):
    """RowParallelLinear.forward with tp>1 (linear.py:149-153): every rank holds a bf16 partial
    [M, N]; the all-reduce sums them.  Summation is done in f32 in rank order and rounded once,
    which is what the HIP/RCCL path is compared with (RCCL's own bf16 ring rounds per hop, in an
    order that depends on the ring; both are within tp bf16 roundings of the exact sum)."""
    acc = np.zeros(bf16_to_f32(partials_bf16[0]).shape, dtype=np.float32)
    for p in partials_bf16:
        acc = (acc + bf16_to_f32(p)).astype(np.float32)
    return f32_to_bf16(acc)


# --------------------------------------------------------------------------- CPU baseline (timed by bench.py)
def cpu_reference_linear_torch(x_t, qw_t, sw_t, group_size=None):
    """The reference's CPU-runnable forward arithmetic on torch tensors, timed on host cores by
    ``bench.py`` as the ``cpu_baseline``: ``x.float() @ (qw.float() * sw).T`` cast back to x's dtype
    (quantization.py:258-259 per-tensor / per-row-broadcast, :293-302 per-group)."""
    import torch
    N, K = qw_t.shape
    if group_size:
        w = (qw_t.reshape(N, K // group_size, group_size).to(torch.float32) * sw_t.unsqueeze(-1)).reshape(N, K)
    else:
        w = qw_t.to(torch.float32) * sw_t.reshape(-1, 1) if sw_t.numel() > 1 else qw_t.to(torch.float32) * sw_t
    return (x_t.to(torch.float32) @ w.T).to(x_t.dtype)
