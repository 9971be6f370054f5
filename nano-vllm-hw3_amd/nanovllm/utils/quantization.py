"""Quantized linear layers on MI355X: the host-side mirror of the reference's
``nanovllm/utils/quantization.py`` (same function / class names, argument meaning and assertions),
with every tensor operation executed by hand-written HIP kernels behind ``libqlinear.so``.

Differences from the reference, all deliberate:
  * the modules are tensor-parallel aware (the reference asserts ``tp_size == 1``,
    quantization.py:360, 378, 397): column-parallel shards keep their own rows' scales,
    row-parallel shards quantize their K-shard and all-reduce the partial outputs exactly as
    ``RowParallelLinear.forward`` does (linear.py:149-153);
  * ``QLinear`` can run weight-only (``weight_only=True``: bf16 activations, W8A16) besides the
    reference's dynamic-activation W8A8 form;
  * per-tensor / per-group forwards fuse the dequant into the MFMA GEMM instead of materialising an
    fp32 copy of the weight per call (quantization.py:258-259, 299-302);
  * the bias is added per output column (the reference's Triton kernel indexes it by tile row,
    quantization.py:133).
"""
import torch
import torch.distributed as dist

from nanovllm import _qlinear as Q

_DTYPES = [torch.float8_e4m3fn, torch.int8]
# o_proj / down_proj of a weight-only quantized model hand their split-K reduction to the norm launch behind them
# (NANOVLLM_NO_DEFER_SPLITK=1 keeps the self-contained GEMM, for A/B measurements)
import os as _os
# Opt-in (NANOVLLM_DYNQ=1): W8A8 o_proj / down_proj take their input's partial row maxima from the kernel that
# produced it and quantize inside the GEMM, one launch less each and bit-identical results - but measured SLOWER than
# the separate ql_quant_per_row launch: every one of the N/32 column-tile workgroups quantizes the activation tile
# again (int8 o_proj at 128 rows: 8.7 us against 7.3 us for quantize + GEMM; fp8 21.6 against 7.7 with the software
# e4m3 conversion; whole bench 30.2 k against 31.5 k tok/s for int8, 24.2 k against 30.3 k for fp8).
FUSE_ACT_QUANT = _os.environ.get("NANOVLLM_DYNQ", "") != ""
DEFER_SPLITK = _os.environ.get("NANOVLLM_NO_DEFER_SPLITK", "") == ""


_scratch16 = {}


def _ws16(device):
    """16-byte scratch of the per-tensor quantizers (their global |max| word).  Deliberately NOT the
    GEMM workspace: that one starts with the split-K ticket counters, which must stay zero."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _scratch16:
        _scratch16[key] = torch.zeros(16, dtype=torch.uint8, device=device)
    return _scratch16[key]


# activations that are not bf16: fp16 / fp32, as the reference's matmuls accept them (they compute in fp32 and return
# x's dtype: quantization.py:258-259, 302, 329); value = the library's QL_IN_* code
_WIDE = {torch.float16: 1, torch.float32: 2}


# --------------------------------------------------------------------------- quantizers
@torch.no_grad()
def per_row_quant(x: torch.Tensor, dtype: torch.dtype):
    """x [T, H] bf16 (or fp16 / fp32: the reference's kernel widens whatever it loads to f32, quantization.py:27)
    -> (q [T, H] dtype, scale [T] f32).  Reference: quantization.py:147-163."""
    assert dtype in _DTYPES
    assert x.dim() == 2
    T, H = x.shape
    qx = torch.empty_like(x, dtype=dtype)
    sx = torch.empty((T,), dtype=torch.float32, device=x.device)
    if x.dtype in _WIDE:
        Q.require_cuda(x, x.dtype, "x")
        Q.check(Q.load().ql_quant_per_row_wide(Q.dtype_code(dtype), _WIDE[x.dtype], x.data_ptr(), T, H, qx.data_ptr(),
                                               sx.data_ptr(), Q.stream_ptr(x.device)))
        return qx, sx
    Q.require_cuda(x, torch.bfloat16, "x")
    Q.check(Q.load().ql_quant_per_row(Q.dtype_code(dtype), x.data_ptr(), T, H, qx.data_ptr(), sx.data_ptr(),
                                      Q.stream_ptr(x.device)))
    return qx, sx


@torch.no_grad()
def per_tensor_quant(x: torch.Tensor, dtype: torch.dtype):
    """One scale for the whole tensor; the scale comes back as a 0-dim tensor of x's dtype, as in the
    reference (quantization.py:166-191)."""
    assert dtype in _DTYPES
    Q.require_cuda(x, torch.bfloat16, "x")
    x2 = x.reshape(-1, x.shape[-1])
    qx = torch.empty_like(x, dtype=dtype)
    scale = torch.empty((1,), dtype=torch.float32, device=x.device)
    Q.check(Q.load().ql_quant_per_tensor(Q.dtype_code(dtype), x2.data_ptr(), x2.shape[0], x2.shape[1], qx.data_ptr(),
                                         scale.data_ptr(), _ws16(x.device).data_ptr(), Q.stream_ptr(x.device)))
    return qx, scale[0].to(x.dtype)


@torch.no_grad()
def per_group_quant(x: torch.Tensor, group_size: int, dtype: torch.dtype):
    """Groups of ``group_size`` along K; scales [N, K/group] f32.  Reference: quantization.py:194-234."""
    assert dtype in _DTYPES
    assert x.dim() == 2
    assert x.size(1) % group_size == 0, f"K dim ({x.size(1)}) must be divisible by group_size ({group_size})"
    Q.require_cuda(x, torch.bfloat16, "x")
    N, K = x.shape
    qx = torch.empty_like(x, dtype=dtype)
    sx = torch.empty((N, K // group_size), dtype=torch.float32, device=x.device)
    Q.check(Q.load().ql_quant_per_group(Q.dtype_code(dtype), x.data_ptr(), N, K, group_size, qx.data_ptr(),
                                        sx.data_ptr(), Q.stream_ptr(x.device)))
    return qx, sx


# --------------------------------------------------------------------------- matmuls
def _flatten(x, n_out):
    y_shape = list(x.shape)[:-1] + [n_out]
    return x.reshape(-1, x.shape[-1]), y_shape


class TiledWeight:
    """A quantized weight re-arranged by ``tile_weight`` (the layout the kernels load fastest): ``data`` holds
    ceil(N/32)*32 x K bytes in tile order, ``shape`` is the logical (N, K)."""

    def __init__(self, data, n, k, dtype):
        self.data, self.shape, self.dtype = data, (n, k), dtype
        self.device = data.device

    def size(self, i):
        return self.shape[i]

    def data_ptr(self):
        return self.data.data_ptr()

    def is_contiguous(self):
        return True

    @property
    def is_cuda(self):
        return self.data.is_cuda

    def to_rowmajor(self):
        """Back to the plain [N, K] tensor (checkpoints store that)."""
        n, k = self.shape
        t = self.data.view(torch.uint8).view(-1, k // 32, 2, 32, 16).permute(0, 3, 1, 2, 4).reshape(-1, k)[:n]
        return t.contiguous().view(self.dtype)


@torch.no_grad()
def tile_weight(qw: torch.Tensor):
    """qw [N, K] int8 / fp8 (row-major, K % 32 == 0) -> TiledWeight: the 32 columns x 32 k one wave loads at a time
    stored as one contiguous KiB in lane order.  Accepted by every matmul here in place of qw."""
    assert qw.dim() == 2 and qw.dtype in _DTYPES and qw.shape[1] % 32 == 0
    Q.require_cuda(qw, None, "qw")
    N, K = qw.shape
    out = torch.empty(((N + 31) // 32 * 32, K), dtype=qw.dtype, device=qw.device)
    Q.check(Q.load().ql_tile_weight(qw.data_ptr(), out.data_ptr(), N, K, Q.stream_ptr(qw.device)))
    return TiledWeight(out, N, K, qw.dtype)


def _wcode(qw, dtype):
    return Q.dtype_code(dtype) | (Q.QL_TILED if isinstance(qw, TiledWeight) else 0)


def _bf16_terms(x2):
    """fp16 / fp32 activations as a sum of bf16 tensors, exactly: an fp16 value is the sum of two bf16 values (its
    leading 8 significant bits, and the rest), an fp32 value of three."""
    xf = x2.float()
    terms = []
    for _ in range(2 if x2.dtype == torch.float16 else 3):
        t = xf.to(torch.bfloat16)
        terms.append(t.contiguous())
        xf = xf - t.float()
    return terms


def _w8a16_wide(x, qw, sw32, bias, granularity, group_size, dtype, silu):
    """The weight-only matmuls on fp16 / fp32 activations (reference: ``x.float() @ w.T`` in fp32, result cast to x's
    dtype, quantization.py:258-261, 299-305).  The activations are split exactly into bf16 terms; every term runs the
    same MFMA kernels in their deferred form (unscaled f32 partial sums, ql_gemm_w8a16_partial), and the f32 slices of
    all terms are added up, scaled and cast once - so the result carries x's precision, not bf16's."""
    lib = Q.load()
    N, K = qw.shape
    x2 = x.reshape(-1, K).contiguous()
    M = x2.shape[0]
    acc = torch.zeros((M, N), dtype=torch.float32, device=x.device)
    if M > 0:
        S = lib.ql_gemm_partial_slices(M, N, K, granularity)
        parts = torch.empty((max(S, 1), M, N), dtype=torch.float32, device=x.device)
        for t in _bf16_terms(x2):
            Q.check(lib.ql_gemm_w8a16_partial(_wcode(qw, dtype), granularity, t.data_ptr(), qw.data_ptr(), sw32.data_ptr(), M, N,
                                              K, group_size or 0, parts.data_ptr(), S, Q.stream_ptr(x.device)))
            for s_ in range(max(S, 1)):
                acc += parts[s_]
    if granularity == Q.QL_PER_ROW:
        y = acc * sw32[None, :]
        y = (y + bias.float()[None, :]) if bias is not None else y              # bf16((sum * s) + b): one rounding
        y = y.to(x.dtype)
    else:
        if granularity == Q.QL_PER_TENSOR:
            acc = acc * sw32.reshape(1, 1)
        y = acc.to(x.dtype)
        if bias is not None:
            y = y + bias.to(x.dtype)[None, :]                                       # (the reference adds it after the cast)
    if silu:                                                                        # rows interleaved in 32-row gate / up strips
        g = y.view(M, N // 64, 2, 32)
        y = (torch.nn.functional.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N // 2)
    return y.view(*x.shape[:-1], y.shape[-1])


def _w8a16(x, qw, sw, bias, granularity, group_size, dtype, silu=False, out=None):
    assert dtype in _DTYPES and qw.dtype == dtype
    if x.dtype in _WIDE:
        assert out is None and x.shape[-1] == qw.shape[1]
        Q.require_cuda(x.reshape(-1, x.shape[-1]), x.dtype, "x")
        sw_ = sw if sw.dtype == torch.float32 else sw.to(torch.float32)
        sw_ = Q.require_cuda(sw_.reshape(-1) if sw_.dim() == 0 else sw_, torch.float32, "sw")
        return _w8a16_wide(x, qw, sw_, bias, granularity, group_size, dtype, silu)
    x2, y_shape = _flatten(x, qw.shape[0] // 2 if silu else qw.shape[0])
    Q.require_cuda(x2, torch.bfloat16, "x")
    Q.require_cuda(qw, dtype, "qw")
    sw32 = sw if sw.dtype == torch.float32 else sw.to(torch.float32)
    sw32 = Q.require_cuda(sw32.reshape(-1) if sw32.dim() == 0 else sw32, torch.float32, "sw")
    if bias is not None:
        Q.require_cuda(bias, torch.bfloat16, "bias")
    M, K = x2.shape
    N = qw.shape[0]
    assert qw.shape[1] == K, f"x has K={K} but qw has K={qw.shape[1]}"
    lib = Q.load()
    if silu:
        y = torch.empty((M, N // 2), dtype=torch.bfloat16, device=x.device)
        ws_ptr, ws_len = Q.workspace.get(x.device, lib.ql_gemm_workspace_bytes(M, N, K, granularity))
        Q.check(lib.ql_gemm_w8a16_silu(_wcode(qw, dtype), granularity, x2.data_ptr(), qw.data_ptr(), sw32.data_ptr(),
                                       bias.data_ptr() if bias is not None else None, y.data_ptr(), M, N, K,
                                       group_size or 0, ws_ptr, ws_len, Q.stream_ptr(x.device)))
        return y.view(y_shape)
    if out is not None:
        assert out.shape == (M, N)
        y = Q.require_cuda(out, torch.bfloat16, "out")
    else:
        y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    ws_ptr, ws_len = Q.workspace.get(x.device, lib.ql_gemm_workspace_bytes(M, N, K, granularity))
    Q.check(lib.ql_gemm_w8a16(_wcode(qw, dtype), granularity, x2.data_ptr(), qw.data_ptr(), sw32.data_ptr(),
                              bias.data_ptr() if bias is not None else None, y.data_ptr(), M, N, K,
                              group_size or 0, ws_ptr, ws_len, Q.stream_ptr(x.device)))
    return y if out is not None else y.view(y_shape)


@torch.no_grad()
def per_tensor_matmul(x, qw, sw, bias, dtype, silu=False, out=None):
    """``(x.float() @ (qw.float() * sw).T).to(bf16) + bias`` fused (quantization.py:237-262).  ``out`` (every
    matmul here): a contiguous [M, N] bf16 tensor to write instead of a new one (x is then [M, K])."""
    return _w8a16(x, qw, sw, bias, Q.QL_PER_TENSOR, 0, dtype, silu, out)


@torch.no_grad()
def per_group_matmul(x, qw, sw, bias, group_size, dtype, silu=False, out=None):
    """Per-group dequant fused into the GEMM (quantization.py:265-307)."""
    assert qw.shape[1] % group_size == 0
    return _w8a16(x, qw, sw, bias, Q.QL_PER_GROUP, group_size, dtype, silu, out)


@torch.no_grad()
def per_row_weight_only_matmul(x, qw, sw, bias, dtype, silu=False, out=None):
    """W8A16 with per-output-channel scales: ``bf16((x @ q.T) * sw + bias)``; the numerics the
    reference reaches through ``F.linear(x, fake_per_row_quant(W))`` (quantization.py:438-439).

    ``silu=True`` (every matmul here): qw / sw / bias rows are in the interleaved gate/up strip order
    of ``interleave_gate_up`` and the result is ``silu(gate) * up`` of shape [..., N/2], bit-identical
    to the plain matmul on the un-interleaved weight followed by SiluAndMul."""
    return _w8a16(x, qw, sw, bias, Q.QL_PER_ROW, 0, dtype, silu, out)


@torch.no_grad()
def w8a8_matmul(qx, sx, qw, sw, bias, dtype, y_shape=None, silu=False, out=None, return_rowmax=False):
    """The dual-scale GEMM on already-quantized activations (quantization.py:54-144, 331-345).  ``return_rowmax``
    (SiLU form only): also f32 [M, N/64], the |max| of every output row over each 32-column strip."""
    assert dtype in _DTYPES
    Q.require_cuda(qx, dtype, "qx")
    Q.require_cuda(qw, dtype, "qw")
    Q.require_cuda(sx, torch.float32, "sx")
    Q.require_cuda(sw, torch.float32, "sw")
    if bias is not None:
        Q.require_cuda(bias, torch.bfloat16, "bias")
    M, K = qx.shape
    N = qw.shape[0]
    assert qw.shape[1] == K
    lib = Q.load()
    assert silu or not return_rowmax
    if silu:
        y = torch.empty((M, N // 2), dtype=torch.bfloat16, device=qx.device)
        ws_ptr, ws_len = Q.workspace.get(qx.device, lib.ql_gemm_workspace_bytes(M, N, K, Q.QL_PER_ROW))
        if return_rowmax:
            rowmax = torch.empty((M, N // 64), dtype=torch.float32, device=qx.device)
            Q.check(lib.ql_gemm_w8a8_silu_rowmax(_wcode(qw, dtype), qx.data_ptr(), sx.data_ptr(), qw.data_ptr(),
                                                 sw.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                 y.data_ptr(), rowmax.data_ptr(), M, N, K, ws_ptr, ws_len,
                                                 Q.stream_ptr(qx.device)))
            return (y if y_shape is None else y.view(y_shape)), rowmax
        Q.check(lib.ql_gemm_w8a8_silu(_wcode(qw, dtype), qx.data_ptr(), sx.data_ptr(), qw.data_ptr(), sw.data_ptr(),
                                      bias.data_ptr() if bias is not None else None, y.data_ptr(), M, N, K,
                                      ws_ptr, ws_len, Q.stream_ptr(qx.device)))
        return y if y_shape is None else y.view(y_shape)
    if out is not None:
        assert out.shape == (M, N)
        y = Q.require_cuda(out, torch.bfloat16, "out")
    else:
        y = torch.empty((M, N), dtype=torch.bfloat16, device=qx.device)
    ws_ptr, ws_len = Q.workspace.get(qx.device, lib.ql_gemm_workspace_bytes(M, N, K, Q.QL_PER_ROW))
    Q.check(lib.ql_gemm_w8a8(_wcode(qw, dtype), qx.data_ptr(), sx.data_ptr(), qw.data_ptr(), sw.data_ptr(),
                             bias.data_ptr() if bias is not None else None, y.data_ptr(), M, N, K, ws_ptr, ws_len,
                             Q.stream_ptr(qx.device)))
    return y if y_shape is None else y.view(y_shape)


def row_absmax_parts(x, parts):
    """f32 [M, parts]: |max| of every row of bf16 x [M, K] over ``parts`` equal column ranges."""
    Q.require_cuda(x, torch.bfloat16, "x")
    M, K = x.shape
    out = torch.empty((M, parts), dtype=torch.float32, device=x.device)
    Q.check(Q.load().ql_row_absmax_parts(x.data_ptr(), M, K, parts, out.data_ptr(), Q.stream_ptr(x.device)))
    return out


@torch.no_grad()
def w8a8_dynq_matmul(x, rowmax_part, qw, sw, bias, dtype, out=None):
    """per_row_matmul (quantization.py:310-346) with its activation quantization (:323) inside the GEMM: ``x`` is
    bf16, ``rowmax_part`` f32 [M, parts] holds partial |max| values of every row of x over any split of its columns
    (emitted by the kernel that produced x).  Bit-identical to per_row_quant(x) + w8a8_matmul."""
    assert dtype in _DTYPES
    x2, y_shape = _flatten(x, qw.size(0))
    Q.require_cuda(x2, torch.bfloat16, "x")
    Q.require_cuda(rowmax_part, torch.float32, "rowmax_part")
    Q.require_cuda(qw, dtype, "qw")
    Q.require_cuda(sw, torch.float32, "sw")
    if bias is not None:
        Q.require_cuda(bias, torch.bfloat16, "bias")
    M, K = x2.shape
    N = qw.shape[0]
    assert qw.shape[1] == K and rowmax_part.dim() == 2 and rowmax_part.shape[0] == M
    if out is not None:
        assert out.shape == (M, N)
        y = Q.require_cuda(out, torch.bfloat16, "out")
    else:
        y = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    lib = Q.load()
    ws_ptr, ws_len = Q.workspace.get(x2.device, lib.ql_gemm_workspace_bytes(M, N, K, Q.QL_PER_ROW))
    Q.check(lib.ql_gemm_w8a8_dynq(_wcode(qw, dtype), x2.data_ptr(), rowmax_part.data_ptr(), rowmax_part.shape[1],
                                  qw.data_ptr(), sw.data_ptr(), bias.data_ptr() if bias is not None else None,
                                  y.data_ptr(), M, N, K, ws_ptr, ws_len, Q.stream_ptr(x2.device)))
    return y if out is not None else y.view(y_shape)


def _per_row_matmul_wide(x, qw, sw, bias, dtype, silu):
    """per_row_matmul on fp16 / fp32 activations: the quantizer widens them to f32 (as the reference's does), the
    contraction runs in the deferred form (raw int32 / f32 slices) and the reference's epilogue
    ``(acc * sx[m]) * sw[n] (+ bias)`` -> x's dtype (quantization.py:111-144) is applied to the summed slices."""
    lib = Q.load()
    N, K = qw.shape
    x2 = x.reshape(-1, K).contiguous()
    M = x2.shape[0]
    qx, sx = per_row_quant(x2, dtype)
    is_int = dtype == torch.int8
    S = lib.ql_gemm_partial_slices(M, N, K, Q.QL_PER_ROW) if M > 0 else 1
    parts = torch.zeros((max(S, 1), M, N), dtype=torch.int32 if is_int else torch.float32, device=x.device)
    if M > 0:
        Q.check(lib.ql_gemm_w8a8_partial(_wcode(qw, dtype), qx.data_ptr(), qw.data_ptr(), M, N, K, parts.data_ptr(), S,
                                         Q.stream_ptr(x.device)))
    acc = parts[0].clone()
    for s_ in range(1, max(S, 1)):
        acc += parts[s_]
    y = acc.float() * sx[:, None] * sw[None, :]
    if bias is not None:
        y = y + bias.float()[None, :]
    y = y.to(x.dtype)
    if silu:
        g = y.view(M, N // 64, 2, 32)
        y = (torch.nn.functional.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N // 2)
    return y.view(*x.shape[:-1], y.shape[-1])


@torch.no_grad()
def per_row_matmul(x, qw, sw, bias, dtype, silu=False, out=None):
    """Dynamic per-token activation quant + W8A8 GEMM (quantization.py:310-346)."""
    if x.dtype in _WIDE:
        return _per_row_matmul_wide(x, qw, sw, bias, dtype, silu)
    x2, y_shape = _flatten(x, qw.size(0) // 2 if silu else qw.size(0))
    assert x2.size(1) % 32 == 0, f"K({x2.size(1)}) must be multiple of 32"
    qx, sx = per_row_quant(x2.contiguous(), dtype)
    return w8a8_matmul(qx, sx, qw, sw, bias, dtype, None if out is not None else y_shape, silu, out)


class PartialSum:
    """Output of a weight-only GEMM whose split-K reduction and epilogue are left to the residual add + RMSNorm
    launch that consumes it (ops.add_rmsnorm accepts it in place of x): ``parts`` [S, M, N] f32 unscaled partial
    sums, plus the scale / bias the epilogue applies."""

    def __init__(self, parts, granularity, sw, bias, shape, sx=None, is_int=False):
        self.parts, self.granularity, self.sw, self.bias, self.shape = parts, granularity, sw, bias, shape
        self.sx, self.is_int = sx, is_int      # W8A8: per-token activation scales; int32 slices (int8 x int8)

    def dim(self):
        return len(self.shape)


@torch.no_grad()
def w8a8_partial_matmul(mod, qx, sx):
    """``mod`` (a W8A8 QLinear) applied to per-row quantized activations with the split-K reduction and the dual-
    scale epilogue deferred: a PartialSum of raw slices (int32 for int8, f32 for fp8) plus both scale vectors."""
    lib = Q.load()
    dtype = mod.dtype
    N, K = mod.qw.shape
    Q.require_cuda(qx, dtype, "qx")
    Q.require_cuda(sx, torch.float32, "sx")
    M = qx.shape[0]
    S = lib.ql_gemm_partial_slices(M, N, K, Q.QL_PER_ROW)
    is_int = dtype == torch.int8
    parts = torch.empty((max(S, 1), M, N), dtype=torch.int32 if is_int else torch.float32, device=qx.device)
    Q.check(lib.ql_gemm_w8a8_partial(_wcode(mod.qw, dtype), qx.data_ptr(), mod.qw.data_ptr(), M, N, K, parts.data_ptr(), S,
                                     Q.stream_ptr(qx.device)))
    return PartialSum(parts, Q.QL_PER_ROW, mod.sw, mod._bias_for_rank(), (M, N), sx=sx, is_int=is_int)


@torch.no_grad()
def w8a16_partial_matmul(mod, x):
    """``mod`` (a weight-only quantized module) applied to x with the split-K reduction deferred: a PartialSum."""
    lib = Q.load()
    dtype = mod.dtype
    N, K = mod.qw.shape
    x2 = Q.require_cuda(x.reshape(-1, x.shape[-1]), torch.bfloat16, "x")
    M = x2.shape[0]
    if isinstance(mod, QGroupLinear):
        gran, group, sw = Q.QL_PER_GROUP, mod.group_size, mod.sw
    elif isinstance(mod, QTensorLinear):
        gran, group, sw = Q.QL_PER_TENSOR, 0, mod._sw32
    else:
        gran, group, sw = Q.QL_PER_ROW, 0, mod.sw
    S = lib.ql_gemm_partial_slices(M, N, K, gran)
    parts = torch.empty((max(S, 1), M, N), dtype=torch.float32, device=x.device)
    Q.check(lib.ql_gemm_w8a16_partial(_wcode(mod.qw, dtype), gran, x2.data_ptr(), mod.qw.data_ptr(), sw.data_ptr(), M, N, K,
                                      group, parts.data_ptr(), S, Q.stream_ptr(x.device)))
    return PartialSum(parts, gran, sw, mod._bias_for_rank(), tuple(x.shape[:-1]) + (N,))


@torch.no_grad()
def qkv_rope_matmul(mod, x, positions, cos_sin, q_norm_w, k_norm_w, eps, num_heads, num_kv_heads, head_dim,
                    k_cache=None, v_cache=None, slot_mapping=None, want_kv=True, quantized=None):
    """The qkv projection of a quantized module with q/k norm + RoPE + KV-cache store in the GEMM epilogue
    (qwen3.py:80-88 + attention.py:52-58 of the reference in one launch): returns (q [T,Hq,D], k, v) like
    ops.qknorm_rope_store and never materialises the [T, N] projection.  ``quantized`` = (qx, sx) for a W8A8
    module whose input was quantized by the preceding norm.  ``positions=None``: ``cos_sin`` is [T, D], the table
    rows of the tokens gathered once by the caller."""
    lib = Q.load()
    dtype = mod.dtype
    N, K = mod.qw.shape
    dev = mod.qw.device
    w8a8 = isinstance(mod, QLinear) and not mod.weight_only
    if w8a8:
        if quantized is None:
            x2 = x.reshape(-1, x.shape[-1])
            qx, sx = per_row_quant(x2.contiguous(), dtype)
        else:
            qx, sx = quantized
        Q.require_cuda(qx, dtype, "qx")
        T = qx.shape[0]
    else:
        x2 = Q.require_cuda(x.reshape(-1, x.shape[-1]), torch.bfloat16, "x")
        T = x2.shape[0]
    q = torch.empty((T, num_heads, head_dim), dtype=torch.bfloat16, device=dev)
    k = torch.empty((T, num_kv_heads, head_dim), dtype=torch.bfloat16, device=dev) if want_kv else None
    v = torch.empty((T, num_kv_heads, head_dim), dtype=torch.bfloat16, device=dev) if want_kv else None
    use_cache = slot_mapping is not None and k_cache is not None and k_cache.numel() > 0
    bias = mod._bias_for_rank()
    if positions is None:
        assert cos_sin.shape[0] == T, "without positions, cos_sin holds one gathered row per token"
    tail = (positions.data_ptr() if positions is not None else None, cos_sin.data_ptr(), q_norm_w.data_ptr() if q_norm_w is not None else None,
            k_norm_w.data_ptr() if k_norm_w is not None else None, eps, q.data_ptr(),
            k.data_ptr() if want_kv else None, v.data_ptr() if want_kv else None,
            k_cache.data_ptr() if use_cache else None, v_cache.data_ptr() if use_cache else None,
            slot_mapping.data_ptr() if use_cache else None, num_heads, num_kv_heads, head_dim)
    if w8a8:
        ws_ptr, ws_len = Q.workspace.get(dev, lib.ql_gemm_workspace_bytes(T, N, K, Q.QL_PER_ROW))
        Q.check(lib.ql_gemm_w8a8_rope(_wcode(mod.qw, dtype), qx.data_ptr(), sx.data_ptr(), mod.qw.data_ptr(),
                                      mod.sw.data_ptr(), bias.data_ptr() if bias is not None else None, T, N, K, *tail,
                                      ws_ptr, ws_len, Q.stream_ptr(dev)))
    else:
        if isinstance(mod, QGroupLinear):
            gran, group, sw = Q.QL_PER_GROUP, mod.group_size, mod.sw
        elif isinstance(mod, QTensorLinear):
            gran, group, sw = Q.QL_PER_TENSOR, 0, mod._sw32
        else:
            gran, group, sw = Q.QL_PER_ROW, 0, mod.sw
        ws_ptr, ws_len = Q.workspace.get(dev, lib.ql_gemm_workspace_bytes(T, N, K, gran))
        Q.check(lib.ql_gemm_w8a16_rope(_wcode(mod.qw, dtype), gran, x2.data_ptr(), mod.qw.data_ptr(), sw.data_ptr(),
                                       bias.data_ptr() if bias is not None else None, T, N, K, group, *tail,
                                       ws_ptr, ws_len, Q.stream_ptr(dev)))
    return q, k, v


def interleave_gate_up(t: torch.Tensor):
    """Rows [gate (I) | up (I)] -> 32-row strips alternating gate, up: the weight arrangement of the
    fused SiLU epilogue (row 64*s + j = gate row 32*s + j for j < 32, up row 32*s + j - 32 otherwise)."""
    I = t.shape[0] // 2
    assert t.shape[0] == 2 * I and I % 32 == 0, f"gate/up halves of {I} rows must be multiples of 32"
    rest = tuple(t.shape[1:])
    g = t[:I].reshape(I // 32, 1, 32, *rest)
    u = t[I:].reshape(I // 32, 1, 32, *rest)
    return torch.cat([g, u], dim=1).reshape(t.shape).contiguous()


def reserve_workspace(max_tokens: int, shapes, device=None):
    """Size the split-K scratch for every [N, K] in ``shapes`` and EVERY row count up to ``max_tokens`` (pass the
    largest batch any call can bring, prefill included; call before hipGraph capture: nothing may allocate
    inside a capture and the captured launches keep the buffer's address)."""
    lib = Q.load()
    device = device if device is not None else torch.cuda.current_device()
    need = 0
    for (N, K) in shapes:
        for gran in (Q.QL_PER_ROW, Q.QL_PER_GROUP):
            need = max(need, lib.ql_gemm_workspace_max_bytes(max(int(max_tokens), 1), N, K, gran))
    Q.workspace.reserve(device, need)
    return need


# --------------------------------------------------------------------------- modules
def _tp_info(other):
    return getattr(other, "tp_dim", None), getattr(other, "tp_rank", 0), getattr(other, "tp_size", 1)


class _QBase(torch.nn.Module):
    """Shared TP behaviour: a module converted from a row-parallel linear (tp_dim == 1) adds the bias
    on rank 0 only and sums the partial outputs over the TP group (linear.py:149-153)."""

    def _init_tp(self, tp_dim=None, tp_rank=0, tp_size=1):
        self.tp_dim, self.tp_rank, self.tp_size = tp_dim, tp_rank, tp_size

    def _bias_for_rank(self):
        if self.tp_dim == 1 and self.tp_rank != 0:
            return None
        return self.bias

    silu_fused = False

    def enable_silu_mul(self):
        """Turn a merged gate/up projection into ``silu(gate) * up``: rearrange the rows once, then every
        forward runs the GEMM with the fused epilogue and returns [..., N/2]."""
        assert self.tp_dim != 1, "a row-parallel linear cannot carry the gate/up epilogue"
        assert not self.silu_fused and not isinstance(self.qw, TiledWeight), "interleave before tiling"
        self.qw = interleave_gate_up(self.qw)
        if self.sw.dim() >= 1 and self.sw.shape[0] == self.qw.shape[0]:      # per-row [N] / per-group [N, K/g]
            self.sw = interleave_gate_up(self.sw)
        if self.bias is not None:
            b = interleave_gate_up(self.bias.data)
            # (the bias of a converted linear is a registered Parameter: a plain tensor cannot replace it)
            self.bias = torch.nn.Parameter(b, requires_grad=False) if isinstance(self.bias, torch.nn.Parameter) else b
        self.silu_fused = True
        return self

    def tile(self):
        """Re-arrange qw into the layout the kernels load fastest (tile_weight); the last step of a conversion,
        after any gate/up interleave."""
        if not isinstance(self.qw, TiledWeight) and self.qw.shape[1] % 32 == 0:
            self.qw = tile_weight(self.qw)
        return self

    def rowmajor_qw(self):
        return self.qw.to_rowmajor() if isinstance(self.qw, TiledWeight) else self.qw

    # largest row count for which a projection that feeds a residual add + RMSNorm leaves its split-K reduction to
    # that launch (decode sizes; a prefill-sized f32 partial would double the bytes of the output)
    PARTIAL_MAX_ROWS = 512

    def forward_partial(self, x):
        """forward() for a projection whose output goes straight into ops.add_rmsnorm: where it pays (weight-only
        arithmetic, decode-sized input, no tensor-parallel all-reduce in between) the result is a PartialSum that
        the norm launch finishes; otherwise the plain output."""
        weight_only = not isinstance(self, QLinear) or self.weight_only
        m = x.numel() // x.shape[-1]
        tpp = self._tp_partial(x)
        if tpp is not None:
            return tpp
        if not weight_only and x.dim() == 2 and x.is_cuda:
            # W8A8: quantize the activations (their own launch), then the deferred form where it pays
            return self.forward_partial_quantized(*per_row_quant(x.contiguous(), self.dtype))
        if (DEFER_SPLITK and weight_only and not self.silu_fused and not (self.tp_dim == 1 and self.tp_size > 1)
                and 0 < m <= self.PARTIAL_MAX_ROWS and x.dim() == 2 and self.fusable):
            gran = Q.QL_PER_GROUP if isinstance(self, QGroupLinear) else Q.QL_PER_TENSOR if isinstance(self, QTensorLinear) else Q.QL_PER_ROW
            if Q.load().ql_gemm_partial_slices(m, self.qw.shape[0], self.qw.shape[1], gran) > 1:   # else: nothing to defer
                return w8a16_partial_matmul(self, x)
        return self(x)

    _defer_reduce = False

    @property
    def fusable(self):
        """The fused forms (SiLU / qkv epilogues, deferred K slices) exist for this module's arithmetic: everything but
        per-group scales whose group size is not a multiple of the MFMA kernels' 32-k step (plain GEMM only)."""
        return getattr(self, "group_size", 32) % 32 == 0

    def _tp_partial(self, x, run=None):
        """A row-parallel shard under TP with a decode-sized input and the one-shot all-reduce switched on
        (layers/comm.py): the shard's own GEMM output, its all-reduce left to the norm launch behind it."""
        if not (self.tp_dim == 1 and self.tp_size > 1 and x.is_cuda and x.dim() == 2):
            return None
        from nanovllm.layers import comm
        c = comm.communicator(self.qw.shape[0], x.device)
        if c is None or not (0 < x.shape[0] <= c.max_rows):
            return None
        self._defer_reduce = True
        try:
            y = self(x) if run is None else run()
        finally:
            self._defer_reduce = False
        return comm.TPPartial(y, c)

    def _finish(self, y):
        if self._defer_reduce:
            return y
        if self.tp_dim == 1 and self.tp_size > 1:
            from nanovllm.layers.linear import tp_all_reduce
            y = tp_all_reduce(y)
        return y

    def _run(self, x, matmul):
        """``matmul(rows, out)`` runs the module's GEMM on ``x[rows]`` (rows=None: all of x, whatever its leading
        dims; out=None: allocate).  A row-parallel shard under TP with a prefill-sized input produces its partial
        output in row slabs, each all-reduced on the side stream while the next slab's GEMM runs
        (linear.py::_SideStreamAllReduce.reduce_slabs); everything else is one GEMM followed, for row-parallel
        shards, by one all-reduce."""
        if self.tp_dim == 1 and self.tp_size > 1 and not self.silu_fused:
            from nanovllm.layers.linear import tp_all_reduce_slabs, tp_slab_bounds
            m = x.numel() // x.shape[-1]
            bounds = tp_slab_bounds(m)
            if bounds is not None:
                y = torch.empty((m, self.qw.shape[0]), dtype=torch.bfloat16, device=x.device)
                tp_all_reduce_slabs(y, bounds, lambda a, b: matmul(slice(a, b), y[a:b]))
                return y.view(*x.shape[:-1], -1)
        return self._finish(matmul(None, None))


class QLinear(_QBase):
    """Per-row quantized Linear layer (reference: quantization.py:349-364).  ``weight_only=False`` is
    the reference's W8A8 form; ``weight_only=True`` keeps activations in bf16 (W8A16)."""

    def __init__(self, dtype, weight: torch.Tensor, bias, weight_only: bool = False):
        super().__init__()
        assert dtype in _DTYPES
        self.dtype = dtype
        self.weight_only = weight_only
        self.qw, self.sw = per_row_quant(weight.contiguous(), dtype)
        self.bias = bias
        self._init_tp()

    @classmethod
    def from_linear(cls, other, dtype, weight_only: bool = False):
        m = cls(dtype, other.weight.data, other.bias, weight_only)
        m._init_tp(*_tp_info(other))
        return m

    @classmethod
    def from_quantized(cls, other, dtype, qw, sw, bias=None, weight_only: bool = False):
        """Build from stored tensors (a pre-quantized checkpoint) instead of quantizing ``other.weight``;
        ``other`` only supplies the tensor-parallel role of the linear being replaced."""
        assert dtype in _DTYPES and qw.dtype == dtype and sw.dtype == torch.float32 and sw.shape == (qw.shape[0],)
        m = cls.__new__(cls)
        torch.nn.Module.__init__(m)
        m.dtype, m.weight_only, m.qw, m.sw, m.bias = dtype, weight_only, qw, sw, bias
        m._init_tp(*_tp_info(other))
        return m

    def forward(self, x: torch.Tensor):
        fn = per_row_weight_only_matmul if self.weight_only else per_row_matmul
        x2 = x.reshape(-1, x.shape[-1])
        return self._run(x, lambda rows, out: fn(x if rows is None else x2[rows], self.qw, self.sw,
                                                 self._bias_for_rank(), self.dtype, self.silu_fused, out))

    @property
    def takes_quantized_input(self):
        """True for the W8A8 form: the caller may quantize the activations itself (e.g. inside the preceding
        RMSNorm launch, ops.add_rmsnorm_quant) and call forward_quantized."""
        return not self.weight_only

    def forward_quantized(self, qx: torch.Tensor, sx: torch.Tensor, return_rowmax: bool = False):
        """forward() on activations already quantized per row: qx [M, K] in self.dtype, sx [M] f32.
        ``return_rowmax`` (a gate/up projection with the SiLU epilogue): also the partial row maxima of the output
        that the next W8A8 linear's forward_with_rowmax consumes."""
        assert not self.weight_only
        if return_rowmax:
            assert self.silu_fused
            return w8a8_matmul(qx, sx, self.qw, self.sw, self._bias_for_rank(), self.dtype, None, True, None, True)
        return self._run(qx, lambda rows, out: w8a8_matmul(
            qx if rows is None else qx[rows], sx if rows is None else sx[rows], self.qw, self.sw,
            self._bias_for_rank(), self.dtype, None, self.silu_fused, out))

    def forward_partial_quantized(self, qx: torch.Tensor, sx: torch.Tensor):
        """forward_quantized() for a projection whose output goes straight into ops.add_rmsnorm /
        ops.add_rmsnorm_quant: at decode sizes, without a tensor-parallel all-reduce in between and when the plan has
        K slices to defer, a PartialSum that the norm launch finishes; otherwise the plain output."""
        m = qx.shape[0]
        tpp = self._tp_partial(qx, run=lambda: self.forward_quantized(qx, sx))
        if tpp is not None:
            return tpp
        if (DEFER_SPLITK and not self.weight_only and not self.silu_fused and not (self.tp_dim == 1 and self.tp_size > 1)
                and 0 < m <= self.PARTIAL_MAX_ROWS
                and Q.load().ql_gemm_partial_slices(m, self.qw.shape[0], self.qw.shape[1], Q.QL_PER_ROW) > 1):
            return w8a8_partial_matmul(self, qx, sx)
        return self.forward_quantized(qx, sx)

    def forward_with_rowmax(self, x: torch.Tensor, rowmax_part: torch.Tensor):
        """forward() of the W8A8 form on bf16 activations whose partial row maxima the producing kernel emitted
        (f32 [M, parts], any split of the columns): the activation quantization of per_row_matmul
        (quantization.py:323) happens inside the GEMM, bit-identical to forward(x) and one launch less."""
        assert not self.weight_only and not self.silu_fused and x.dim() == 2
        return self._run(x, lambda rows, out: w8a8_dynq_matmul(
            x if rows is None else x[rows], rowmax_part if rows is None else rowmax_part[rows], self.qw, self.sw,
            self._bias_for_rank(), self.dtype, out))


class QTensorLinear(_QBase):
    """Per-tensor quantized Linear layer (reference: quantization.py:367-382)."""

    def __init__(self, dtype, weight: torch.Tensor, bias):
        super().__init__()
        assert dtype in _DTYPES
        self.dtype = dtype
        self.qw, self.sw = per_tensor_quant(weight.contiguous(), dtype)
        self._sw32 = self.sw.to(torch.float32).reshape(1)
        self.bias = bias
        self._init_tp()

    @classmethod
    def from_linear(cls, other, dtype):
        m = cls(dtype, other.weight.data, other.bias)
        m._init_tp(*_tp_info(other))
        return m

    @classmethod
    def from_quantized(cls, other, dtype, qw, sw, bias=None):
        assert dtype in _DTYPES and qw.dtype == dtype and sw.numel() == 1
        m = cls.__new__(cls)
        torch.nn.Module.__init__(m)
        m.dtype, m.qw, m.bias = dtype, qw, bias
        m.sw = sw.reshape(()).to(torch.bfloat16)
        m._sw32 = m.sw.to(torch.float32).reshape(1)
        m._init_tp(*_tp_info(other))
        return m

    def forward(self, x: torch.Tensor):
        x2 = x.reshape(-1, x.shape[-1])
        return self._run(x, lambda rows, out: per_tensor_matmul(x if rows is None else x2[rows], self.qw, self._sw32,
                                                                self._bias_for_rank(), self.dtype, self.silu_fused, out))


class QGroupLinear(_QBase):
    """Per-group quantized Linear layer (reference: quantization.py:385-401)."""

    def __init__(self, dtype, weight: torch.Tensor, bias, group_size: int):
        super().__init__()
        assert dtype in _DTYPES
        self.dtype = dtype
        self.group_size = group_size
        self.qw, self.sw = per_group_quant(weight.contiguous(), group_size, dtype)
        self.bias = bias
        self._init_tp()

    @classmethod
    def from_linear(cls, other, dtype, group_size: int):
        m = cls(dtype, other.weight.data, other.bias, group_size)
        m._init_tp(*_tp_info(other))
        return m

    @classmethod
    def from_quantized(cls, other, dtype, qw, sw, bias, group_size: int):
        assert dtype in _DTYPES and qw.dtype == dtype and sw.dtype == torch.float32
        assert qw.shape[1] % group_size == 0 and sw.shape == (qw.shape[0], qw.shape[1] // group_size)
        m = cls.__new__(cls)
        torch.nn.Module.__init__(m)
        m.dtype, m.group_size, m.qw, m.sw, m.bias = dtype, group_size, qw, sw, bias
        m._init_tp(*_tp_info(other))
        return m

    def forward(self, x: torch.Tensor):
        x2 = x.reshape(-1, x.shape[-1])
        return self._run(x, lambda rows, out: per_group_matmul(x if rows is None else x2[rows], self.qw, self.sw,
                                                               self._bias_for_rank(), self.group_size, self.dtype,
                                                               self.silu_fused, out))


# --------------------------------------------------------------------------- fake quant
@torch.no_grad()
def _fake_per_block_quant(x: torch.Tensor, blk_m: int, blk_n: int, dtype: torch.dtype):
    """Quantize-dequantize in bf16 (reference: quantization.py:404-431).  Supported block shapes are
    the ones the reference's wrappers use: (-1,-1) per tensor, (1,-1) per row, (1,g) per group."""
    assert dtype in _DTYPES
    Q.require_cuda(x, torch.bfloat16, "x")
    x2 = x.reshape(-1, x.size(-1))
    m, n = x2.shape
    if blk_m == -1 and blk_n == -1:
        gran, g = Q.QL_PER_TENSOR, 0
    elif blk_m == 1 and (blk_n == -1 or blk_n == n):
        gran, g = Q.QL_PER_ROW, 0
    elif blk_m == 1:
        assert n % blk_n == 0, f"n({n}) must be multiple of blk_n({blk_n})"
        gran, g = Q.QL_PER_GROUP, blk_n
    else:
        raise AssertionError(f"block shape ({blk_m},{blk_n}) is not one of per-tensor / per-row / per-group")
    out = torch.empty_like(x2)
    Q.check(Q.load().ql_fake_quant(Q.dtype_code(dtype), gran, x2.data_ptr(), m, n, g, out.data_ptr(),
                                   _ws16(x.device).data_ptr(), Q.stream_ptr(x.device)))
    return out.reshape(x.shape)


def fake_per_tensor_quant(x, dtype):
    return _fake_per_block_quant(x, -1, -1, dtype)


def fake_per_row_quant(x, dtype):
    return _fake_per_block_quant(x, 1, -1, dtype)


def fake_per_group_quant(x, group_size, dtype):
    return _fake_per_block_quant(x, 1, group_size, dtype)


# --------------------------------------------------------------------------- model rewriters
def _layers(model):
    return model.model.layers


def _empty_cache():
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def apply_weight_fake_quant(model, weight_fake_quant_fn=lambda x: x):
    """Overwrite the four linear weights of every decoder layer with fn(weight) (quantization.py:448-460)."""
    for l in _layers(model):
        for mod in (l.self_attn.qkv_proj, l.self_attn.o_proj, l.mlp.gate_up_proj, l.mlp.down_proj):
            mod.weight.data = weight_fake_quant_fn(mod.weight.data)


apply_weight_quant = apply_weight_fake_quant   # quantization.py:499-513 is the same walk


def _replace(model, make, fuse_silu=True):
    for l in _layers(model):
        l.self_attn.qkv_proj = make(l.self_attn.qkv_proj)
        l.self_attn.o_proj = make(l.self_attn.o_proj)
        l.mlp.gate_up_proj = make(l.mlp.gate_up_proj)
        if fuse_silu and (l.mlp.gate_up_proj.qw.shape[0] // 2) % 32 == 0 and l.mlp.gate_up_proj.fusable:
            l.mlp.gate_up_proj.enable_silu_mul()
        l.mlp.down_proj = make(l.mlp.down_proj)
        for m in (l.self_attn.qkv_proj, l.self_attn.o_proj, l.mlp.gate_up_proj, l.mlp.down_proj):
            m.tile()
    _empty_cache()


def apply_per_row_quant(model, dtype, weight_only: bool = False, fuse_silu: bool = True):
    """Replace the four linears per layer by QLinear (quantization.py:463-472).  ``fuse_silu`` (all
    three rewriters) folds the MLP's SiLU * mul into the gate/up GEMM epilogue; results are bit-identical."""
    _replace(model, lambda lin: QLinear.from_linear(lin, dtype, weight_only), fuse_silu)


def apply_tensor_quant(model, dtype, fuse_silu: bool = True):
    """quantization.py:475-484"""
    _replace(model, lambda lin: QTensorLinear.from_linear(lin, dtype), fuse_silu)


def apply_group_quant(model, dtype, group_size: int, fuse_silu: bool = True):
    """quantization.py:487-496"""
    _replace(model, lambda lin: QGroupLinear.from_linear(lin, dtype, group_size), fuse_silu)


def apply_quant_torchao(model, linear_dtype=torch.float8_e4m3fn):
    """The reference hands the model to torchao here (quantization.py:516-524).  torchao is a
    third-party package outside this build; selecting it is an error rather than a silent no-op."""
    raise NotImplementedError("quant_type='smoothquant' delegates to torchao, which is not part of this build")


def apply_quant_from_config(model, config):
    """The dispatch of ModelRunner.__init__ (model_runner.py:36-57): weight_quant_fn wins, then
    quant_type, then a non-bf16 linear_dtype means per-row.  ``per_row_weight_only`` (alias
    ``w8a16``) is this build's addition: per-row scales with bf16 activations."""
    if getattr(config, "weight_quant_fn", None) is not None:
        apply_weight_fake_quant(model, config.weight_quant_fn)
    elif getattr(config, "quant_type", None) is not None:
        qt = config.quant_type
        if qt == "per_tensor":
            apply_tensor_quant(model, config.linear_dtype)
        elif qt == "per_row":
            apply_per_row_quant(model, config.linear_dtype)
        elif qt in ("per_row_weight_only", "w8a16"):
            apply_per_row_quant(model, config.linear_dtype, weight_only=True)
        elif qt == "per_group":
            apply_group_quant(model, config.linear_dtype, config.group_size)
        elif qt == "smoothquant":
            apply_quant_torchao(model, config.linear_dtype)
    elif config.linear_dtype != torch.bfloat16:
        apply_per_row_quant(model, config.linear_dtype)
    if getattr(config, "quantize_lm_head", False):
        apply_lm_head_quant(model, config.linear_dtype if config.linear_dtype in _DTYPES else torch.int8)


@torch.no_grad()
def apply_lm_head_quant(model, dtype=torch.int8):
    """Opt-in: the LM head (``ParallelLMHead``, nanovllm/layers/embed_head.py:45-66 in the reference, bf16
    there) as a per-row weight-only quantized linear over this rank's vocabulary shard.  The embedding keeps
    its bf16 weight even when the two are tied."""
    head = model.lm_head
    q = QLinear(dtype, head.weight.data, None, weight_only=True)
    head.quantized = q.tile()
    return model
