"""Thin wrappers over the non-GEMM entry points of libqlinear (norm, activation, rope+KV store,
decode attention).  GPU only; every call is one kernel launch on the current stream."""
import torch

from nanovllm import _qlinear as Q
from nanovllm import cpu_ops

BF16 = torch.bfloat16
# Operators of the UNQUANTIZED model also accept CPU tensors (plain torch, nanovllm/cpu_ops.py: engine plumbing on
# a host without a GPU); add_rmsnorm_quant and everything in utils/quantization.py stay GPU only.


def add_rmsnorm(x, residual, weight, eps):
    """Returns (normed, new_residual); new_residual is x when residual is None.  x may be a PartialSum (a GEMM
    output whose split-K reduction and epilogue this launch performs)."""
    if hasattr(x, "parts"):
        return _add_rmsnorm_partial(x, residual, weight, eps)
    if hasattr(x, "tp_partial"):       # a row-parallel shard's partial sum: all-reduce + this norm in one launch
        return x.finish(residual, weight, eps)
    if not x.is_cuda:
        return cpu_ops.add_rmsnorm(x, residual, weight, eps)
    Q.require_cuda(x, BF16, "x")
    H = x.shape[-1]
    T = x.numel() // H
    out = torch.empty_like(x)
    if residual is None:
        Q.check(Q.load().ql_add_rmsnorm(x.data_ptr(), None, weight.data_ptr(), eps, out.data_ptr(), None, T, H,
                                        Q.stream_ptr(x.device)))
        return out, x
    Q.require_cuda(residual, BF16, "residual")
    res_out = torch.empty_like(x)
    Q.check(Q.load().ql_add_rmsnorm(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), eps, out.data_ptr(),
                                    res_out.data_ptr(), T, H, Q.stream_ptr(x.device)))
    return out, res_out


def _add_rmsnorm_partial(ps, residual, weight, eps, quant_dtype=None):
    """The launch that finishes a deferred GEMM: (normed, new_residual), or with ``quant_dtype`` (a W8A8 linear
    follows) (q, scale, new_residual)."""
    assert residual is not None, "a deferred GEMM output always joins a residual"
    Q.require_cuda(residual, BF16, "residual")
    S, T, H = ps.parts.shape
    res_out = torch.empty(ps.shape, dtype=BF16, device=residual.device)
    if getattr(ps, "sx", None) is not None:                    # W8A8: raw int32 / f32 slices + both scale vectors
        out = q = scale = None
        if quant_dtype is None:
            out = torch.empty_like(res_out)
        else:
            q = torch.empty((T, H), dtype=quant_dtype, device=residual.device)
            scale = torch.empty((T,), dtype=torch.float32, device=residual.device)
        Q.check(Q.load().ql_add_rmsnorm_partial_w8a8(
            -1 if quant_dtype is None else Q.dtype_code(quant_dtype), ps.parts.data_ptr(), S, 1 if ps.is_int else 0,
            ps.sx.data_ptr(), ps.sw.data_ptr(), ps.bias.data_ptr() if ps.bias is not None else None, residual.data_ptr(),
            weight.data_ptr(), eps, out.data_ptr() if out is not None else None, res_out.data_ptr(),
            q.data_ptr() if q is not None else None, scale.data_ptr() if scale is not None else None, T, H,
            Q.stream_ptr(residual.device)))
        return (out, res_out) if quant_dtype is None else (q, scale, res_out)
    out = torch.empty_like(res_out)
    Q.check(Q.load().ql_add_rmsnorm_partial(ps.parts.data_ptr(), S, ps.granularity, ps.sw.data_ptr(),
                                            ps.bias.data_ptr() if ps.bias is not None else None, residual.data_ptr(),
                                            weight.data_ptr(), eps, out.data_ptr(), res_out.data_ptr(), T, H,
                                            Q.stream_ptr(residual.device)))
    if quant_dtype is None:
        return out, res_out
    # a weight-only linear in front of a W8A8 one (mixed per-module quantization): the bf16 norm output is
    # quantized by its own launch, which is what the unfused pair computes
    from nanovllm.utils.quantization import per_row_quant
    q, scale = per_row_quant(out.reshape(T, H), quant_dtype)
    return q, scale, res_out


def add_rmsnorm_quant(x, residual, weight, eps, dtype):
    """add_rmsnorm + per-row activation quantization in one launch: returns (q, scale, new_residual) with
    (q, scale) bit-identical to per_row_quant(add_rmsnorm(...)[0], dtype).  x may be a W8A8 PartialSum."""
    if hasattr(x, "parts"):
        return _add_rmsnorm_partial(x, residual, weight, eps, quant_dtype=dtype)
    if hasattr(x, "tp_partial"):
        from nanovllm.utils.quantization import per_row_quant
        out, res_out = x.finish(residual, weight, eps)
        q, scale = per_row_quant(out, dtype)
        return q, scale, res_out
    Q.require_cuda(x, BF16, "x")
    H = x.shape[-1]
    T = x.numel() // H
    q = torch.empty((T, H), dtype=dtype, device=x.device)
    scale = torch.empty((T,), dtype=torch.float32, device=x.device)
    res_out = None
    if residual is not None:
        Q.require_cuda(residual, BF16, "residual")
        res_out = torch.empty_like(x)
    Q.check(Q.load().ql_add_rmsnorm_quant(Q.dtype_code(dtype), x.data_ptr(), residual.data_ptr() if residual is not None else None,
                                          weight.data_ptr(), eps, q.data_ptr(), scale.data_ptr(),
                                          res_out.data_ptr() if res_out is not None else None, T, H,
                                          Q.stream_ptr(x.device)))
    return q, scale, (x if residual is None else res_out)


def silu_mul(x):
    if not x.is_cuda:
        return cpu_ops.silu_mul(x)
    Q.require_cuda(x, BF16, "x")
    I = x.shape[-1] // 2
    T = x.numel() // (2 * I)
    out = torch.empty(list(x.shape[:-1]) + [I], dtype=BF16, device=x.device)
    Q.check(Q.load().ql_silu_mul(x.data_ptr(), out.data_ptr(), T, I, Q.stream_ptr(x.device)))
    return out


def qknorm_rope_store(qkv, positions, cos_sin, q_norm_w, k_norm_w, eps, num_heads, num_kv_heads, head_dim,
                      k_cache=None, v_cache=None, slot_mapping=None, want_kv=True):
    """qkv [T, (Hq+2Hkv)*D] -> q [T,Hq,D] (and contiguous k, v [T,Hkv,D] when want_kv), writing k/v into
    the paged cache rows named by slot_mapping."""
    if not qkv.is_cuda:
        return cpu_ops.qknorm_rope_store(qkv, positions, cos_sin, q_norm_w, k_norm_w, eps, num_heads, num_kv_heads,
                                         head_dim, k_cache, v_cache, slot_mapping, want_kv)
    Q.require_cuda(qkv, BF16, "qkv")
    T = qkv.shape[0]
    q = torch.empty((T, num_heads, head_dim), dtype=BF16, device=qkv.device)
    k = torch.empty((T, num_kv_heads, head_dim), dtype=BF16, device=qkv.device) if want_kv else None
    v = torch.empty((T, num_kv_heads, head_dim), dtype=BF16, device=qkv.device) if want_kv else None
    use_cache = slot_mapping is not None and k_cache is not None and k_cache.numel() > 0
    Q.check(Q.load().ql_qknorm_rope_store(
        qkv.data_ptr(), positions.data_ptr() if positions is not None else None, cos_sin.data_ptr(),
        q_norm_w.data_ptr() if q_norm_w is not None else None,
        k_norm_w.data_ptr() if k_norm_w is not None else None, eps, q.data_ptr(),
        k.data_ptr() if want_kv else None, v.data_ptr() if want_kv else None,
        k_cache.data_ptr() if use_cache else None, v_cache.data_ptr() if use_cache else None,
        slot_mapping.data_ptr() if use_cache else None, T, num_heads, num_kv_heads, head_dim,
        Q.stream_ptr(qkv.device)))
    return q, k, v


def reserve_attn_workspace(max_batch, num_heads, num_kv_heads, block_size, max_blocks, device=None):
    """Allocate the decode-attention workspace for the largest batch before graph capture."""
    device = device if device is not None else torch.cuda.current_device()
    n = Q.load().ql_paged_attn_workspace_bytes(max_batch, num_heads, num_kv_heads, block_size, max_blocks)
    Q.attn_workspace.reserve(device, n)
    return n


def paged_attn_decode(q, k_cache, v_cache, block_tables, context_lens, scale, return_rowmax=False, quant_dtype=None):
    """q [B,Hq,D]; caches [num_blocks, block_size, Hkv, D] -> out [B,Hq,D].  ``return_rowmax``: also f32 [B, Hkv],
    the |max| of every output row over each kv head's columns (the partial row maxima a W8A8 o_proj that quantizes
    its input inside the GEMM asks for).  ``quant_dtype`` (torch.int8 / float8_e4m3fn): returns (out, qx, sx) with
    the rows of out quantized as per_row_quant would, done by the merge launch."""
    if not q.is_cuda:
        assert not return_rowmax and quant_dtype is None
        return cpu_ops.paged_attn_decode(q, k_cache, v_cache, block_tables, context_lens, scale)
    Q.require_cuda(q, BF16, "q")
    B, Hq, D = q.shape
    _, block_size, Hkv, _ = k_cache.shape
    out = torch.empty_like(q)
    lib = Q.load()
    max_blocks = block_tables.shape[1]
    ws_ptr, ws_len = Q.attn_workspace.get(q.device, lib.ql_paged_attn_workspace_bytes(B, Hq, Hkv, block_size, max_blocks))
    if quant_dtype is not None:
        # a W8A8 o_proj consumes the output: the merge launch also quantizes every row (per_row_quant semantics)
        qx = torch.empty((B, Hq * D), dtype=quant_dtype, device=q.device)
        sx = torch.empty((B,), dtype=torch.float32, device=q.device)
        Q.check(lib.ql_paged_attn_decode_quant(Q.QL_INT8 if quant_dtype == torch.int8 else Q.QL_FP8, q.data_ptr(),
                                               k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(),
                                               context_lens.data_ptr(), out.data_ptr(), qx.data_ptr(), sx.data_ptr(), scale,
                                               B, Hq, Hkv, D, block_size, max_blocks, ws_ptr, ws_len,
                                               Q.stream_ptr(q.device)))
        return out, qx, sx
    if return_rowmax:
        rowmax = torch.empty((B, Hkv), dtype=torch.float32, device=q.device)
        Q.check(lib.ql_paged_attn_decode_rowmax(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                                block_tables.data_ptr(), context_lens.data_ptr(), out.data_ptr(),
                                                rowmax.data_ptr(), scale, B, Hq, Hkv, D, block_size, max_blocks, ws_ptr,
                                                ws_len, Q.stream_ptr(q.device)))
        return out, rowmax
    Q.check(lib.ql_paged_attn_decode(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(),
                                     context_lens.data_ptr(), out.data_ptr(), scale, B, Hq, Hkv, D, block_size,
                                     max_blocks, ws_ptr, ws_len, Q.stream_ptr(q.device)))
    return out


def attn_prefill(q, k, v, cu_seqlens, max_seqlen, scale):
    """Causal self-attention over packed sequences: q [T,Hq,D], k/v [T,Hkv,D], cu_seqlens int32 [S+1] on
    the device (sequence s owns rows cu[s]:cu[s+1]) -> out [T,Hq,D].  The varlen call of the reference's
    prefill branch (nanovllm/layers/attention.py:64-69) without a prefix cache."""
    if not q.is_cuda:
        return cpu_ops.attn_prefill(q, k, v, cu_seqlens, max_seqlen, scale)
    Q.require_cuda(q, BF16, "q")
    Q.require_cuda(k, BF16, "k")
    Q.require_cuda(v, BF16, "v")
    Q.require_cuda(cu_seqlens, torch.int32, "cu_seqlens")
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    assert k.shape == v.shape == (T, Hkv, D)
    out = torch.empty_like(q)
    Q.check(Q.load().ql_attn_prefill(q.data_ptr(), k.data_ptr(), v.data_ptr(), cu_seqlens.data_ptr(), out.data_ptr(),
                                     scale, cu_seqlens.numel() - 1, int(max_seqlen), Hq, Hkv, D, Q.stream_ptr(q.device)))
    return out


def lm_head_logits(x, weight):
    """bf16 logits [M, V] = x [M, K] @ weight [V, K]^T (f32 accumulation), the output projection of
    ParallelLMHead.forward (nanovllm/layers/embed_head.py:61) on the rows the caller selected."""
    Q.require_cuda(x, BF16, "x")
    Q.require_cuda(weight, BF16, "weight")
    M, K = x.shape
    V = weight.shape[0]
    assert weight.shape == (V, K)
    out = torch.empty((M, V), dtype=BF16, device=x.device)
    Q.check(Q.load().ql_lmhead_logits(x.data_ptr(), weight.data_ptr(), out.data_ptr(), M, V, K, Q.stream_ptr(x.device)))
    return out


def reserve_head_workspace(max_rows, vocab, device=None):
    """Allocate the sampler epilogue's workspace for the largest step before graph capture."""
    Q.head_workspace.reserve(device or "cuda", Q.load().ql_lmhead_workspace_bytes(max_rows, vocab))


def lm_head_argmax(x, weight, out=None, return_max=False):
    """Greedy tokens int64 [M] = argmax_rows(lm_head_logits(x, weight)) bit for bit, the sampler running in the
    projection's epilogue: the logits are never written (embed_head.py:61 + layers/sampler.py:10-12).
    ``return_max``: also the winning logits as f32 [M] (-inf for a row without a finite maximum)."""
    Q.require_cuda(x, BF16, "x")
    Q.require_cuda(weight, BF16, "weight")
    M, K = x.shape
    V = weight.shape[0]
    assert weight.shape == (V, K)
    if out is None:
        out = torch.empty((M,), dtype=torch.int64, device=x.device)
    assert out.dtype == torch.int64 and out.numel() == M and out.is_contiguous()
    lib = Q.load()
    ws, nbytes = Q.head_workspace.get(x.device, lib.ql_lmhead_workspace_bytes(M, V))
    mx = torch.empty((M,), dtype=torch.float32, device=x.device) if return_max else None
    Q.check(lib.ql_lmhead_argmax(x.data_ptr(), weight.data_ptr(), out.data_ptr(), mx.data_ptr() if return_max else None,
                                 M, V, K, ws, nbytes, Q.stream_ptr(x.device)))
    return (out, mx) if return_max else out


def argmax_rows(logits):
    """Greedy sampling: int64 [T] = argmax over the last dim of bf16 logits [T, V] (first index among ties)."""
    Q.require_cuda(logits, BF16, "logits")
    T, V = logits.shape
    out = torch.empty((T,), dtype=torch.int64, device=logits.device)
    Q.check(Q.load().ql_argmax_rows(logits.data_ptr(), out.data_ptr(), T, V, Q.stream_ptr(logits.device)))
    return out
