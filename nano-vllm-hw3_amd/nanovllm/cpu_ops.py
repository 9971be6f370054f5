"""Plain-torch counterparts of nanovllm/ops.py for tensors that live on the CPU.  PLUMBING ONLY: they let the
unquantized model run through the engine on a host without a GPU (Config(device="cpu")), following the formulas
of the reference's layers (layernorm.py:16-40, rotary_embedding.py:7-48, activation.py:11-14, attention.py:43-75,
sampler.py:10-12).  Nothing quantized goes through here and no HIP kernel falls back to this file: ops.py
dispatches on the device of its input and refuses CPU tensors for every quantized operator."""
import torch
import torch.nn.functional as F


def add_rmsnorm(x, residual, weight, eps):
    s = x.float() if residual is None else x.float() + residual.float()
    new_res = x if residual is None else s.to(x.dtype)
    out = (s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype) * weight
    return out, new_res


def silu_mul(x):
    a, b = x.chunk(2, -1)
    return F.silu(a) * b


def _rope(x, positions, cos_sin):
    cs = cos_sin[positions] if positions is not None else cos_sin      # [T, D] f32: cos | sin
    half = cs.shape[-1] // 2
    cos, sin = cs[:, None, :half], cs[:, None, half:]
    x1, x2 = x.float()[..., :half], x.float()[..., half:]
    return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), -1).to(x.dtype)


def _head_norm(x, w, eps):
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype) * w


def qknorm_rope_store(qkv, positions, cos_sin, q_norm_w, k_norm_w, eps, num_heads, num_kv_heads, head_dim,
                      k_cache=None, v_cache=None, slot_mapping=None, want_kv=True):
    T = qkv.shape[0]
    q, k, v = qkv.split([num_heads * head_dim, num_kv_heads * head_dim, num_kv_heads * head_dim], -1)
    q, k, v = q.view(T, num_heads, head_dim), k.view(T, num_kv_heads, head_dim), v.view(T, num_kv_heads, head_dim)
    if q_norm_w is not None:
        q, k = _head_norm(q, q_norm_w, eps), _head_norm(k, k_norm_w, eps)
    q, k = _rope(q, positions, cos_sin), _rope(k, positions, cos_sin)
    if slot_mapping is not None and k_cache is not None and k_cache.numel() > 0:
        slots = slot_mapping.long()
        keep = slots >= 0
        kc, vc = k_cache.view(-1, num_kv_heads, head_dim), v_cache.view(-1, num_kv_heads, head_dim)
        kc[slots[keep]] = k[keep]
        vc[slots[keep]] = v[keep]
    return q.contiguous(), (k.contiguous() if want_kv else None), (v.contiguous() if want_kv else None)


def _sdpa(q, k, v, scale, causal):
    # q [Tq, Hq, D], k/v [Tk, Hkv, D] -> [Tq, Hq, D]
    o = F.scaled_dot_product_attention(q.transpose(0, 1)[None].float(), k.transpose(0, 1)[None].float(),
                                       v.transpose(0, 1)[None].float(), is_causal=causal, scale=scale, enable_gqa=True)
    return o[0].transpose(0, 1).to(q.dtype)


def attn_prefill(q, k, v, cu_seqlens, max_seqlen, scale):
    cu = cu_seqlens.tolist()
    out = torch.empty_like(q)
    for a, b in zip(cu[:-1], cu[1:]):
        out[a:b] = _sdpa(q[a:b], k[a:b], v[a:b], scale, True)
    return out


def paged_attn_decode(q, k_cache, v_cache, block_tables, context_lens, scale):
    B = q.shape[0]
    block_size = k_cache.shape[1]
    out = torch.zeros_like(q)
    for b in range(B):
        n = int(context_lens[b])
        if n == 0:
            continue
        t = torch.arange(n)
        rows = block_tables[b, t // block_size].long() * block_size + t % block_size
        kk = k_cache.view(-1, *k_cache.shape[2:])[rows]
        vv = v_cache.view(-1, *v_cache.shape[2:])[rows]
        out[b:b + 1] = _sdpa(q[b:b + 1], kk, vv, scale, False)
    return out


def argmax_rows(logits):
    return logits.float().argmax(-1)
