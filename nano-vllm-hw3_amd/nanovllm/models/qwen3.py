"""Qwen3 decoder (the only model family of the reference, nanovllm/models/qwen3.py) as the host of
the four quantized-linear call sites per layer.  Everything between the GEMMs runs as one fused HIP
launch per operator (nanovllm/ops.py); attention is a flash-style varlen kernel for prefill and a paged
single-query kernel for decode."""
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from nanovllm import ops
from nanovllm.layers.linear import (MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear, _tp_world,
                                    tp_all_reduce)


@dataclass
class AttnContext:
    """Per-step attention metadata (the reference keeps it in a module global, utils/context.py)."""
    is_prefill: bool = False
    cu_seqlens: Optional[list] = None          # python list, prefill only
    slot_mapping: Optional[torch.Tensor] = None
    context_lens: Optional[torch.Tensor] = None
    block_tables: Optional[torch.Tensor] = None
    cu_seqlens_dev: Optional[torch.Tensor] = None   # cu_seqlens as int32 on the device (built on demand)
    max_seqlen: int = 0
    # optional [T, head_dim] f32: the RoPE table rows of this step's tokens gathered by the caller (measured no
    # faster than letting every layer's launch look its rows up, so the engine leaves it unset)
    cos_sin_rows: Optional[torch.Tensor] = None


_CTX = AttnContext()

# the qkv GEMM of a quantized model carries the q/k norm + RoPE + KV store in its epilogue (bit-identical to the
# separate operator on the same tile; NANOVLLM_NO_QKV_ROPE=1 keeps the two launches, for A/B measurements)
import os as _os
FUSE_QKV_ROPE = _os.environ.get("NANOVLLM_NO_QKV_ROPE", "") == ""
# The output projection runs on this package's bf16 GEMM, with the greedy sampler in its epilogue where the caller
# only wants tokens (NANOVLLM_NO_HEAD_ARGMAX=1 keeps torch's GEMM and the separate argmax launch, for A/B measurements)
# A W8A8 o_proj takes its input already quantized from the attention merge launch (NANOVLLM_NO_ATTN_QUANT=1 keeps the
# separate ql_quant_per_row launch, for A/B measurements; bit-identical results)
FUSE_ATTN_QUANT = _os.environ.get("NANOVLLM_NO_ATTN_QUANT", "") == ""
FUSE_HEAD_ARGMAX = _os.environ.get("NANOVLLM_NO_HEAD_ARGMAX", "") == ""


def set_context(ctx: AttnContext):
    global _CTX
    _CTX = ctx


def get_context() -> AttnContext:
    return _CTX


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)

    def forward(self, x, residual=None):
        out, res = ops.add_rmsnorm(x, residual, self.weight, self.eps)
        return out if residual is None else (out, res)


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.tp_rank, self.tp_size = _tp_world()
        assert num_embeddings % self.tp_size == 0
        self.per_part = num_embeddings // self.tp_size
        self.start = self.per_part * self.tp_rank
        self.weight = nn.Parameter(torch.empty(self.per_part, embedding_dim), requires_grad=False)
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param, loaded_weight):
        param.data.copy_(loaded_weight.narrow(0, self.start, self.per_part))

    def forward(self, x):
        if self.tp_size > 1:
            mask = (x >= self.start) & (x < self.start + self.per_part)
            x = mask * (x - self.start)
        y = F.embedding(x, self.weight)
        if self.tp_size > 1:
            y = mask.unsqueeze(1) * y
            y = tp_all_reduce(y)
        return y


def combine_shard_maxima(vals, toks):
    """vals f32 [tp, M], toks int64 [tp, M]: per-rank maxima over contiguous ascending vocabulary shards and their
    (global) tokens -> the tokens of the global maxima; among equal maxima the lowest rank, i.e. the smallest
    token, wins (argmax returns the first maximal index), and a row whose shards all report -inf gets rank 0's."""
    r = vals.t().argmax(dim=1, keepdim=True)
    return toks.t().gather(1, r)[:, 0]


def _gather_rows(t, world):
    """[M] from every rank -> [world, M] on every rank (gloo, the test transport, moves host memory)."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        host = t.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        return torch.stack(parts).to(t.device)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


class ParallelLMHead(VocabParallelEmbedding):
    """bf16 (unquantized, as in the reference) vocab-parallel output projection; with TP the shards'
    logits are all-gathered so every rank can take the same argmax."""

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__(num_embeddings, embedding_dim)
        self.quantized = None     # a per-row W8A16 QLinear over this rank's vocabulary shard (opt-in: quantize_lm_head)
        self.force_exchange = False    # tests: run the shard-maxima exchange (all_gather_into_tensor) on a one-rank group

    def fused_sampler(self, x):
        """True when greedy_tokens(x) can run: bf16 table (or this rank's shard of it) on this GPU, K a multiple of 32."""
        return (FUSE_HEAD_ARGMAX and self.quantized is None and x.is_cuda and
                x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16 and x.shape[-1] % 32 == 0)

    def greedy_tokens(self, x, out=None):
        """argmax(forward(x), -1) without the logits reaching memory: the sampler (layers/sampler.py:10-12) runs in
        the projection's epilogue; bit-identical to ops.argmax_rows(ops.lm_head_logits(x, weight)).  Vocabulary-
        parallel ranks take the maximum over their shard and exchange {winning logit, token} pairs (8 bytes per row
        and rank) where the reference gathers the shards' logits (embed_head.py:62-65)."""
        from nanovllm import ops
        if self.tp_size == 1 and not self.force_exchange:
            return ops.lm_head_argmax(x, self.weight, out=out)
        tok, val = ops.lm_head_argmax(x, self.weight, return_max=True)
        tok = tok + self.start
        vals, toks = _gather_rows(val, self.tp_size), _gather_rows(tok, self.tp_size)
        best = combine_shard_maxima(vals, toks)
        if out is None:
            return best
        out.copy_(best)
        return out

    def forward(self, x):
        if self.quantized is not None:
            logits = self.quantized(x)
        elif FUSE_HEAD_ARGMAX and x.is_cuda and x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16 \
                and x.dim() == 2 and x.shape[0] <= 1024 and x.shape[-1] % 32 == 0:
            from nanovllm import ops
            logits = ops.lm_head_logits(x.contiguous(), self.weight)      # (the kernel greedy_tokens shares)
        else:
            logits = F.linear(x, self.weight)
        if self.tp_size > 1:
            if dist.get_backend() == "gloo" and logits.is_cuda:     # test transport: gloo moves host memory
                host = logits.cpu()
                parts = [torch.empty_like(host) for _ in range(self.tp_size)]
                dist.all_gather(parts, host)
                logits = torch.cat(parts, -1).to(logits.device)
            else:
                parts = [torch.empty_like(logits) for _ in range(self.tp_size)]
                dist.all_gather(parts, logits)
                logits = torch.cat(parts, -1)
        return logits


def _takes_rowmax(linear):
    """A W8A8 module that can quantize its bf16 input inside the GEMM from partial row maxima."""
    from nanovllm.utils import quantization as NQ
    return NQ.FUSE_ACT_QUANT and isinstance(linear, NQ.QLinear) and not linear.weight_only and linear.qw.shape[1] % 32 == 0


class Qwen3Attention(nn.Module):
    def __init__(self, cfg, cos_sin):
        super().__init__()
        _, tp = _tp_world()
        self.num_heads = cfg.num_attention_heads // tp
        self.num_kv_heads = cfg.num_key_value_heads // tp
        self.head_dim = cfg.head_dim
        self.scaling = self.head_dim ** -0.5
        self.eps = cfg.rms_norm_eps
        self.qkv_bias = cfg.attention_bias
        self.qkv_proj = QKVParallelLinear(cfg.hidden_size, self.head_dim, cfg.num_attention_heads,
                                          cfg.num_key_value_heads, bias=self.qkv_bias)
        self.o_proj = RowParallelLinear(cfg.num_attention_heads * self.head_dim, cfg.hidden_size, bias=False)
        if not self.qkv_bias:
            self.q_norm = RMSNorm(self.head_dim, eps=self.eps)
            self.k_norm = RMSNorm(self.head_dim, eps=self.eps)
        self.cos_sin = cos_sin
        self.k_cache = self.v_cache = torch.tensor([])

    def forward(self, positions, hidden_states, quantized=None):
        ctx = get_context()
        qw = self.q_norm.weight if not self.qkv_bias else None
        kw = self.k_norm.weight if not self.qkv_bias else None
        if (FUSE_QKV_ROPE and hasattr(self.qkv_proj, "qw") and self.head_dim == 128 and self.qkv_proj.qw.shape[1] % 32 == 0
                and self.qkv_proj.fusable):
            # quantized qkv projection: q/k norm + RoPE + KV-cache store run in the GEMM's epilogue (one launch)
            from nanovllm.utils.quantization import qkv_rope_matmul
            rows = ctx.cos_sin_rows
            q, k, v = qkv_rope_matmul(self.qkv_proj, hidden_states, positions if rows is None else None,
                                      self.cos_sin if rows is None else rows, qw, kw, self.eps,
                                      self.num_heads, self.num_kv_heads, self.head_dim, self.k_cache, self.v_cache,
                                      ctx.slot_mapping, want_kv=ctx.is_prefill, quantized=quantized)
        else:
            qkv = self.qkv_proj(hidden_states) if quantized is None else self.qkv_proj.forward_quantized(*quantized)
            q, k, v = ops.qknorm_rope_store(qkv, positions, self.cos_sin, qw, kw, self.eps, self.num_heads,
                                            self.num_kv_heads, self.head_dim, self.k_cache, self.v_cache,
                                            ctx.slot_mapping, want_kv=ctx.is_prefill)
        if ctx.is_prefill:
            cu_dev = ctx.cu_seqlens_dev
            if cu_dev is None:
                cu_dev = torch.tensor(ctx.cu_seqlens, dtype=torch.int32, device=q.device)
            max_len = ctx.max_seqlen or max(b - a for a, b in zip(ctx.cu_seqlens[:-1], ctx.cu_seqlens[1:]))
            o = ops.attn_prefill(q, k, v, cu_dev, max_len, self.scaling)
        elif FUSE_ATTN_QUANT and getattr(self.o_proj, "takes_quantized_input", False) and q.is_cuda:
            # a W8A8 o_proj: the attention merge launch also quantizes every output row (the per_row_quant(x) of
            # per_row_matmul, quantization.py:323), so the projection starts from (q, scale) with no launch in between
            o, qx, sx = ops.paged_attn_decode(q, self.k_cache, self.v_cache, ctx.block_tables, ctx.context_lens,
                                              self.scaling, quant_dtype=self.o_proj.dtype)
            return self.o_proj.forward_partial_quantized(qx, sx)
        elif _takes_rowmax(self.o_proj):
            # a W8A8 o_proj: the attention launches also emit the partial row maxima of their output, and the
            # projection quantizes its input inside the GEMM (no per_row_quant launch in between)
            o, rowmax = ops.paged_attn_decode(q, self.k_cache, self.v_cache, ctx.block_tables, ctx.context_lens,
                                              self.scaling, return_rowmax=True)
            return self.o_proj.forward_with_rowmax(o.view(o.shape[0], -1), rowmax)
        else:
            o = ops.paged_attn_decode(q, self.k_cache, self.v_cache, ctx.block_tables, ctx.context_lens, self.scaling)
        o = o.view(o.shape[0], -1)
        # (a quantized o_proj may leave its split-K reduction to the residual add + RMSNorm launch behind it)
        return self.o_proj.forward_partial(o) if hasattr(self.o_proj, "forward_partial") else self.o_proj(o)


class Qwen3MLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        assert cfg.hidden_act == "silu"
        self.gate_up_proj = MergedColumnParallelLinear(cfg.hidden_size, [cfg.intermediate_size] * 2, bias=False)
        self.down_proj = RowParallelLinear(cfg.intermediate_size, cfg.hidden_size, bias=False)

    def forward(self, x, quantized=None):
        if quantized is not None and getattr(self.gate_up_proj, "silu_fused", False) and _takes_rowmax(self.down_proj):
            # W8A8 MLP: the gate/up GEMM's SiLU epilogue also emits the partial row maxima of its output, and the
            # down projection quantizes its input inside the GEMM
            h, rowmax = self.gate_up_proj.forward_quantized(*quantized, return_rowmax=True)
            return self.down_proj.forward_with_rowmax(h, rowmax)
        h = self.gate_up_proj(x) if quantized is None else self.gate_up_proj.forward_quantized(*quantized)
        if not getattr(self.gate_up_proj, "silu_fused", False):   # else the GEMM epilogue already applied SiLU * mul
            h = ops.silu_mul(h)
        return self.down_proj.forward_partial(h) if hasattr(self.down_proj, "forward_partial") else self.down_proj(h)


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, cfg, cos_sin):
        super().__init__()
        self.self_attn = Qwen3Attention(cfg, cos_sin)
        self.mlp = Qwen3MLP(cfg)
        self.input_layernorm = RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)

    def forward(self, positions, hidden_states, residual):
        # A W8A8 linear (dynamic per-row activation quantization) takes its input already quantized: the
        # RMSNorm launch in front of it emits (q, scale) directly instead of the bf16 row.
        qkv, gate_up = self.self_attn.qkv_proj, self.mlp.gate_up_proj
        if getattr(qkv, "takes_quantized_input", False) and hidden_states.dim() == 2:
            q, s, residual = ops.add_rmsnorm_quant(hidden_states, residual, self.input_layernorm.weight,
                                                   self.input_layernorm.eps, qkv.dtype)
            hidden_states = self.self_attn(positions, None, quantized=(q, s))
        else:
            if residual is None:
                hidden_states, residual = self.input_layernorm(hidden_states), hidden_states
            else:
                hidden_states, residual = self.input_layernorm(hidden_states, residual)
            hidden_states = self.self_attn(positions, hidden_states)
        if getattr(gate_up, "takes_quantized_input", False) and hidden_states.dim() == 2:
            q, s, residual = ops.add_rmsnorm_quant(hidden_states, residual, self.post_attention_layernorm.weight,
                                                   self.post_attention_layernorm.eps, gate_up.dtype)
            return self.mlp(None, quantized=(q, s)), residual
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


def build_cos_sin(head_dim, max_position, base, device):
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float, device=device) / head_dim))
    t = torch.arange(max_position, dtype=torch.float, device=device)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).contiguous()   # [max_pos, head_dim] f32


class Qwen3Model(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        dev = torch.empty(0).device
        self.cos_sin = build_cos_sin(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta, dev)
        self.embed_tokens = VocabParallelEmbedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([Qwen3DecoderLayer(cfg, self.cos_sin) for _ in range(cfg.num_hidden_layers)])
        self.norm = RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)

    def forward(self, input_ids, positions):
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class Qwen3ForCausalLM(nn.Module):
    packed_modules_mapping = {
        "q_proj": ("qkv_proj", "q"), "k_proj": ("qkv_proj", "k"), "v_proj": ("qkv_proj", "v"),
        "gate_proj": ("gate_up_proj", 0), "up_proj": ("gate_up_proj", 1),
    }

    def __init__(self, cfg):
        super().__init__()
        self.model = Qwen3Model(cfg)
        self.lm_head = ParallelLMHead(cfg.vocab_size, cfg.hidden_size)
        if cfg.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight

    def forward(self, input_ids, positions):
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states):
        return self.lm_head(hidden_states)

    def compute_tokens(self, hidden_states, out=None):
        """Greedy tokens of the given rows; the caller checks lm_head.fused_sampler(hidden_states) first."""
        return self.lm_head.greedy_tokens(hidden_states, out=out)
