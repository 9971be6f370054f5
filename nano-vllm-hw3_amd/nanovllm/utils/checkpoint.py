"""Pre-quantized checkpoints: the quantized weights and scales of the four linears per layer written
once, so that start-up neither reads bf16 weights nor re-quantizes them (the reference re-quantizes at
every start, nanovllm/engine/model_runner.py:36-57, and its Q*Linear tensors are invisible to
``state_dict``, nanovllm/utils/quantization.py:355, 373, 392).

Layout of ``<dir>``:
  config.json          the model's HF config, copied
  quant_config.json    {"format_version", "quant_type", "linear_dtype", "group_size", "weight_only"}
  model.safetensors    ``<linear>.qweight`` uint8 [N, K] (the int8 / e4m3fn bit patterns, rows in the packed
                       [q|k|v] / [gate|up] order of the unsharded model), ``<linear>.scales`` f32 ([N] per row,
                       [1] per tensor, [N, K/g] per group), ``<linear>.bias`` if any, and every other
                       parameter (embeddings, norms) under its own name.

Tensor parallelism: every rank reads the full tensors and keeps its shard through the same
``weight_loader`` the bf16 path uses, so q/k/v and gate/up land where the kernels expect them.
Column-parallel shards keep their rows' scales; a row-parallel shard takes its K columns of ``qweight``
and, per row, the FULL row's scale (per group: its K/g columns of scales), so the sharded model
multiplies exactly the same quantized weights as the unsharded one.
"""
import json
import os
import shutil

import torch

FORMAT_VERSION = 1
_LINEARS = (("self_attn", "qkv_proj"), ("self_attn", "o_proj"), ("mlp", "gate_up_proj"), ("mlp", "down_proj"))
_DT = {"int8": torch.int8, "float8_e4m3fn": torch.float8_e4m3fn}


def deinterleave_gate_up(t: torch.Tensor):
    """Inverse of quantization.interleave_gate_up: 32-row strips alternating gate, up -> [gate | up]."""
    I = t.shape[0] // 2
    rest = tuple(t.shape[1:])
    s = t.reshape(I // 32, 2, 32, *rest)
    return torch.cat([s[:, 0].reshape(I, *rest), s[:, 1].reshape(I, *rest)], dim=0).contiguous()


def _quant_meta(mod):
    from nanovllm.utils import quantization as NQ
    if isinstance(mod, NQ.QGroupLinear):
        return "per_group", mod.group_size, True
    if isinstance(mod, NQ.QTensorLinear):
        return "per_tensor", 0, True
    if isinstance(mod, NQ.QLinear):
        return ("per_row_weight_only" if mod.weight_only else "per_row"), 0, mod.weight_only
    raise TypeError(f"{type(mod).__name__} is not a quantized linear")


@torch.no_grad()
def save_quantized(model, hf_config_dir: str, out_dir: str):
    """Write the (already quantized, unsharded) model.  Run on one GPU with tensor_parallel_size = 1."""
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    tensors = {n: p.data.detach().cpu().contiguous() for n, p in model.named_parameters()}
    meta = None
    for i, layer in enumerate(model.model.layers):
        for parent, name in _LINEARS:
            mod = getattr(getattr(layer, parent), name)
            assert getattr(mod, "tp_size", 1) == 1, "save from an unsharded (TP=1) model"
            m = _quant_meta(mod) + (str(mod.dtype).replace("torch.", ""),)
            assert meta in (None, m), "mixed quantization schemes are not representable"
            meta = m
            qw, sw, bias = mod.rowmajor_qw(), mod.sw, mod.bias
            if mod.silu_fused:
                qw = deinterleave_gate_up(qw)
                sw = deinterleave_gate_up(sw) if sw.dim() >= 1 and sw.shape[0] == qw.shape[0] else sw
                bias = deinterleave_gate_up(bias.data) if bias is not None else None
            pre = f"model.layers.{i}.{parent}.{name}"
            tensors[pre + ".qweight"] = qw.view(torch.uint8).cpu().contiguous()
            tensors[pre + ".scales"] = sw.detach().float().reshape(-1 if sw.dim() < 2 else sw.shape).cpu().contiguous()
            if bias is not None:
                tensors[pre + ".bias"] = bias.detach().cpu().contiguous()
    save_file(tensors, os.path.join(out_dir, "model.safetensors"))
    with open(os.path.join(out_dir, "quant_config.json"), "w") as f:
        json.dump({"format_version": FORMAT_VERSION, "quant_type": meta[0], "group_size": meta[1],
                   "weight_only": meta[2], "linear_dtype": meta[3]}, f, indent=1)
    src = os.path.join(hf_config_dir, "config.json")
    if os.path.abspath(src) != os.path.abspath(os.path.join(out_dir, "config.json")):
        shutil.copy(src, os.path.join(out_dir, "config.json"))


def read_quant_config(path: str):
    with open(os.path.join(path, "quant_config.json")) as f:
        qc = json.load(f)
    assert qc.get("format_version") == FORMAT_VERSION, f"unsupported quantized checkpoint version {qc.get('format_version')}"
    assert qc["linear_dtype"] in _DT and qc["quant_type"] in ("per_row", "per_row_weight_only", "per_tensor", "per_group")
    return qc


def _through_loader(lin, full: torch.Tensor, parts=None):
    """This rank's shard of ``full`` (rows packed as in the unsharded model), cut by the bf16 module's own
    weight_loader.  ``parts`` = [(shard_id, rows)] for packed parameters."""
    shape = list(lin.weight.shape)
    if full.dim() == 2 and full.shape[1] != (lin.weight.shape[1] * (lin.tp_size if lin.tp_dim == 1 else 1)):
        # a scales tensor: same rows as the weight, its own number of columns
        div = lin.tp_size if lin.tp_dim == 1 else 1
        assert full.shape[1] % div == 0, "a quantization group may not straddle two tensor-parallel shards"
        cols = full.shape[1] // div
        shape = [shape[0], cols]
    dst = torch.nn.Parameter(torch.empty(shape, dtype=full.dtype, device=full.device), requires_grad=False)
    if parts is None:
        lin.weight_loader(dst, full)
    else:
        row = 0
        for shard_id, rows in parts:
            lin.weight_loader(dst, full.narrow(0, row, rows), shard_id)
            row += rows
    return dst.data


def shard_quantized(lin, qw, sw, bias, quant_type):
    """(qw, sw, bias) of the unsharded linear -> the tensors of this rank's shard of ``lin``."""
    from nanovllm.layers.linear import MergedColumnParallelLinear, QKVParallelLinear
    parts = None
    if isinstance(lin, QKVParallelLinear):
        tp = lin.tp_size
        parts = [("q", lin.num_heads * tp * lin.head_size), ("k", lin.num_kv_heads * tp * lin.head_size),
                 ("v", lin.num_kv_heads * tp * lin.head_size)]
    elif isinstance(lin, MergedColumnParallelLinear):
        parts = list(enumerate(lin.output_sizes))
    q = _through_loader(lin, qw, parts)
    if quant_type == "per_tensor":
        s = sw
    elif sw.dim() == 1:
        # per row: a column-parallel shard keeps its rows' scales, a row-parallel shard needs every row's
        s = _through_loader(lin, sw.unsqueeze(1), parts).squeeze(1) if lin.tp_dim == 0 else sw
    else:
        s = _through_loader(lin, sw, parts)
    b = None
    if bias is not None:
        b = _through_loader(lin, bias.unsqueeze(1), parts).squeeze(1) if lin.tp_dim == 0 else bias
    return q.contiguous(), s.contiguous(), b


@torch.no_grad()
def load_quantized(model, path: str, fuse_silu: bool = True):
    """Fill ``model`` (bf16 skeleton on the GPU, weights uninitialised) from a quantized checkpoint: the
    four linears per layer become Q*Linear modules built directly from the stored tensors."""
    from safetensors import safe_open
    from nanovllm.utils import quantization as NQ
    from nanovllm.utils.loader import _assign
    qc = read_quant_config(path)
    dtype = _DT[qc["linear_dtype"]]
    params = dict(model.named_parameters())
    dev = next(model.parameters()).device
    with safe_open(os.path.join(path, "model.safetensors"), "pt", "cpu") as f:
        keys = set(f.keys())
        for i, layer in enumerate(model.model.layers):
            for parent, name in _LINEARS:
                pre = f"model.layers.{i}.{parent}.{name}"
                lin = getattr(getattr(layer, parent), name)
                qw = f.get_tensor(pre + ".qweight").to(dev).view(dtype)
                sw = f.get_tensor(pre + ".scales").to(dev)
                bias = f.get_tensor(pre + ".bias").to(dev) if pre + ".bias" in keys else None
                keys -= {pre + ".qweight", pre + ".scales", pre + ".bias"}
                q, s, b = shard_quantized(lin, qw, sw, bias, qc["quant_type"])
                if qc["quant_type"] == "per_group":
                    mod = NQ.QGroupLinear.from_quantized(lin, dtype, q, s, b, qc["group_size"])
                elif qc["quant_type"] == "per_tensor":
                    mod = NQ.QTensorLinear.from_quantized(lin, dtype, q, s, b)
                else:
                    mod = NQ.QLinear.from_quantized(lin, dtype, q, s, b, weight_only=qc["weight_only"])
                if name == "gate_up_proj" and fuse_silu and (q.shape[0] // 2) % 32 == 0:
                    mod.enable_silu_mul()
                mod.tile()
                setattr(getattr(layer, parent), name, mod)
        for name in sorted(keys):
            _assign(model, params, name, f.get_tensor(name))
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return qc
