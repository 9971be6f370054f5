import os
from glob import glob

import torch


def default_weight_loader(param, loaded_weight):
    param.data.copy_(loaded_weight)


def _assign(model, params, name, tensor):
    """Hand one checkpoint tensor to its parameter's ``weight_loader`` (q/k/v and gate/up go into the
    packed parameters with their shard id), as the reference's nanovllm/utils/loader.py:12-28 does."""
    packed = getattr(model, "packed_modules_mapping", {})
    for k, (v, shard_id) in packed.items():
        if k in name:
            pname = name.replace(k, v)
            if pname in params:
                p = params[pname]
                p.weight_loader(p, tensor, shard_id)
            return
    if name in params:
        p = params[name]
        getattr(p, "weight_loader", default_weight_loader)(p, tensor)


def load_model(model, path: str):
    """Read every *.safetensors under ``path``; every rank reads the full tensors and keeps its shard."""
    from safetensors import safe_open
    params = dict(model.named_parameters())
    for file in sorted(glob(os.path.join(path, "*.safetensors"))):
        with safe_open(file, "pt", "cpu") as f:
            for name in f.keys():
                _assign(model, params, name, f.get_tensor(name))


@torch.no_grad()
def init_dummy_weights(model, cfg, seed: int = 0, std: float = 0.02):
    """Random weights for benchmarks without a checkpoint.  The FULL (unsharded) tensors of a Qwen3
    checkpoint are drawn in a fixed order from one seeded generator and passed through the same
    weight loaders a real checkpoint goes through, so every tensor-parallel rank holds its shard of
    the same model and TP=N reproduces TP=1 up to reduction order."""
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    params = dict(model.named_parameters())
    H, D, I = cfg.hidden_size, cfg.head_dim, cfg.intermediate_size
    Hq, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads

    def rnd(*shape):
        return (torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * std).to(cfg.torch_dtype)

    def ones(n):
        return torch.ones(n, device=dev, dtype=cfg.torch_dtype)

    _assign(model, params, "model.embed_tokens.weight", rnd(cfg.vocab_size, H))
    if not cfg.tie_word_embeddings:
        _assign(model, params, "lm_head.weight", rnd(cfg.vocab_size, H))
    _assign(model, params, "model.norm.weight", ones(H))
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        _assign(model, params, pre + "input_layernorm.weight", ones(H))
        _assign(model, params, pre + "post_attention_layernorm.weight", ones(H))
        _assign(model, params, pre + "self_attn.q_proj.weight", rnd(Hq * D, H))
        _assign(model, params, pre + "self_attn.k_proj.weight", rnd(Hkv * D, H))
        _assign(model, params, pre + "self_attn.v_proj.weight", rnd(Hkv * D, H))
        _assign(model, params, pre + "self_attn.o_proj.weight", rnd(H, Hq * D))
        if cfg.attention_bias:
            _assign(model, params, pre + "self_attn.q_proj.bias", rnd(Hq * D))
            _assign(model, params, pre + "self_attn.k_proj.bias", rnd(Hkv * D))
            _assign(model, params, pre + "self_attn.v_proj.bias", rnd(Hkv * D))
        else:
            _assign(model, params, pre + "self_attn.q_norm.weight", ones(D))
            _assign(model, params, pre + "self_attn.k_norm.weight", ones(D))
        _assign(model, params, pre + "mlp.gate_proj.weight", rnd(I, H))
        _assign(model, params, pre + "mlp.up_proj.weight", rnd(I, H))
        _assign(model, params, pre + "mlp.down_proj.weight", rnd(H, I))
