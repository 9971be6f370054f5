import json
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, Optional

import torch


def load_hf_config(model_dir: str) -> SimpleNamespace:
    """config.json -> attribute namespace (no dependency on a particular transformers version)."""
    with open(os.path.join(model_dir, "config.json")) as f:
        d = json.load(f)
    rope_theta = d.get("rope_theta")
    if rope_theta is None:
        rope_theta = (d.get("rope_parameters") or {}).get("rope_theta", 1000000.0)
    d["rope_theta"] = rope_theta
    d.setdefault("head_dim", d["hidden_size"] // d["num_attention_heads"])
    d.setdefault("attention_bias", False)
    d.setdefault("tie_word_embeddings", False)
    d.setdefault("rms_norm_eps", 1e-6)
    d.setdefault("hidden_act", "silu")
    d.setdefault("max_position_embeddings", 40960)
    dt = d.get("torch_dtype", d.get("dtype", "bfloat16"))
    d["torch_dtype"] = getattr(torch, dt) if isinstance(dt, str) else dt
    return SimpleNamespace(**d)


@dataclass
class Config:
    """The reference's Config (nanovllm/config.py:9-35) plus ``load_format``: "auto" reads *.safetensors
    and quantizes as configured; "dummy" random-initialises the weights on the GPU (the benchmark has no
    checkpoint to read); "quantized" reads a pre-quantized checkpoint written by
    nanovllm.utils.checkpoint.save_quantized (its quant_config.json overrides quant_type / linear_dtype)."""
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: Optional[SimpleNamespace] = None
    eos: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1
    linear_dtype: torch.dtype = torch.bfloat16
    weight_quant_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None
    quant_type: Optional[str] = None   # "per_tensor", "per_row", "per_group", "per_row_weight_only"
    group_size: int = 128
    load_format: str = "auto"
    # opt-in, off by default because it changes the logits the reference computes in bf16: the vocabulary
    # projection (311 MB of bf16 per decode step for Qwen3-0.6B) as a per-row weight-only quantized linear
    quantize_lm_head: bool = False
    # "cuda" = the product path (HIP kernels).  "cpu" = plumbing only: the UNQUANTIZED model with plain torch
    # operators, eager, for checking engine wiring on a host without a GPU; every quantized option is refused there.
    device: str = "cuda"

    def __post_init__(self):
        assert os.path.isdir(self.model), f"{self.model} is not a directory"
        assert self.kvcache_block_size % 256 == 0
        assert 1 <= self.tensor_parallel_size <= 8
        self.hf_config = load_hf_config(self.model)
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
        assert self.max_num_batched_tokens >= self.max_model_len
        assert self.device in ("cuda", "cpu")
        if self.device == "cpu":
            quantized = (self.quant_type is not None or self.weight_quant_fn is not None or self.quantize_lm_head or
                         self.linear_dtype != torch.bfloat16 or self.load_format == "quantized")
            assert not quantized, "the quantized-linear path is HIP only: device='cpu' runs the unquantized model"
            assert self.tensor_parallel_size == 1
            self.enforce_eager = True
        if self.hf_config.__dict__.get("eos_token_id") is not None and self.eos == -1:
            e = self.hf_config.eos_token_id
            self.eos = e[0] if isinstance(e, (list, tuple)) else e
