from dataclasses import dataclass


@dataclass
class SamplingParams:
    """Same fields as the reference (nanovllm/sampling_params.py:4-8).  The sampler is greedy
    (argmax), so ``temperature`` is accepted and ignored, as in the reference fork
    (nanovllm/layers/sampler.py:10-12)."""
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False
