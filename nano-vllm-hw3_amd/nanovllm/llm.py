"""User-facing name of the engine: ``from nanovllm import LLM`` constructs and drives an ``LLMEngine`` exactly as
in the reference (its nanovllm/llm.py:4-5 is the same one-line subclass); every keyword is a ``Config`` field."""
from nanovllm.engine import llm_engine as _engine


class LLM(_engine.LLMEngine):
    """``LLM(model_dir, **config_fields).generate(prompts, sampling_params)``; adds nothing to the engine."""
