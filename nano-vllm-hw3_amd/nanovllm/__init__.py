"""MI355X-native quantized-linear path for nano-vllm.  ``LLM`` / ``SamplingParams`` are imported
lazily so that the operator layer (``nanovllm.utils.quantization``) can be used on its own."""


def __getattr__(name):
    if name == "LLM":
        from nanovllm.llm import LLM
        return LLM
    if name == "SamplingParams":
        from nanovllm.sampling_params import SamplingParams
        return SamplingParams
    raise AttributeError(name)
