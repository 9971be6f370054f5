import os
from dataclasses import fields

import torch
import torch.distributed as dist

from nanovllm.config import Config
from nanovllm.engine.model_runner import ModelRunner
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import Sequence
from nanovllm.sampling_params import SamplingParams


class LLMEngine:
    """``LLM(model_dir, **config_fields).generate(prompts, sampling_params)`` as in the reference
    (nanovllm/engine/llm_engine.py).  Tensor parallelism is SPMD: launch one process per GPU with
    ``python -m torch.distributed.run --nproc-per-node N``; every rank builds the same engine and
    calls generate() with the same arguments, and the collectives live inside the model."""

    def __init__(self, model, **kwargs):
        names = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in names})
        self.rank = 0
        if config.tensor_parallel_size > 1:
            if not dist.is_initialized():
                rank = int(os.environ.get("RANK", "0"))
                world = int(os.environ.get("WORLD_SIZE", "1"))
                assert world == config.tensor_parallel_size, (
                    f"tensor_parallel_size={config.tensor_parallel_size} needs that many ranks "
                    f"(torch.distributed.run), got WORLD_SIZE={world}")
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
                # "nccl" is RCCL on ROCm; NANOVLLM_DIST_BACKEND=gloo lets two ranks share one GPU in tests
                dist.init_process_group(os.environ.get("NANOVLLM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
            self.rank = dist.get_rank()
        self.config = config
        self.model_runner = ModelRunner(config, self.rank)
        self.tokenizer = None
        self.scheduler = Scheduler(config)

    def _tok(self):
        if self.tokenizer is None:      # token-id prompts never need a tokenizer
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(self.config.model, use_fast=True)
            self.config.eos = self.scheduler.eos = self.tokenizer.eos_token_id
        return self.tokenizer

    def add_request(self, prompt, sampling_params: SamplingParams):
        if isinstance(prompt, str):
            prompt = self._tok().encode(prompt)
        self.scheduler.add(Sequence(prompt, sampling_params))

    def step(self):
        seqs, is_prefill = self.scheduler.schedule()
        token_ids = self.model_runner.run(seqs, is_prefill)
        self.scheduler.postprocess(seqs, token_ids)
        outputs = [(s.seq_id, s.completion_token_ids) for s in seqs if s.is_finished]
        num_tokens = sum(len(s) for s in seqs) if is_prefill else -len(seqs)
        return outputs, num_tokens

    def is_finished(self):
        return self.scheduler.is_finished()

    def generate(self, prompts, sampling_params, use_tqdm: bool = False):
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for p, sp in zip(prompts, sampling_params):
            self.add_request(p, sp)
        outputs = {}
        while not self.is_finished():
            out, _ = self.step()
            for seq_id, toks in out:
                outputs[seq_id] = toks
        outs = [outputs[k] for k in sorted(outputs)]
        decode = (lambda t: self.tokenizer.decode(t)) if self.tokenizer is not None else (lambda t: "")
        return [{"text": decode(t), "token_ids": t} for t in outs]
