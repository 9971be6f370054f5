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
        self.pipeline_decode = os.environ.get("NANOVLLM_NO_PIPELINE", "") == ""

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
        done = self.scheduler.postprocess(seqs, token_ids)
        outputs = [(s.seq_id, s.completion_token_ids) for s in done]
        num_tokens = sum(len(s) for s in seqs) if is_prefill else -len(seqs)
        return outputs, num_tokens

    def _can_pipeline(self):
        """The next step may be launched before the tokens of the one in flight are known if it can only be a
        decode step over a subset of the same sequences: nothing is waiting for prefill, every running sequence
        is in every decode step, and there are enough free KV blocks that nothing can be preempted."""
        sch = self.scheduler
        return (self.pipeline_decode and not sch.waiting and len(sch.running) <= sch.max_num_seqs and
                len(sch.block_manager.free) >= len(sch.running) and self.model_runner.can_stage(sch.running))

    def is_finished(self):
        return self.scheduler.is_finished()

    def generate(self, prompts, sampling_params, use_tqdm: bool = False):
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for p, sp in zip(prompts, sampling_params):
            self.add_request(p, sp)
        outputs = {}
        sch, runner = self.scheduler, self.model_runner
        pending = None                      # (seqs, token indices, handle, covers) of the step in flight

        def drain(p):
            for s in sch.commit(p[0], p[1], runner.wait(p[2])):
                outputs[s.seq_id] = s.completion_token_ids

        # Decode steps are pipelined one deep: step i+1 is scheduled and launched while step i runs, its input
        # ids gathered on the device from step i's sampled tokens, and only then are step i's tokens read on
        # the host.  The GPU therefore always has the next step queued; results are those of the step-by-step
        # loop (the reference's llm_engine.py:66-85), an EOS merely ends its sequence one launch later.
        while True:
            if pending is not None and not (pending[3] and self._can_pipeline()):
                drain(pending)
                pending = None
            if self.is_finished():
                if pending is None:
                    break
                drain(pending)
                pending = None
                continue
            seqs, is_prefill = sch.schedule()
            handle = runner.launch(seqs, is_prefill, prev=pending[2] if pending is not None else None)
            covers = len(seqs) == len(sch.running)      # a later step can only hold sequences of this one
            idx = sch.advance(seqs)
            if pending is not None:
                drain(pending)
            pending = (seqs, idx, handle, covers)
            if is_prefill:
                drain(pending)
                pending = None
        outs = [outputs[k] for k in sorted(outputs)]
        decode = (lambda t: self.tokenizer.decode(t)) if self.tokenizer is not None else (lambda t: "")
        return [{"text": decode(t), "token_ids": t} for t in outs]
