from collections import deque

from nanovllm.engine.block_manager import BlockManager
from nanovllm.engine.sequence import Sequence, SequenceStatus


class Scheduler:
    """Prefill-first continuous batching with tail preemption: a step is either all-prefill (as many
    waiting sequences as fit max_num_seqs / max_num_batched_tokens / free KV blocks) or all-decode
    (behaviour of the reference's nanovllm/engine/scheduler.py:24-71)."""

    def __init__(self, config):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting = deque()
        self.running = deque()

    def is_finished(self):
        return not self.waiting and not self.running

    def add(self, seq: Sequence):
        self.waiting.append(seq)

    def schedule(self):
        picked, tokens = [], 0
        while self.waiting and len(picked) < self.max_num_seqs:
            seq = self.waiting[0]
            if tokens + len(seq) > self.max_num_batched_tokens or not self.block_manager.can_allocate(seq):
                break
            self.block_manager.allocate(seq)
            tokens += len(seq)
            seq.status = SequenceStatus.RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            picked.append(seq)
        if picked:
            return picked, True
        while self.running and len(picked) < self.max_num_seqs:
            seq = self.running.popleft()
            while not self.block_manager.can_append(seq):
                if self.running:
                    self.preempt(self.running.pop())
                else:
                    self.preempt(seq)
                    break
            else:
                self.block_manager.may_append(seq)
                picked.append(seq)
        assert picked
        self.running.extendleft(reversed(picked))
        return picked, False

    def preempt(self, seq):
        seq.status = SequenceStatus.WAITING
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    def postprocess(self, seqs, token_ids):
        for seq, tok in zip(seqs, token_ids):
            seq.append_token(tok)
            if (not seq.ignore_eos and tok == self.eos) or seq.num_completion_tokens == seq.max_tokens:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
