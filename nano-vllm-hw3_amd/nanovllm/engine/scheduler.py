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

    PENDING = 0   # stands in for a token that the GPU is still computing

    def _finish(self, seq):
        seq.status = SequenceStatus.FINISHED
        self.block_manager.deallocate(seq)
        self.running.remove(seq)

    def advance(self, seqs):
        """Account for the token every sequence of a launched step will receive, before its value is known:
        lengths, block tables and the max_tokens limit only depend on the COUNT of tokens, so the next step
        can be scheduled while this one runs.  Returns the index each token will occupy."""
        idx = []
        for seq in seqs:
            idx.append(seq.num_tokens)
            seq.append_token(self.PENDING)
            if seq.num_completion_tokens == seq.max_tokens:
                self._finish(seq)
        return idx

    def commit(self, seqs, idx, token_ids):
        """Fill in the tokens of a step that advance() accounted for; returns the sequences whose output is
        complete.  An EOS ends its sequence one step late when the next step was launched before the EOS
        reached the host: that step's token (beyond the EOS) is dropped."""
        done = []
        for seq, i, tok in zip(seqs, idx, token_ids):
            if i >= seq.num_tokens:            # cut off by an EOS of an earlier step
                continue
            seq.token_ids[i] = tok
            if i == seq.num_tokens - 1:
                seq.last_token = tok
            if not seq.ignore_eos and tok == self.eos:
                del seq.token_ids[i + 1:]
                seq.num_tokens = i + 1
                seq.last_token = tok
                if not seq.is_finished:
                    self._finish(seq)
            if seq.is_finished and i == seq.num_tokens - 1:
                done.append(seq)
        return done

    def postprocess(self, seqs, token_ids):
        return self.commit(seqs, self.advance(seqs), token_ids)
