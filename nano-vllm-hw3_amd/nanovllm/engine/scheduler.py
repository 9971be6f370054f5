from collections import deque

from nanovllm.engine.block_manager import BlockManager
from nanovllm.engine.sequence import ACTIVE, DONE, QUEUED, Sequence


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
        """One step's batch: (sequences, is_prefill).  Admission has strict priority over decoding."""
        batch = self._admit()
        if batch:
            return batch, True
        batch = self._decode_batch()
        assert batch, "nothing to run: a single sequence does not fit the KV cache"
        return batch, False

    def _admit(self):
        """Move requests from the head of the queue into the running set while the sequence count, the token
        budget of one step and the free KV blocks allow; FIFO, no skipping ahead."""
        bm, batch, budget = self.block_manager, [], self.max_num_batched_tokens
        while self.waiting and len(batch) < self.max_num_seqs:
            seq = self.waiting[0]
            if len(seq) > budget or not bm.can_allocate(seq):
                break
            self.waiting.popleft()
            bm.allocate(seq)
            budget -= len(seq)
            seq.state = ACTIVE
            batch.append(seq)
        self.running.extend(batch)
        return batch

    def _decode_batch(self):
        """Every running sequence gets its next token; when the cache cannot hold one more token for all of
        them, the most recently admitted ones go back to the head of the queue (their blocks are freed and they
        will be prefilled again), the sequence under consideration itself last."""
        bm = self.block_manager
        keep = []
        candidates = deque(self.running)
        while candidates and len(keep) < self.max_num_seqs:
            seq = candidates.popleft()
            evicted_self = False
            while not bm.can_append(seq):
                victim = candidates.pop() if candidates else seq
                self.preempt(victim)
                if victim is seq:
                    evicted_self = True
                    break
            if not evicted_self:
                bm.may_append(seq)
                keep.append(seq)
        self.running = deque(keep + list(candidates))
        return keep

    def preempt(self, seq):
        if seq in self.running:
            self.running.remove(seq)
        self.block_manager.deallocate(seq)
        seq.state = QUEUED
        self.waiting.appendleft(seq)

    PENDING = 0   # stands in for a token that the GPU is still computing

    def _finish(self, seq):
        seq.state = DONE
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
