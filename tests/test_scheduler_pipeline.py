"""Host logic of the one-deep decode pipeline: the scheduler accounts for a token before its value is known
(advance) and fills it in later (commit).  No GPU: the model is a deterministic function of the sequence."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))

from nanovllm.engine.scheduler import Scheduler          # noqa: E402
from nanovllm.engine.sequence import Sequence            # noqa: E402
from nanovllm.sampling_params import SamplingParams      # noqa: E402

EOS = 7


def _cfg(blocks=64, max_num_seqs=8):
    return types.SimpleNamespace(max_num_seqs=max_num_seqs, max_num_batched_tokens=4096, eos=EOS,
                                 num_kvcache_blocks=blocks, kvcache_block_size=256)


def _model(seq, pos):
    """Next token after the first `pos` tokens of the sequence (depends on the prompt and the position only)."""
    return (seq.token_ids[0] * 31 + pos * 17) % 11


def _run(pipelined, specs):
    sch = Scheduler(_cfg())
    seqs = [Sequence([p, 1, 2], SamplingParams(max_tokens=m, ignore_eos=ig)) for p, m, ig in specs]
    for s in seqs:
        sch.add(s)
    done, pending = {}, None

    def drain(p):
        for s in sch.commit(*p):
            done[s.seq_id] = list(s.completion_token_ids)
    while True:
        if pending is not None and not pipelined:
            drain(pending); pending = None
        if sch.is_finished():
            if pending is None:
                break
            drain(pending); pending = None
            continue
        batch, is_prefill = sch.schedule()
        toks = [_model(s, len(s)) for s in batch]        # what the GPU would produce for this launch
        idx = sch.advance(batch)
        if pending is not None:
            drain(pending)
        pending = (batch, idx, toks)
        if is_prefill:
            drain(pending); pending = None
    assert len(sch.block_manager.free) == sch.block_manager.num_blocks
    return [done[s.seq_id] for s in seqs]


def test_advance_commit_equals_postprocess():
    specs = [(3, 5, True), (4, 9, True), (5, 1, True), (6, 12, False), (9, 12, False), (2, 30, False)]
    a, b = _run(False, specs), _run(True, specs)
    assert a == b
    for (p, m, ig), out in zip(specs, a):
        assert len(out) <= m
        if ig:
            assert len(out) == m
        else:
            assert EOS not in out[:-1]
            assert out[-1] == EOS or len(out) == m
    assert any(out[-1] == EOS and len(out) < m for (p, m, ig), out in zip(specs, a) if not ig)


def test_commit_drops_token_launched_beyond_an_eos():
    sch = Scheduler(_cfg())
    s = Sequence([1, 2, 3], SamplingParams(max_tokens=10, ignore_eos=False))
    sch.add(s)
    batch, _ = sch.schedule()
    sch.commit(batch, sch.advance(batch), [5])           # prefill
    b1, _ = sch.schedule(); i1 = sch.advance(b1)         # step 1 in flight
    b2, _ = sch.schedule(); i2 = sch.advance(b2)         # step 2 launched before step 1's token is known
    assert len(s) == 6
    assert sch.commit(b1, i1, [EOS]) == [s]              # EOS: finished, the pending token is cut
    assert s.is_finished and s.completion_token_ids == [5, EOS] and s.last_token == EOS
    assert sch.commit(b2, i2, [4]) == []                 # the late token is ignored
    assert s.completion_token_ids == [5, EOS] and sch.is_finished()
