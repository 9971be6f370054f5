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


def _cfg(blocks=64, max_num_seqs=8, block_size=256, budget=4096):
    return types.SimpleNamespace(max_num_seqs=max_num_seqs, max_num_batched_tokens=budget, eos=EOS,
                                 num_kvcache_blocks=blocks, kvcache_block_size=block_size)


def _model(seq, pos):
    """Next token after the first `pos` tokens of the sequence (depends on the prompt and the position only)."""
    return (seq.token_ids[0] * 31 + pos * 17) % 11


def _run(pipelined, specs, cfg=None, log=None):
    sch = Scheduler(cfg or _cfg())
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
        if log is not None:
            log.append((is_prefill, [s.seq_id for s in batch], sum(len(b.block_table) for b in batch)))
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


def test_admission_limits_and_prefill_priority():
    """A step is all-prefill or all-decode; admission is FIFO and stops at the sequence limit, at the token budget
    of one step, or when the KV blocks run out; waiting requests are admitted before anything decodes."""
    log = []
    specs = [(i + 1, 3, True) for i in range(7)]
    out = _run(False, specs, _cfg(max_num_seqs=3, budget=8), log)          # 3-token prompts: two fit the 8-token budget
    assert [len(o) for o in out] == [3] * 7
    prefills = [ids for pf, ids, _ in log if pf]
    assert all(len(ids) <= 2 for ids in prefills) and sum(len(i) for i in prefills) == 7
    first_decode = next(i for i, (pf, _, _) in enumerate(log) if not pf)
    assert all(pf for pf, _, _ in log[:first_decode]) and first_decode == 4  # 2+2+2+1 admitted before any decode
    assert all(len(ids) <= 3 for pf, ids, _ in log if not pf)                # decode steps respect max_num_seqs


def test_preemption_under_kv_pressure_gives_the_same_tokens():
    """With too few KV blocks the most recently admitted sequences are pushed back to the head of the queue and
    prefilled again later; outputs equal those of an unconstrained run and every block is returned."""
    specs = [(3, 14, True), (4, 14, True), (5, 9, True), (6, 14, False), (8, 11, True)]
    roomy = _run(False, specs, _cfg(blocks=64, block_size=4))
    log = []
    tight = _run(False, specs, _cfg(blocks=7, block_size=4), log)
    assert tight == roomy
    seen, again = set(), 0
    for pf, ids, _ in log:
        if pf:
            again += sum(1 for i in ids if i in seen)
            seen.update(ids)
    assert again > 0, "the pool was never tight enough to preempt"
    assert _run(True, specs, _cfg(blocks=7, block_size=4)) == roomy           # and with the one-deep pipeline
