"""Output projection on the package's bf16 GEMM, and the greedy sampler in its epilogue (through the C ABI).

What is asserted: the logits against the exact product within the f32-accumulation tolerance (the reference is
``F.linear(x, weight)`` in bf16, nanovllm/layers/embed_head.py:61, whose summation order is not specified either);
the fused tokens against ``ql_argmax_rows(ql_lmhead_logits(...))`` BIT FOR BIT, including exact ties (smallest
index wins, nanovllm/layers/sampler.py:10-12 takes ``argmax``), rows without a finite maximum, ragged vocabulary
sizes and every tile the output projection is built for; and that the engine's captured graphs, which end in the
fused sampler, produce the tokens of the eager logits + argmax path.
"""
import numpy as np
import pytest
import torch

from oracle import qlinear_oracle as O
from tests.util import assert_gemm_close, bits16

pytestmark = pytest.mark.gpu

HEAD_TILES = [(1, 4, 1, 4), (2, 4, 1, 4), (4, 4, 1, 4), (4, 8, 1, 4), (8, 8, 1, 2), (5, 8, 1, 4), (6, 8, 1, 4), (7, 8, 1, 4),
              (5, 4, 1, 4)]    # QL_TILE_LIST_HEAD


def _ops():
    from nanovllm import ops, _qlinear as Q
    return ops, Q


def _rand(M, V, K, seed, wstd=0.05):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(V, K, generator=g) * wstd).to(torch.bfloat16).cuda()
    return x, w


@pytest.mark.parametrize("M,V,K", [(1, 1000, 64), (5, 4096, 1024), (33, 1000, 256), (64, 3000, 1024), (100, 2048, 96),
                                   (200, 5000, 1024), (256, 1184, 2048), (300, 1000, 128)])
def test_logits_against_exact_product(M, V, K):
    ops, _ = _ops()
    x, w = _rand(M, V, K, seed=M + V)
    y = ops.lm_head_logits(x, w)
    xf = O.bf16_to_f32(bits16(x)).astype(np.float64)
    wf = O.bf16_to_f32(bits16(w)).astype(np.float64)
    assert_gemm_close(bits16(y), xf @ wf.T, np.abs(xf) @ np.abs(wf).T, what=f"lm_head_logits {M}x{V}x{K}")


@pytest.mark.parametrize("M,V,K", [(1, 1000, 64), (5, 4096, 1024), (32, 129, 32), (33, 1000, 256), (64, 3000, 1024),
                                   (100, 2048, 96), (128, 151936, 1024), (200, 5000, 1024), (256, 1184, 2048),
                                   (300, 1000, 128), (513, 777, 64)])
def test_fused_sampler_bit_identical_to_logits_then_argmax(M, V, K):
    ops, _ = _ops()
    x, w = _rand(M, V, K, seed=7 * M + V)
    logits = ops.lm_head_logits(x, w)
    want = ops.argmax_rows(logits) if V % 8 == 0 else logits.float().argmax(-1)
    got = ops.lm_head_argmax(x, w)
    assert got.dtype == torch.int64 and got.shape == (M,)
    if V % 8 == 0:
        assert torch.equal(got, want)
    # independent of any argmax implementation: the chosen logit is the row maximum, and nothing before it equals it
    lf = logits.float()
    mx = lf.max(-1).values
    assert torch.equal(lf.gather(1, got[:, None])[:, 0], mx)
    first = (lf == mx[:, None]).float().argmax(-1)              # first index holding the maximum
    assert torch.equal(got, first)
    # a second call into the same workspace, with fewer rows
    if M > 2:
        assert torch.equal(ops.lm_head_argmax(x[: M // 2].contiguous(), w), got[: M // 2])


@pytest.mark.parametrize("tile", HEAD_TILES)
def test_every_head_tile_and_exact_ties(tile):
    """Forced tiles; rows of the table duplicated so that the maximum is attained twice, in one column tile and in
    two different ones: the smaller index must win; all-zero and NaN activations give token 0."""
    ops, Q = _ops()
    lib = Q.load()
    M, V, K = 70, 1000, 128
    x, w = _rand(M, V, K, seed=3)
    # row m of x points along table row (13 * m) % V, so that row is the winner by a wide margin ...
    win = (13 * torch.arange(M)) % V
    x = (w[win.cuda()].float() * 40).to(torch.bfloat16)
    # ... and gets an exact duplicate: in the same 128-column tile for even m, far away for odd m
    dup = torch.where(torch.arange(M) % 2 == 0, (win // 128) * 128 + (win + 5) % 104, (win + 517) % V)
    w = w.clone()
    w[dup.cuda()] = w[win.cuda()]
    x[3] = 0                                                   # every logit 0: first index
    x[4] = float("nan")                                        # no finite maximum
    want_tok = torch.minimum(win, dup).cuda()
    # (a later duplicate may have overwritten an earlier winner's row: recompute from the plain kernel)
    assert lib.ql_gemm_force_config(*tile, 1) == 0
    try:
        logits = ops.lm_head_logits(x, w)
        got = ops.lm_head_argmax(x, w)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
    ref = ops.argmax_rows(logits)
    assert torch.equal(got, ref)
    assert got[3].item() == 0 and got[4].item() == 0
    lf = logits.float()
    rows = [m for m in range(M) if m not in (3, 4)]
    tied = sum(int((lf[m] == lf[m].max()).sum().item() >= 2) for m in rows)
    assert tied >= len(rows) * 3 // 4                          # the construction does produce exact ties
    same = sum(int(got[m].item() == want_tok[m].item()) for m in rows)
    assert same >= len(rows) * 3 // 4


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_vocabulary_shards_combine_to_the_unsharded_tokens(shards):
    """What tensor-parallel ranks do, in one process: every shard's {winning logit, token} pair, combined by
    combine_shard_maxima, equals the sampler over the whole table - including maxima attained in two shards
    (the smaller token wins) and rows without a finite maximum."""
    from nanovllm.models.qwen3 import combine_shard_maxima
    ops, _ = _ops()
    M, V, K = 45, 1024 * shards, 128
    x, w = _rand(M, V, K, seed=shards)
    win = (37 * torch.arange(M)) % V
    x = (w[win.cuda()].float() * 40).to(torch.bfloat16)
    w = w.clone()
    w[((win + V // 2 + 3) % V).cuda()] = w[win.cuda()]          # the same row again, in another shard
    x[5] = float("nan")
    x[6] = 0
    per = V // shards
    vals, toks = [], []
    for r in range(shards):
        t, v = ops.lm_head_argmax(x, w[r * per:(r + 1) * per].contiguous(), return_max=True)
        vals.append(v)
        toks.append(t + r * per)
    got = combine_shard_maxima(torch.stack(vals), torch.stack(toks))
    want, vmax = ops.lm_head_argmax(x, w, return_max=True)
    assert torch.equal(got, want)
    lf = ops.lm_head_logits(x, w).float()
    assert torch.equal(vmax[7:], lf.max(-1).values[7:])
    assert vmax[5].item() == float("-inf") and want[5].item() == 0 and want[6].item() == 0
    assert sum(int((lf[m] == lf[m].max()).sum() >= 2) for m in range(7, M)) >= (M - 7) * 3 // 4


def test_rejections():
    ops, Q = _ops()
    lib = Q.load()
    x, w = _rand(4, 256, 64, seed=1)
    out = torch.empty(4, dtype=torch.int64, device="cuda")
    assert lib.ql_lmhead_argmax(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, 4, 256, 64, None, 0, None) == -3
    assert b"workspace" in lib.ql_last_error()
    assert lib.ql_lmhead_logits(x.data_ptr(), w.data_ptr(), out.data_ptr(), 4, 256, 48, None) == -2
    assert lib.ql_lmhead_workspace_bytes(256, 151936) == 256 * 1187 * 8


def test_engine_graphs_end_in_the_fused_sampler(tmp_path_factory):
    """Captured decode graphs produce tokens directly; the same engine with the sampler kept separate (the plain
    kernel's logits + ql_argmax_rows, eager) generates the same tokens."""
    import json
    from nanovllm import LLM, SamplingParams
    from nanovllm.models import qwen3
    d = tmp_path_factory.mktemp("tiny_head")
    cfg = dict(architectures=["Qwen3ForCausalLM"], model_type="qwen3", vocab_size=4000, hidden_size=256,
               intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
               max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=True,
               torch_dtype="bfloat16", attention_bias=False, hidden_act="silu", eos_token_id=1)
    (d / "config.json").write_text(json.dumps(cfg))
    g = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, 4000, (int(l),), generator=g).tolist() for l in torch.randint(3, 60, (9,), generator=g)]
    sp = SamplingParams(max_tokens=24, ignore_eos=True)
    fused = LLM(str(d), load_format="dummy", enforce_eager=False, max_model_len=512, max_num_seqs=16,
                quant_type="per_row_weight_only", linear_dtype=torch.int8)
    assert "tokens" in fused.model_runner.graph_vars and "logits" not in fused.model_runner.graph_vars
    out_f = [o["token_ids"] for o in fused.generate(prompts, sp)]
    orig = qwen3.ParallelLMHead.fused_sampler
    qwen3.ParallelLMHead.fused_sampler = lambda self, x: False
    try:
        plain = LLM(str(d), load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16,
                    quant_type="per_row_weight_only", linear_dtype=torch.int8)
        out_p = [o["token_ids"] for o in plain.generate(prompts, sp)]
    finally:
        qwen3.ParallelLMHead.fused_sampler = orig
    assert out_f == out_p
