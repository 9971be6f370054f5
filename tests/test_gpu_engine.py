"""GPU tests of the plumbing around the quantized linears: the fused norm / rope / activation /
decode-attention launches against plain PyTorch fp32 references of the same operators, and the
engine (scheduler + runner + hipGraph decode) end to end on a tiny random Qwen3."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TINY = dict(architectures=["Qwen3ForCausalLM"], hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
            num_key_value_heads=2, head_dim=128, intermediate_size=512, vocab_size=1024, rms_norm_eps=1e-6,
            rope_theta=1000000.0, max_position_embeddings=2048, tie_word_embeddings=True, attention_bias=False,
            hidden_act="silu", torch_dtype="bfloat16")


@pytest.fixture(scope="module")
def tiny_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("tiny_qwen3")
    with open(d / "config.json", "w") as f:
        json.dump(TINY, f)
    return str(d)


def _close(a, b, rtol=2e-2, atol=2e-2):
    a, b = a.float(), b.float()
    assert torch.isfinite(a).all()
    err = (a - b).abs()
    assert (err <= atol + rtol * b.abs()).all(), f"max err {err.max().item()} at scale {b.abs().max().item()}"


# ----------------------------------------------------------------------------- operators
def test_add_rmsnorm_matches_torch():
    from nanovllm import ops
    torch.manual_seed(0)
    for T, H in [(1, 128), (7, 1024), (300, 4096), (3, 8192)]:
        x = torch.randn(T, H, device="cuda", dtype=torch.bfloat16)
        r = torch.randn(T, H, device="cuda", dtype=torch.bfloat16)
        w = (1 + 0.1 * torch.randn(H, device="cuda")).to(torch.bfloat16)
        out, res = ops.add_rmsnorm(x, r, w, 1e-6)
        s = x.float() + r.float()
        ref_res = s.to(torch.bfloat16)
        ref = (s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16) * w
        assert torch.equal(res, ref_res)
        _close(out, ref, rtol=2 ** -6, atol=1e-6)
        out2, res2 = ops.add_rmsnorm(x, None, w, 1e-6)
        xf = x.float()
        ref2 = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16) * w
        _close(out2, ref2, rtol=2 ** -6, atol=1e-6)
        assert res2 is x


@pytest.mark.parametrize("dtype", [torch.int8, torch.float8_e4m3fn])
def test_add_rmsnorm_quant_equals_norm_then_per_row_quant(dtype):
    """The W8A8 prologue fusion: (q, scale) bit-identical to the two separate launches (and so to the oracle's
    per_row_quant of the normalised row, which the parity suite pins); all-zero rows hit the 1e-8 floor."""
    from nanovllm import ops
    from nanovllm.utils import quantization as NQ
    torch.manual_seed(6)
    for T, H in [(1, 128), (7, 1024), (130, 1024), (33, 4096), (3, 8192)]:
        x = torch.randn(T, H, device="cuda", dtype=torch.bfloat16) * 3
        r = torch.randn(T, H, device="cuda", dtype=torch.bfloat16)
        if T > 2:
            x[2] = 0
            r[2] = 0
        w = (1 + 0.1 * torch.randn(H, device="cuda")).to(torch.bfloat16)
        for res in (r, None):
            out, res_ref = ops.add_rmsnorm(x, res, w, 1e-6)
            q_ref, s_ref = NQ.per_row_quant(out, dtype)
            q, s, res_out = ops.add_rmsnorm_quant(x, res, w, 1e-6, dtype)
            assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(s, s_ref)
            if res is None:
                assert res_out is x
            else:
                assert torch.equal(res_out, res_ref)


def test_silu_mul_matches_torch():
    from nanovllm import ops
    torch.manual_seed(1)
    for T, I in [(1, 64), (5, 3072), (257, 9728)]:
        x = torch.randn(T, 2 * I, device="cuda", dtype=torch.bfloat16) * 3
        ref = F.silu(x[:, :I].float()) * x[:, I:].float()
        _close(ops.silu_mul(x), ref, rtol=2 ** -7, atol=1e-5)


def test_argmax_rows_matches_torch():
    from nanovllm import ops
    torch.manual_seed(5)
    for T, V in [(1, 8), (3, 1024), (130, 151936), (7, 4104)]:
        x = torch.randn(T, V, device="cuda", dtype=torch.bfloat16)
        assert torch.equal(ops.argmax_rows(x), x.float().argmax(-1))
    x = torch.zeros(4, 4096, device="cuda", dtype=torch.bfloat16)
    x[0, 77] = x[0, 3000] = 5.0                     # ties: the smallest index
    x[1, 4095] = 1.0
    x[2, :] = -3.0
    x[3, 10] = float("nan"); x[3, 20] = 0.5         # NaN never wins
    assert ops.argmax_rows(x).tolist() == [77, 4095, 0, 20]


def _rope_ref(x, pos, cos_sin):
    cs = cos_sin[pos]                                   # [T, 128]
    cos, sin = cs[:, None, :64], cs[:, None, 64:]
    x1, x2 = x.float()[..., :64], x.float()[..., 64:]
    return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), -1).to(x.dtype)


def test_qknorm_rope_store_matches_torch():
    from nanovllm import ops
    from nanovllm.models.qwen3 import build_cos_sin
    torch.manual_seed(2)
    T, Hq, Hkv, D = 37, 4, 2, 128
    cos_sin = build_cos_sin(D, 512, 1e6, "cuda")
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.bfloat16)
    pos = torch.randint(0, 512, (T,), device="cuda")
    qw = (1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16)
    kw = (1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16)
    kc = torch.zeros(4, 256, Hkv, D, device="cuda", dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    slots = torch.randperm(1024, device="cuda")[:T].to(torch.int32)
    slots[3] = -1
    q, k, v = ops.qknorm_rope_store(qkv, pos, cos_sin, qw, kw, 1e-6, Hq, Hkv, D, kc, vc, slots)

    def norm(x, w):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16) * w
    q0, k0, v0 = qkv.split([Hq * D, Hkv * D, Hkv * D], -1)
    q_ref = _rope_ref(norm(q0.view(T, Hq, D), qw), pos, cos_sin)
    k_ref = _rope_ref(norm(k0.view(T, Hkv, D), kw), pos, cos_sin)
    _close(q, q_ref, rtol=2 ** -6, atol=2e-3)
    _close(k, k_ref, rtol=2 ** -6, atol=2e-3)
    assert torch.equal(v, v0.view(T, Hkv, D))
    live = slots >= 0
    assert torch.equal(kc.view(-1, Hkv, D)[slots[live].long()], k[live])
    assert torch.equal(vc.view(-1, Hkv, D)[slots[live].long()], v[live])
    assert kc.view(-1, Hkv, D).abs().sum(dim=(1, 2)).ne(0).sum().item() == int(live.sum())


@pytest.mark.parametrize("G", [1, 2, 4, 5, 8])      # (5, 8: no kernel of their own, dealt to workgroups of 1 / 4 heads)
def test_paged_attn_decode_matches_torch(G):
    from nanovllm import ops
    torch.manual_seed(3)
    B, Hkv, D, bs = 9, 2, 128, 256
    Hq = Hkv * G
    lens = [1, 5, 16, 17, 255, 256, 257, 700, 0]
    nblk = sum((l + bs - 1) // bs for l in lens) + 1
    kc = torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16)
    tables = torch.zeros(B, 4, dtype=torch.int32, device="cuda")
    nxt = 1
    for b, l in enumerate(lens):
        for j in range((l + bs - 1) // bs):
            tables[b, j] = nxt
            nxt += 1
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.bfloat16)
    ctx = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5)
    for b, l in enumerate(lens):
        if l == 0:
            assert out[b].abs().max().item() == 0
            continue
        idx = torch.cat([tables[b, j].long() * bs + torch.arange(bs, device="cuda") for j in range((l + bs - 1) // bs)])[:l]
        k = kc.view(-1, Hkv, D)[idx].float().repeat_interleave(G, dim=1)   # [l, Hq, D]
        v = vc.view(-1, Hkv, D)[idx].float().repeat_interleave(G, dim=1)
        s = torch.einsum("hd,lhd->hl", q[b].float(), k) * D ** -0.5
        ref = torch.einsum("hl,lhd->hd", s.softmax(-1), v)
        _close(out[b], ref, rtol=2 ** -6, atol=2e-2)
    # the variant that also emits partial row maxima (for a W8A8 o_proj that quantizes its input in the GEMM): same
    # outputs bit for bit; every (sequence, kv head) entry is exactly the |max| of that head group's outputs, for
    # single-chunk rows (attention launch), multi-chunk rows (merge launch) and padded rows
    # the variant whose merge launch also quantizes the rows for a W8A8 o_proj: ql_quant_per_row's bytes and scales
    from nanovllm.utils import quantization as NQ
    for qd in (torch.int8, torch.float8_e4m3fn):
        out3, qx, sx = ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5, quant_dtype=qd)
        assert torch.equal(out3, out)
        qr, sr = NQ.per_row_quant(out.view(B, Hq * D), qd)
        assert torch.equal(sx, sr) and torch.equal(qx.view(torch.uint8), qr.view(torch.uint8))
    if G not in (1, 2, 4):            # per-kv-head row maxima exist for the ratios with a kernel of their own
        with pytest.raises(Exception, match="row maxima"):
            ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5, return_rowmax=True)
        return
    rm0 = torch.full((B, Hkv), -1.0, device="cuda")
    out2, rm = ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5, return_rowmax=True)
    assert torch.equal(out2, out) and rm.shape == rm0.shape
    assert torch.equal(rm, out.view(B, Hkv, G * D).float().abs().amax(-1))


@pytest.mark.parametrize("G", [1, 2, 4])
def test_paged_attn_decode_lengths_around_the_kernel_steps(G):
    """Context lengths on both sides of every boundary of the decode kernel's token walk: the 32 tokens a workgroup's
    slots take per step, the 64 / 128 tokens of one loop iteration (two / four tokens per slot in flight), the 256-token
    cache block and the 512-token chunk; block tables in scattered order."""
    from nanovllm import ops
    torch.manual_seed(11)
    Hkv, D, bs = 2, 128, 256
    Hq = Hkv * G
    lens = [2, 7, 8, 9, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 254, 258, 383, 511, 512, 513, 640, 1023, 1024, 1025, 1500]
    B = len(lens)
    nblk = sum((l + bs - 1) // bs for l in lens) + 1
    kc = torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16)
    order = (torch.randperm(nblk - 1) + 1).tolist()
    tables = torch.zeros(B, 8, dtype=torch.int32, device="cuda")
    for b, l in enumerate(lens):
        for j in range((l + bs - 1) // bs):
            tables[b, j] = order.pop()
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.bfloat16)
    ctx = torch.tensor(lens, dtype=torch.int32, device="cuda")
    ops.reserve_attn_workspace(B, Hq, Hkv, bs, 8, "cuda")
    out = ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5)
    for b, l in enumerate(lens):
        idx = torch.cat([tables[b, j].long() * bs + torch.arange(bs, device="cuda") for j in range((l + bs - 1) // bs)])[:l]
        k = kc.view(-1, Hkv, D)[idx].float().repeat_interleave(G, dim=1)
        v = vc.view(-1, Hkv, D)[idx].float().repeat_interleave(G, dim=1)
        s = torch.einsum("hd,lhd->hl", q[b].float(), k) * D ** -0.5
        ref = torch.einsum("hl,lhd->hd", s.softmax(-1), v)
        _close(out[b], ref, rtol=2 ** -6, atol=2e-2)


@pytest.mark.parametrize("bs", [16, 128, 384, 512, 1024])
@pytest.mark.parametrize("G", [2, 4])
def test_paged_attn_decode_long_contexts_and_block_sizes(G, bs):
    """Sequences of up to twelve 512-token chunks (the merge launch beyond a handful of chunk states) over caches whose
    block is smaller than, equal to, not a divisor of, and larger than the chunk (a chunk then spans many, two or
    one block); the quantizing merge stays bit-identical to per_row_quant of the plain output."""
    from nanovllm import ops
    from nanovllm.utils import quantization as NQ
    torch.manual_seed(17 + bs)
    Hkv, D = 2, 128
    Hq = Hkv * G
    lens = [4095, 4096, 4097, 6100, 2049, 513, 1]
    B = len(lens)
    width = (max(lens) + bs - 1) // bs
    nblk = sum((l + bs - 1) // bs for l in lens) + 1
    kc = torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16)
    order = (torch.randperm(nblk - 1) + 1).tolist()
    tables = torch.zeros(B, width, dtype=torch.int32)
    for b, l in enumerate(lens):
        for j in range((l + bs - 1) // bs):
            tables[b, j] = order.pop()
    tables = tables.cuda()
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.bfloat16)
    ctx = torch.tensor(lens, dtype=torch.int32, device="cuda")
    ops.reserve_attn_workspace(B, Hq, Hkv, bs, width, "cuda")
    out = ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5)
    for b, l in enumerate(lens):
        nb = (l + bs - 1) // bs
        idx = (tables[b, :nb].long()[:, None] * bs + torch.arange(bs, device="cuda")[None, :]).reshape(-1)[:l]
        k = kc.view(-1, Hkv, D)[idx].float().repeat_interleave(G, dim=1)
        v = vc.view(-1, Hkv, D)[idx].float().repeat_interleave(G, dim=1)
        s = torch.einsum("hd,lhd->hl", q[b].float(), k) * D ** -0.5
        ref = torch.einsum("hl,lhd->hd", s.softmax(-1), v)
        _close(out[b], ref, rtol=2 ** -6, atol=2e-2)
    out2, rm = ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5, return_rowmax=True)
    assert torch.equal(out2, out)
    assert torch.equal(rm, out.view(B, Hkv, G * D).float().abs().amax(-1))
    for qd in (torch.int8, torch.float8_e4m3fn):
        out3, qx, sx = ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5, quant_dtype=qd)
        assert torch.equal(out3, out)
        qr, sr = NQ.per_row_quant(out.view(B, Hq * D), qd)
        assert torch.equal(sx, sr) and torch.equal(qx.view(torch.uint8), qr.view(torch.uint8))


def test_paged_attn_decode_bench_size_properties():
    """At the benchmark's size (256 sequences of 100-2047 cached tokens, Qwen3-0.6B heads) the properties that need no
    reference: the same call twice gives the same bits, a sequence run alone gives the bits it has inside the batch, and
    moving every sequence's tokens to other physical cache blocks (tables rewritten accordingly) changes nothing."""
    from nanovllm import ops
    g = torch.Generator().manual_seed(23)
    B, Hq, Hkv, D, bs, width = 256, 16, 8, 128, 256, 8
    lens = torch.randint(100, 2048, (B,), generator=g)
    nb = (lens + bs - 1) // bs
    nblk = int(nb.sum()) + 1
    kc = torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16)
    tables = torch.zeros(B, width, dtype=torch.int32)
    nxt = 1
    for b in range(B):
        for j in range(int(nb[b])):
            tables[b, j] = nxt
            nxt += 1
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.bfloat16)
    ctx = lens.to(torch.int32).cuda()
    ops.reserve_attn_workspace(B, Hq, Hkv, bs, width, "cuda")
    out = ops.paged_attn_decode(q, kc, vc, tables.cuda(), ctx, D ** -0.5)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(ops.paged_attn_decode(q, kc, vc, tables.cuda(), ctx, D ** -0.5), out)
    for b in (0, 17, 255, int(lens.argmax()), int(lens.argmin())):
        alone = ops.paged_attn_decode(q[b:b + 1].contiguous(), kc, vc, tables[b:b + 1].cuda(), ctx[b:b + 1].contiguous(),
                                      D ** -0.5)
        assert torch.equal(alone[0], out[b])
    perm = torch.cat([torch.zeros(1, dtype=torch.long), torch.randperm(nblk - 1, generator=g) + 1])   # old block -> new
    kc2, vc2 = torch.empty_like(kc), torch.empty_like(vc)
    kc2[perm.cuda()] = kc
    vc2[perm.cuda()] = vc
    tables2 = perm[tables.long()].to(torch.int32)
    tables2[tables == 0] = 0
    assert torch.equal(ops.paged_attn_decode(q, kc2, vc2, tables2.cuda(), ctx, D ** -0.5), out)


@pytest.mark.parametrize("G", [1, 2, 4, 5, 8])
def test_attn_prefill_varlen_matches_torch(G):
    """Packed causal self-attention against an fp32 softmax(QK^T)V per sequence; lengths straddle the
    32-row query blocks and the 64-key tiles.  Tolerance: bf16 output rounding plus bf16 probabilities."""
    from nanovllm import ops
    torch.manual_seed(4)
    Hkv, D = 2, 128
    Hq = Hkv * G
    lens = [1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 200, 517, 1024]
    cu = [0]
    for l in lens:
        cu.append(cu[-1] + l)
    T = cu[-1]
    q = torch.randn(T, Hq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(T, Hkv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(T, Hkv, D, device="cuda", dtype=torch.bfloat16)
    out = ops.attn_prefill(q, k, v, torch.tensor(cu, dtype=torch.int32, device="cuda"), max(lens), D ** -0.5)
    assert out.shape == q.shape and torch.isfinite(out.float()).all()
    for i, l in enumerate(lens):
        a, b = cu[i], cu[i + 1]
        kk = k[a:b].float().repeat_interleave(G, dim=1)
        vv = v[a:b].float().repeat_interleave(G, dim=1)
        s = torch.einsum("qhd,khd->hqk", q[a:b].float(), kk) * D ** -0.5
        s = s.masked_fill(torch.triu(torch.ones(l, l, dtype=torch.bool, device="cuda"), 1), float("-inf"))
        ref = torch.einsum("hqk,khd->qhd", s.softmax(-1), vv)
        _close(out[a:b], ref, rtol=2 ** -6, atol=1.5e-2)
    # sharper scores (large logits): the online rescale path must hold up
    out2 = ops.attn_prefill(q * 4, k * 4, v, torch.tensor(cu, dtype=torch.int32, device="cuda"), max(lens), D ** -0.5)
    a, b = cu[-2], cu[-1]
    kk = (k[a:b] * 4).float().repeat_interleave(G, dim=1)
    s = torch.einsum("qhd,khd->hqk", (q[a:b] * 4).float(), kk) * D ** -0.5
    s = s.masked_fill(torch.triu(torch.ones(b - a, b - a, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    ref = torch.einsum("hqk,khd->qhd", s.softmax(-1), v[a:b].float().repeat_interleave(G, dim=1))
    _close(out2[a:b], ref, rtol=2 ** -6, atol=2e-2)


# ----------------------------------------------------------------------------- engine
def _prompts(n=4):
    g = torch.Generator().manual_seed(0)
    return [torch.randint(0, 1000, (int(l),), generator=g).tolist() for l in torch.randint(3, 40, (n,), generator=g)]


def _ref_forward(model, ids):
    """Plain PyTorch forward of the same (unquantized) weights for one sequence: fp32 math."""
    m = model.model
    T = len(ids)
    pos = torch.arange(T, device="cuda")
    x = F.embedding(torch.tensor(ids, device="cuda"), m.embed_tokens.weight).float()

    def rms(v, w, eps=1e-6):
        return v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + eps) * w.float()
    for l in m.layers:
        a = l.self_attn
        h = rms(x, l.input_layernorm.weight)
        qkv = h @ a.qkv_proj.weight.float().T
        q, k, v = qkv.split([a.num_heads * 128, a.num_kv_heads * 128, a.num_kv_heads * 128], -1)
        q = rms(q.view(T, -1, 128), a.q_norm.weight)
        k = rms(k.view(T, -1, 128), a.k_norm.weight)
        q, k = _rope_ref(q, pos, m.cos_sin).float(), _rope_ref(k, pos, m.cos_sin).float()
        v = v.view(T, -1, 128)
        o = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None],
                                           is_causal=True, enable_gqa=True)[0].transpose(0, 1).reshape(T, -1)
        x = x + o @ a.o_proj.weight.float().T
        h = rms(x, l.post_attention_layernorm.weight)
        gu = h @ l.mlp.gate_up_proj.weight.float().T
        I = gu.shape[-1] // 2
        x = x + (F.silu(gu[:, :I]) * gu[:, I:]) @ l.mlp.down_proj.weight.float().T
    return rms(x, m.norm.weight) @ model.lm_head.weight.float().T


def test_engine_bf16_eager_matches_torch_reference_and_graph(tiny_dir):
    from nanovllm import LLM, SamplingParams
    sp = SamplingParams(temperature=0.6, max_tokens=12, ignore_eos=True)
    prompts = _prompts()
    eager = LLM(tiny_dir, load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16)
    out_e = eager.generate(prompts, sp)
    assert [len(o["token_ids"]) for o in out_e] == [12] * 4
    # greedy continuation by the fp32 PyTorch forward of the same weights: same first tokens
    model = eager.model_runner.model
    agree = 0
    for p, o in zip(prompts, out_e):
        logits = _ref_forward(model, p)[-1]
        top2 = logits.topk(2).values
        if o["token_ids"][0] == int(logits.argmax()):
            agree += 1
        else:   # allow a flip only when the fp32 reference itself is nearly tied
            assert (top2[0] - top2[1]).item() < 0.05 * logits.abs().max().item()
    assert agree >= 3
    # decode through captured hipGraphs gives exactly the eager tokens (same kernels, same order)
    graph = LLM(tiny_dir, load_format="dummy", enforce_eager=False, max_model_len=512, max_num_seqs=16)
    out_g = graph.generate(prompts, sp)
    assert [o["token_ids"] for o in out_g] == [o["token_ids"] for o in out_e]


@pytest.mark.parametrize("kw", [
    dict(quant_type="per_row_weight_only", linear_dtype=torch.int8),
    dict(quant_type="per_row", linear_dtype=torch.int8),
    dict(quant_type="per_row", linear_dtype=torch.float8_e4m3fn),
    dict(quant_type="per_group", linear_dtype=torch.int8, group_size=128),
    dict(quant_type="per_group", linear_dtype=torch.int8, group_size=16),   # below the MFMA step: plain GEMMs, no fused forms
    dict(quant_type="per_tensor", linear_dtype=torch.int8),
    dict(linear_dtype=torch.int8),   # no quant_type: non-bf16 linear dtype means per-row (model_runner.py:55-57)
])
def test_engine_quantized_variants(tiny_dir, kw):
    from nanovllm import LLM, SamplingParams
    from nanovllm.models.qwen3 import AttnContext, set_context
    from nanovllm.utils import quantization as NQ
    llm = LLM(tiny_dir, load_format="dummy", enforce_eager=False, max_model_len=512, max_num_seqs=16, **kw)
    layer = llm.model_runner.model.model.layers[0]
    want = {"per_group": NQ.QGroupLinear, "per_tensor": NQ.QTensorLinear}.get(kw.get("quant_type"), NQ.QLinear)
    for m in (layer.self_attn.qkv_proj, layer.self_attn.o_proj, layer.mlp.gate_up_proj, layer.mlp.down_proj):
        assert isinstance(m, want)
    out = llm.generate(_prompts(), SamplingParams(max_tokens=8, ignore_eos=True))
    assert all(len(o["token_ids"]) == 8 for o in out)
    # hidden states stay close to the unquantized model with the same (seeded) weights
    ref = LLM(tiny_dir, load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16)
    ids = torch.tensor(_prompts(1)[0], device="cuda")
    pos = torch.arange(ids.numel(), device="cuda")
    with torch.inference_mode():
        hs = []
        for e in (ref, llm):
            set_context(AttnContext(is_prefill=True, cu_seqlens=[0, ids.numel()], slot_mapping=None))
            hs.append(e.model_runner.model(ids, pos).float())
    rel = (hs[0] - hs[1]).abs().max() / hs[0].abs().max()
    assert rel < 0.1, float(rel)


def test_attn_prefill_bench_size_properties():
    """One prefill batch of the benchmark's size (16,384 packed tokens, Qwen3-0.6B heads): the same call twice gives
    the same bits, a sequence packed alone gives the bits it has inside the batch, and - causality - the first n rows
    of a sequence do not depend on the tokens after them (the sequence cut to n tokens gives the same bits)."""
    from nanovllm import ops
    g = torch.Generator().manual_seed(29)
    Hq, Hkv, D = 16, 8, 128
    lens = []
    while sum(lens) < 16384 - 1024:
        lens.append(int(torch.randint(100, 1025, (1,), generator=g)))
    lens.append(16384 - sum(lens))
    T = sum(lens)
    cu = [0]
    for l in lens:
        cu.append(cu[-1] + l)
    q = torch.randn(T, Hq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(T, Hkv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(T, Hkv, D, device="cuda", dtype=torch.bfloat16)
    cu_t = torch.tensor(cu, dtype=torch.int32, device="cuda")
    out = ops.attn_prefill(q, k, v, cu_t, max(lens), D ** -0.5)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(ops.attn_prefill(q, k, v, cu_t, max(lens), D ** -0.5), out)
    for s in (0, len(lens) // 2, len(lens) - 1):
        a, b = cu[s], cu[s + 1]
        one = torch.tensor([0, b - a], dtype=torch.int32, device="cuda")
        alone = ops.attn_prefill(q[a:b].contiguous(), k[a:b].contiguous(), v[a:b].contiguous(), one, b - a, D ** -0.5)
        assert torch.equal(alone, out[a:b])
        for n in (1, 33, (b - a) // 2, b - a - 1):
            cut = torch.tensor([0, n], dtype=torch.int32, device="cuda")
            head = ops.attn_prefill(q[a:a + n].contiguous(), k[a:a + n].contiguous(), v[a:a + n].contiguous(), cut, n, D ** -0.5)
            assert torch.equal(head, out[a:a + n])


@pytest.mark.parametrize("dtype", [torch.int8, torch.float8_e4m3fn])
def test_w8a8_engine_without_activation_quant_launches(tiny_dir, dtype):
    """W8A8 engine with NANOVLLM_DYNQ: o_proj and down_proj quantize their input inside the GEMM from the partial
    row maxima that decode attention / the gate-up SiLU epilogue emit.  Same tokens as the default engine, which
    keeps the separate ql_quant_per_row launches (the kernels are bit-identical), through eager steps and graphs."""
    from nanovllm import LLM, SamplingParams
    from nanovllm.utils import quantization as NQ
    sp = SamplingParams(max_tokens=20, ignore_eos=True)
    kw = dict(load_format="dummy", max_model_len=512, max_num_seqs=16, quant_type="per_row", linear_dtype=dtype)
    was = NQ.FUSE_ACT_QUANT                     # (an opt-in: measured slower than the separate launch)
    calls = []
    orig = NQ.w8a8_dynq_matmul
    NQ.w8a8_dynq_matmul = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    NQ.FUSE_ACT_QUANT = True
    try:
        out_f = [o["token_ids"] for o in LLM(tiny_dir, enforce_eager=False, **kw).generate(_prompts(6), sp)]
        out_fe = [o["token_ids"] for o in LLM(tiny_dir, enforce_eager=True, **kw).generate(_prompts(6), sp)]
        assert calls, "the fused path did not run"
        NQ.FUSE_ACT_QUANT = False
        n = len(calls)
        out_p = [o["token_ids"] for o in LLM(tiny_dir, enforce_eager=True, **kw).generate(_prompts(6), sp)]
        assert len(calls) == n, "the switch did not turn the fused path off"
    finally:
        NQ.w8a8_dynq_matmul = orig
        NQ.FUSE_ACT_QUANT = was
    assert out_f == out_p and out_fe == out_p


@pytest.mark.parametrize("dtype", [torch.int8, torch.float8_e4m3fn])
def test_w8a8_o_proj_input_quantized_by_the_attention_merge_launch(tiny_dir, dtype):
    """Default W8A8 decode: the attention merge launch hands o_proj (q, scale) directly.  Same tokens as with the
    separate ql_quant_per_row launch."""
    from nanovllm import LLM, SamplingParams
    from nanovllm.models import qwen3
    sp = SamplingParams(max_tokens=20, ignore_eos=True)
    kw = dict(load_format="dummy", max_model_len=512, max_num_seqs=16, quant_type="per_row", linear_dtype=dtype)
    assert qwen3.FUSE_ATTN_QUANT
    out_f = [o["token_ids"] for o in LLM(tiny_dir, enforce_eager=False, **kw).generate(_prompts(6), sp)]
    qwen3.FUSE_ATTN_QUANT = False
    try:
        out_p = [o["token_ids"] for o in LLM(tiny_dir, enforce_eager=True, **kw).generate(_prompts(6), sp)]
    finally:
        qwen3.FUSE_ATTN_QUANT = True
    assert out_f == out_p


def test_quantized_lm_head_opt_in(tiny_dir):
    """quantize_lm_head=True: the vocabulary projection runs through the per-row W8A16 kernel; logits stay
    within int8 weight-quantization noise of the bf16 head (max error < 2 % of the logit range, the bar the
    reference sets for its per-row linears is 8 %, quantization.py:549-553) and graph-mode decoding works."""
    from nanovllm import LLM, SamplingParams
    from nanovllm.utils import quantization as NQ
    kw = dict(load_format="dummy", max_model_len=512, max_num_seqs=16, quant_type="per_row_weight_only",
              linear_dtype=torch.int8)
    llm = LLM(tiny_dir, enforce_eager=False, quantize_lm_head=True, **kw)
    head = llm.model_runner.model.lm_head
    assert isinstance(head.quantized, NQ.QLinear) and head.quantized.weight_only
    x = torch.randn(37, head.weight.shape[1], device="cuda", dtype=torch.bfloat16)
    with torch.inference_mode():
        got = head(x).float()
        ref = torch.nn.functional.linear(x, head.weight).float()
    assert got.shape == ref.shape
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 0.02
    out = llm.generate(_prompts(), SamplingParams(max_tokens=8, ignore_eos=True))
    assert all(len(o["token_ids"]) == 8 for o in out)
    plain = LLM(tiny_dir, enforce_eager=False, **kw)
    assert plain.model_runner.model.lm_head.quantized is None


def test_weight_fake_quant_dispatch(tiny_dir):
    """weight_quant_fn takes precedence and keeps bf16 linears (model_runner.py:37-39)."""
    from nanovllm import LLM, SamplingParams
    from nanovllm.layers.linear import LinearBase
    from nanovllm.utils import quantization as NQ
    llm = LLM(tiny_dir, load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16,
              weight_quant_fn=lambda w: NQ.fake_per_row_quant(w, torch.int8), quant_type="per_row",
              linear_dtype=torch.int8)
    assert isinstance(llm.model_runner.model.model.layers[0].mlp.down_proj, LinearBase)
    out = llm.generate(_prompts(2), SamplingParams(max_tokens=4, ignore_eos=True))
    assert all(len(o["token_ids"]) == 4 for o in out)


def test_scheduler_continuous_batching_and_finish(tiny_dir):
    """More sequences than max_num_seqs, different lengths: all finish with exactly max_tokens."""
    from nanovllm import LLM, SamplingParams
    llm = LLM(tiny_dir, load_format="dummy", enforce_eager=False, max_model_len=512, max_num_seqs=4,
              quant_type="per_row_weight_only", linear_dtype=torch.int8)
    prompts = _prompts(9)
    sps = [SamplingParams(max_tokens=3 + i, ignore_eos=True) for i in range(9)]
    out = llm.generate(prompts, sps)
    assert [len(o["token_ids"]) for o in out] == [3 + i for i in range(9)]


def test_pipelined_decode_matches_step_by_step(tiny_dir):
    """Launching step i+1 before step i's tokens reach the host (input ids gathered on the device) gives the
    outputs of the step-by-step loop: different lengths (rows leave the batch at different steps), and with a
    live EOS, which ends its sequence one launch late and must drop the token launched beyond it."""
    from nanovllm import LLM, SamplingParams
    llm = LLM(tiny_dir, load_format="dummy", enforce_eager=False, max_model_len=512, max_num_seqs=16,
              quant_type="per_row_weight_only", linear_dtype=torch.int8)
    prompts = _prompts(12)
    sps = [SamplingParams(max_tokens=5 + 3 * i, ignore_eos=True) for i in range(12)]
    launches = []
    orig = llm.model_runner.launch
    llm.model_runner.launch = lambda seqs, is_prefill, prev=None: (launches.append(prev is not None), orig(seqs, is_prefill, prev=prev))[1]
    llm.pipeline_decode = True
    piped = [o["token_ids"] for o in llm.generate(prompts, sps)]
    assert sum(launches) >= 30, "decode steps were not pipelined"
    llm.pipeline_decode = False
    launches.clear()
    plain = [o["token_ids"] for o in llm.generate(prompts, sps)]
    assert not any(launches)
    assert piped == plain and [len(t) for t in plain] == [5 + 3 * i for i in range(12)]
    # a token that some sequence produces in the middle of its output becomes the EOS
    eos = next(t[len(t) // 2] for t in plain if len(t) >= 8)
    llm.scheduler.eos = eos
    sps = [SamplingParams(max_tokens=5 + 3 * i, ignore_eos=False) for i in range(12)]
    outs = []
    for mode in (True, False):
        llm.pipeline_decode = mode
        outs.append([o["token_ids"] for o in llm.generate(prompts, sps)])
    assert outs[0] == outs[1]
    for full, cut in zip(plain, outs[0]):
        want = full[:full.index(eos) + 1] if eos in full else full
        assert cut == want
    assert any(len(c) < len(f) for f, c in zip(plain, outs[0]))
    assert len(llm.scheduler.block_manager.free) == llm.scheduler.block_manager.num_blocks
    llm.model_runner.launch = orig


def test_more_running_sequences_than_graph_buckets(tiny_dir):
    """max_num_seqs above the 512-row staging / graph limit: such batches decode eagerly and step by step (the
    pipelined loop needs the staged inputs), with the outputs of the small-batch route."""
    from nanovllm import LLM, SamplingParams
    llm = LLM(tiny_dir, load_format="dummy", enforce_eager=False, max_model_len=256, max_num_seqs=600,
              max_num_batched_tokens=4096, quant_type="per_row_weight_only", linear_dtype=torch.int8)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, 1000, (3,), generator=g).tolist() for _ in range(530)]
    sizes = []
    orig = llm.model_runner.launch
    llm.model_runner.launch = lambda seqs, is_prefill, prev=None: (sizes.append((len(seqs), is_prefill, prev is not None)),
                                                                   orig(seqs, is_prefill, prev=prev))[1]
    out = llm.generate(prompts, SamplingParams(max_tokens=4, ignore_eos=True))
    llm.model_runner.launch = orig
    assert [len(o["token_ids"]) for o in out] == [4] * 530
    assert any(n > 512 and not pf for n, pf, _ in sizes), "no decode step above the staging limit ran"
    assert not any(piped for n, pf, piped in sizes if n > 512)
    small = LLM(tiny_dir, load_format="dummy", enforce_eager=False, max_model_len=256, max_num_seqs=64,
                quant_type="per_row_weight_only", linear_dtype=torch.int8)
    ref = small.generate(prompts[:40], SamplingParams(max_tokens=4, ignore_eos=True))
    assert [o["token_ids"] for o in ref] == [o["token_ids"] for o in out[:40]]


def test_captured_graphs_survive_a_workspace_that_is_outgrown(tmp_path):
    """The decode hipGraphs hold the split-K workspace by address.  When another caller later needs a larger one,
    the old buffer must stay alive and untouched: graph decode keeps producing the eager tokens.  (The MLP is
    3072 wide so that down_proj really splits K three ways and uses the workspace.)"""
    from nanovllm import LLM, SamplingParams, _qlinear as Q
    tiny_dir = str(tmp_path)
    with open(tmp_path / "config.json", "w") as f:
        json.dump(dict(TINY, intermediate_size=3072), f)
    assert Q.load().ql_gemm_workspace_bytes(4, 256, 3072, 0) > 0
    sp = SamplingParams(max_tokens=10, ignore_eos=True)
    kw = dict(load_format="dummy", max_model_len=512, max_num_seqs=16, quant_type="per_row_weight_only",
              linear_dtype=torch.int8)
    llm = LLM(tiny_dir, enforce_eager=False, **kw)
    before = [o["token_ids"] for o in llm.generate(_prompts(), sp)]
    dev = llm.model_runner.device
    old = Q.workspace.buf[Q.workspace._key(dev)]
    Q.workspace.reserve(dev, old.numel() + (8 << 20))             # somebody needs more: a new buffer
    assert Q.workspace.buf[Q.workspace._key(dev)] is not old
    assert any(old is b for b in Q.workspace.pinned[Q.workspace._key(dev)])
    junk = [torch.full((old.numel(),), 0xFF, dtype=torch.uint8, device=dev) for _ in range(4)]   # would land in a freed buffer
    torch.cuda.synchronize()
    assert int(old[: 64 * 1024].max()) == 0                       # the ticket counters of the pinned buffer are intact
    after = [o["token_ids"] for o in llm.generate(_prompts(), sp)]
    del junk
    assert after == before
    eager = LLM(tiny_dir, enforce_eager=True, **kw)
    assert [o["token_ids"] for o in eager.generate(_prompts(), sp)] == before


def test_rccl_allreduce_on_side_stream_is_graph_capturable():
    """The row-parallel reduction (RCCL all-reduce enqueued on a side HIP stream with event fork/join)
    must be capturable into the decode hipGraph.  One GPU here, so the group has a single rank; the
    capture/replay mechanics (stream fork, collective node, join) are the same for any group size."""
    import torch.distributed as dist
    from nanovllm.layers.linear import tp_all_reduce
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29713")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        y = torch.randn(64, 1024, device="cuda", dtype=torch.bfloat16)
        ref = y.clone()
        tp_all_reduce(y)                      # eager once: communicator set-up happens outside capture
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
        g = torch.cuda.CUDAGraph()
        out = torch.empty_like(y)
        with torch.cuda.graph(g):
            z = y * 2
            tp_all_reduce(z)
            out.copy_(z + 1)
        y.copy_(ref * 3)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, (ref * 3) * 2 + 1)
    finally:
        dist.destroy_process_group()


def test_row_parallel_slab_pipeline_on_rccl():
    """A prefill-sized row-parallel output is produced in row slabs, each all-reduced by RCCL on the side stream
    while the next slab's GEMM runs.  One GPU, so the group has one rank (the sum is the identity): the result
    must be exactly the per-slab GEMMs, eagerly and replayed from a hipGraph, for W8A16, W8A8 and per-group."""
    import torch.distributed as dist
    from nanovllm.layers import linear as L
    from nanovllm.utils import quantization as NQ
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29714")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        M, N, K = 4096 + 2048 + 100, 1024, 1536
        g = torch.Generator(device="cuda").manual_seed(2)
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        bounds = L.tp_slab_bounds(M)
        assert bounds == [(0, 2048), (2048, 4096), (4096, 6144), (6144, M)]
        mods = [NQ.QLinear(torch.int8, w, None, weight_only=True), NQ.QLinear(torch.float8_e4m3fn, w, None),
                NQ.QGroupLinear(torch.int8, w, None, 128)]
        for m in mods:
            m.tile()
            ref = torch.cat([m(x[a:b].contiguous()) for a, b in bounds])      # plain module: no TP role
            m._init_tp(1, 0, 2)                                                  # row-parallel shard of a 2-way group
            y = m(x)
            torch.cuda.synchronize()
            assert torch.equal(y, ref), type(m).__name__
            gr = torch.cuda.CUDAGraph()
            out = torch.empty_like(ref)
            with torch.cuda.graph(gr):
                out.copy_(m(x))
            out.zero_()
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref), type(m).__name__
    finally:
        dist.destroy_process_group()


def test_tensor_parallel_decode_step_captured_on_rccl(tiny_dir):
    """What a multi-GPU run of the benchmark does in every decode step, on the collective library itself: a model whose
    modules play their tensor-parallel roles (vocabulary-parallel embedding with its all-reduce, row-parallel o_proj /
    down_proj with the side-stream all-reduce, the sampler's exchange of shard maxima through all_gather_into_tensor)
    is captured into the decode hipGraphs in "thread_local" capture mode on an RCCL group.  One GPU, so the group has
    one rank and every collective is the identity: captured tokens == eager tokens == tokens of the role-free model."""
    import torch.distributed as dist
    from nanovllm import LLM, SamplingParams
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29716")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        kw = dict(load_format="dummy", max_model_len=512, max_num_seqs=16, quant_type="per_row_weight_only",
                  linear_dtype=torch.int8)
        sp = SamplingParams(max_tokens=12, ignore_eos=True)
        llm = LLM(tiny_dir, enforce_eager=True, **kw)
        before = [o["token_ids"] for o in llm.generate(_prompts(), sp)]
        runner = llm.model_runner
        model = runner.model
        model.model.embed_tokens.tp_size = 2                      # (the whole table is this rank's "shard")
        for layer in model.model.layers:
            layer.self_attn.o_proj._init_tp(1, 0, 2)              # row-parallel shard of a 2-way group
            layer.mlp.down_proj._init_tp(1, 0, 2)
        model.lm_head.force_exchange = True
        calls = {"all_reduce": 0, "all_gather": 0}
        real_ar, real_ag = dist.all_reduce, dist.all_gather_into_tensor

        def counting_ar(*a, **k):
            calls["all_reduce"] += 1
            return real_ar(*a, **k)

        def counting_ag(*a, **k):
            calls["all_gather"] += 1
            return real_ag(*a, **k)
        dist.all_reduce, dist.all_gather_into_tensor = counting_ar, counting_ag
        try:
            assert [o["token_ids"] for o in llm.generate(_prompts(), sp)] == before          # eager, with the collectives
            n_layers = len(model.model.layers)
            assert calls["all_reduce"] >= 2 * n_layers + 1 and calls["all_gather"] >= 2
            runner.world_size = 2                                     # -> capture_error_mode="thread_local"
            calls.update(all_reduce=0, all_gather=0)
            runner.capture_graphs()
            # every captured step issued its collectives under capture: embedding + 2 per layer, and the sampler's two
            assert calls["all_reduce"] >= len(runner.graph_bs) * (2 * n_layers + 1)
            assert calls["all_gather"] >= 2 * len(runner.graph_bs)
        finally:
            dist.all_reduce, dist.all_gather_into_tensor = real_ar, real_ag
        runner.enforce_eager = False
        assert [o["token_ids"] for o in llm.generate(_prompts(), sp)] == before              # replayed graphs
        assert [o["token_ids"] for o in llm.generate(_prompts(), sp)] == before
    finally:
        dist.destroy_process_group()


def _tp_worker(rank, world, port, model_dir, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "nano-vllm-hw3_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NANOVLLM_DIST_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo")
    print(f"[tp worker {rank}] start", flush=True)
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)    # a stalled worker reports where (all threads) and exits
    import warnings
    warnings.simplefilter("ignore")
    from nanovllm import LLM, SamplingParams
    from nanovllm.models.qwen3 import AttnContext, set_context
    res = {}
    for name, kw in (("bf16", {}), ("w8a16", dict(quant_type="per_row_weight_only", linear_dtype=torch.int8)),
                     ("w8a16_oneshot", dict(quant_type="per_row_weight_only", linear_dtype=torch.int8))):
        # the third pass: row-parallel partial sums all-reduced by the one-shot kernel over peer-mapped buffers, inside
        # the residual add + RMSNorm launch (nanovllm/layers/comm.py), instead of the process group's all-reduce
        os.environ["NANOVLLM_ONESHOT_AR"] = "1" if name.endswith("oneshot") else "0"
        llm = LLM(model_dir, load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16,
                  tensor_parallel_size=world, **kw)
        out = llm.generate(_prompts(), SamplingParams(max_tokens=6, ignore_eos=True))
        ids = torch.tensor(_prompts(1)[0], device="cuda")
        with torch.inference_mode():
            set_context(AttnContext(is_prefill=True, cu_seqlens=[0, ids.numel()], slot_mapping=None))
            h = llm.model_runner.model(ids, torch.arange(ids.numel(), device="cuda")).float().cpu()
        res[name] = dict(tokens=[o["token_ids"] for o in out], hidden=h)
        # the vocabulary-parallel sampler: per-shard maxima exchanged as {logit, token} pairs must give exactly the
        # tokens of the unsharded table (same rows on both ranks, the table re-assembled from the shards)
        import torch.distributed as dist
        from nanovllm import ops
        head = llm.model_runner.model.lm_head
        xg = torch.randn(37, head.weight.shape[1], generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
        shard = head.weight.data.cpu()
        parts = [torch.empty_like(shard) for _ in range(world)]
        dist.all_gather(parts, shard)
        with torch.inference_mode():
            assert head.fused_sampler(xg)
            res[name]["head_equal"] = bool(torch.equal(head.greedy_tokens(xg), ops.lm_head_argmax(xg, torch.cat(parts).cuda())))
        if name.endswith("oneshot"):
            from nanovllm.layers import comm
            res[name]["comm"] = comm.statuses()
        print(f"[tp worker {rank}] {name} done", flush=True)
    if rank == 0:
        torch.save(res, out_path)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_tensor_parallel_engine_two_ranks_one_gpu(tiny_dir, tmp_path):
    """The whole TP path (sharded weight loading from one seeded full model, column/row-parallel
    quantized linears, vocab-parallel embedding and head, SPMD scheduling) with two processes sharing
    the single GPU over gloo: TP=2 must reproduce TP=1 - hidden states to bf16 reduction noise (bf16)
    or quantization noise (row-parallel shards quantize their own K-shard), and both ranks must agree."""
    import torch.multiprocessing as mp
    from nanovllm import LLM, SamplingParams
    from nanovllm.models.qwen3 import AttnContext, set_context
    out_path = str(tmp_path / "tp2.pt")
    import time
    ctx = mp.spawn(_tp_worker, args=(2, 29731 + os.getpid() % 100, tiny_dir, out_path), nprocs=2, join=False)
    deadline = time.time() + 150
    done = False
    while time.time() < deadline and not done:
        # a worker that stalls dumps the Python stack of all its threads after 120 s (faulthandler in _tp_worker)
        # and exits non-zero: join() then raises ProcessExitedException and the test FAILS with the dump in the
        # captured stderr - a stall is never turned into a skip
        done = ctx.join(timeout=5)
    if not done:
        for pr in ctx.processes:
            pr.kill()
        pytest.fail("the two TP ranks neither finished nor reported a stall within 150 s")
    tp2 = torch.load(out_path, weights_only=False)
    for name, kw, tol in (("bf16", {}, 0.03), ("w8a16", dict(quant_type="per_row_weight_only", linear_dtype=torch.int8), 0.08)):
        llm = LLM(tiny_dir, load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16, **kw)
        ids = torch.tensor(_prompts(1)[0], device="cuda")
        with torch.inference_mode():
            set_context(AttnContext(is_prefill=True, cu_seqlens=[0, ids.numel()], slot_mapping=None))
            h1 = llm.model_runner.model(ids, torch.arange(ids.numel(), device="cuda")).float().cpu()
        rel = (h1 - tp2[name]["hidden"]).abs().max() / h1.abs().max()
        assert rel < tol, (name, float(rel))
        assert all(len(t) == 6 for t in tp2[name]["tokens"])
        assert tp2[name]["head_equal"], "vocabulary-parallel fused sampler differs from the unsharded one"
    # the one-shot all-reduce: used (calls were made, no wait timed out), and with two ranks the f32 sum of two bf16
    # partials rounds exactly like the process group's bf16 sum, so tokens and hidden states equal the library path's
    one = tp2["w8a16_oneshot"]
    assert one["comm"] and all(calls > 0 and timeouts == 0 for calls, timeouts in one["comm"].values()), one["comm"]
    assert one["tokens"] == tp2["w8a16"]["tokens"]
    assert torch.equal(one["hidden"], tp2["w8a16"]["hidden"])


@pytest.mark.parametrize("quant", [dict(quant_type="per_row_weight_only", linear_dtype=torch.int8),
                                   dict(quant_type="per_row", linear_dtype=torch.float8_e4m3fn),
                                   dict(quant_type="per_group", linear_dtype=torch.float8_e4m3fn, group_size=128),
                                   dict(quant_type="per_tensor", linear_dtype=torch.int8)])
def test_quantized_checkpoint_round_trip(tiny_dir, tmp_path, quant):
    """save_quantized -> LLM(load_format="quantized"): the stored tensors come back bit-identical (through
    the gate/up de-interleave and re-interleave) and the engine generates the same tokens without reading
    bf16 weights or quantizing."""
    from nanovllm import LLM, SamplingParams
    from nanovllm.utils.checkpoint import save_quantized
    sp = SamplingParams(max_tokens=8, ignore_eos=True)
    a = LLM(tiny_dir, load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16, **quant)
    out_dir = str(tmp_path / "q")
    save_quantized(a.model_runner.model, tiny_dir, out_dir)
    b = LLM(out_dir, load_format="quantized", enforce_eager=True, max_model_len=512, max_num_seqs=16)
    la, lb = a.model_runner.model.model.layers, b.model_runner.model.model.layers
    for x, y in zip(la, lb):
        for ma, mb in ((x.self_attn.qkv_proj, y.self_attn.qkv_proj), (x.self_attn.o_proj, y.self_attn.o_proj),
                       (x.mlp.gate_up_proj, y.mlp.gate_up_proj), (x.mlp.down_proj, y.mlp.down_proj)):
            assert type(ma) is type(mb) and ma.silu_fused == mb.silu_fused
            assert torch.equal(ma.rowmajor_qw().view(torch.uint8), mb.rowmajor_qw().view(torch.uint8))
            assert type(ma.qw).__name__ == type(mb.qw).__name__ == "TiledWeight"
            assert torch.equal(ma.sw.float().reshape(-1), mb.sw.float().reshape(-1))
    assert x.mlp.gate_up_proj.silu_fused
    ta = [o["token_ids"] for o in a.generate(_prompts(), sp)]
    tb = [o["token_ids"] for o in b.generate(_prompts(), sp)]
    assert ta == tb


def test_perplexity_harness_quantized_vs_bf16(tiny_dir):
    """The PPL loop on the tiny random model: quantized configurations must land next to bf16 (weights
    ~N(0, 0.02) make the logits small, so the NLL is ~ln(vocab) and quantization may move it only slightly),
    and the W8A16 per-row configuration closest of all."""
    import math
    from nanovllm import LLM
    from nanovllm.utils.ppl import total_nll
    g = torch.Generator().manual_seed(7)
    seqs = [torch.randint(0, 1000, (n,), generator=g).tolist() for n in (300, 70, 2, 1)]
    res = {}
    for name, kw in (("bf16", {}), ("w8a16", dict(quant_type="per_row_weight_only", linear_dtype=torch.int8)),
                     ("w8a8_fp8", dict(quant_type="per_row", linear_dtype=torch.float8_e4m3fn)),
                     ("g128", dict(quant_type="per_group", linear_dtype=torch.int8, group_size=128))):
        llm = LLM(tiny_dir, load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16, **kw)
        loss, count = total_nll(llm.model_runner.model, seqs, max_length=128)
        assert count == 299 + 69 + 1
        res[name] = loss / count
    assert abs(res["bf16"] - math.log(1024)) < 0.5
    for name in ("w8a16", "w8a8_fp8", "g128"):
        assert abs(res[name] - res["bf16"]) < 0.02, res
