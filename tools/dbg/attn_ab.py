"""Digest + timing of the decode-attention entry points (plain, row-max, quantizing) on multi-chunk sequences.
Run once per library build (QLINEAR_LIB=...) and compare the printed digests: equal digests = bit-identical outputs."""
import hashlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
from nanovllm import ops

D, bs, Hkv = 128, 256, 8
h = hashlib.sha256()


def case(B, Hq, lo, hi, width, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi, (B,), generator=g)
    nblk = int(((lens + bs - 1) // bs).sum()) + 1
    kc = torch.randn(nblk, bs, Hkv, D, generator=g).to("cuda", torch.bfloat16)
    vc = torch.randn(nblk, bs, Hkv, D, generator=g).to("cuda", torch.bfloat16)
    tables = torch.zeros(B, width, dtype=torch.int32)
    nxt = 1
    for b in range(B):
        for j in range((int(lens[b]) + bs - 1) // bs):
            tables[b, j] = nxt; nxt += 1
    tables = tables.cuda()
    q = torch.randn(B, Hq, D, generator=g).to("cuda", torch.bfloat16)
    ctx = lens.to(torch.int32).cuda()
    ops.reserve_attn_workspace(B, Hq, Hkv, bs, width)
    outs = [ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5)]
    if Hq // Hkv in (1, 2, 4):
        outs += list(ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5, return_rowmax=True))
    for dt in (torch.int8, torch.float8_e4m3fn):
        outs += list(ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5, quant_dtype=dt))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        raw = o.contiguous().view(torch.uint8).cpu().numpy().tobytes()
        h.update(raw)
        print(f"  out[{i}] {tuple(o.shape)} {o.dtype} {hashlib.sha256(raw).hexdigest()[:12]}")
    # timing of the plain call in a graph
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(32): ops.paged_attn_decode(q, kc, vc, tables, ctx, D ** -0.5)
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    print(f"B={B} Hq={Hq} ctx {lo}..{hi}: {s.elapsed_time(e) * 1e3 / 32:.2f} us/call", flush=True)


for Hq in (8, 16, 32):
    case(64, Hq, 300, 1800, 8, 1)
    case(24, Hq, 1500, 6100, 24, 2)     # up to 12 chunks: beyond the merge's in-flight depth
    case(8, Hq, 513, 1025, 8, 3)
print("digest", h.hexdigest())
