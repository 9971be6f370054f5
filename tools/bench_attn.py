"""Micro-benchmark of decode attention on the bench workload's shape mix (diagnostic)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
from nanovllm import ops
torch.manual_seed(0)
B, Hq, Hkv, D, bs = int(os.environ.get("BATCH", "131")), int(os.environ.get("HQ", "16")), 8, 128, 256      # HQ=32: the Qwen3-4B / 8B head ratio
lens = torch.randint(300, 1800, (B,))
nblk = int(((lens + bs - 1) // bs).sum()) + 1
L = 8   # layers' worth of distinct caches so that the KV streams from HBM
kc = [torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16) for _ in range(L)]
vc = [torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16) for _ in range(L)]
W = int(os.environ.get("WIDTH", "16"))          # block-table width: 16 = max_model_len 4096; the grid has W / 2 chunks
tables = torch.zeros(B, W, dtype=torch.int32, device="cuda")
nxt = 1
for b in range(B):
    for j in range((int(lens[b]) + bs - 1) // bs):
        tables[b, j] = nxt; nxt += 1
q = torch.randn(B, Hq, D, device="cuda", dtype=torch.bfloat16)
ctx = lens.to(torch.int32).cuda()
ops.reserve_attn_workspace(B, Hq, Hkv, bs, W)
for i in range(L): ops.paged_attn_decode(q, kc[i], vc[i], tables, ctx, D ** -0.5)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(4):
        for i in range(L): ops.paged_attn_decode(q, kc[i], vc[i], tables, ctx, D ** -0.5)
g.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / (4 * L)
byt = float(lens.sum()) * Hkv * D * 2 * 2
print(f"decode attention: {us:.1f} us/call, {byt/1e6:.0f} MB KV, {byt/us/1e6:.2f} TB/s")
