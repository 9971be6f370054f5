"""Decode attention throughput vs how evenly the (sequence, kv head, chunk) workgroups fill the chip (diagnostic)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
from nanovllm import ops
Hq, Hkv, D, bs = 16, 8, 128, 256
def run(B, L):
    torch.manual_seed(0)
    nb = (L + bs - 1) // bs
    nblk = B * nb + 1
    Lyr = max(2, int(600e6 // (nblk * bs * Hkv * D * 4)) + 1)
    kc = [torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16) for _ in range(Lyr)]
    vc = [torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16) for _ in range(Lyr)]
    tables = (torch.arange(B * nb, device="cuda", dtype=torch.int32) + 1).view(B, nb)
    tables = torch.nn.functional.pad(tables, (0, 16 - nb)).contiguous()
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.bfloat16)
    ctx = torch.full((B,), L, dtype=torch.int32, device="cuda")
    ops.reserve_attn_workspace(B, Hq, Hkv, bs, 16)
    for i in range(Lyr): ops.paged_attn_decode(q, kc[i], vc[i], tables, ctx, D ** -0.5)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            for i in range(Lyr): ops.paged_attn_decode(q, kc[i], vc[i], tables, ctx, D ** -0.5)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / (4 * Lyr)
    byt = B * L * Hkv * D * 4
    print(f"B={B:4d} L={L:5d} blocks={B*Hkv*((L+511)//512):5d}  {us:7.1f} us  {byt/us/1e6:5.2f} TB/s", flush=True)
for B, L in [(160, 512), (176, 512), (240, 512), (320, 512), (160, 1024), (176, 1024), (131, 1024), (131, 1100), (64, 2048), (32, 4096), (256, 1024)]:
    run(B, L)
