"""gate/up projection with the SiLU epilogue at decode buckets, per forced tile (weights cycled through HBM)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from nanovllm import _qlinear as Q
from nanovllm.utils import quantization as NQ
import bench_kernels as BK
lib = Q.load()
N, K = 6144, 1024
nc = 100
ws = []
for i in range(nc):
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    q, s = NQ.per_row_quant(w, torch.int8)
    ws.append((NQ.tile_weight(NQ.interleave_gate_up(q)), NQ.interleave_gate_up(s)))
for M in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "128,160,192,224,256").split(",")]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    row = []
    for t in [None, (2, 4, 2, 4), (2, 4, 2, 16), (1, 4, 2, 16), (2, 2, 4, 8), (1, 2, 4, 8), (2, 4, 1, 4), (4, 4, 2, 4)]:
        if t: assert lib.ql_gemm_force_config(*t, 1) == 0
        us = BK.time_fn(lambda i: NQ.per_row_weight_only_matmul(x, ws[i][0], ws[i][1], None, torch.int8, silu=True), nc, 100)
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
        row.append(f"{'auto' if t is None else '.'.join(map(str, t))}: {us:5.2f}")
    print(f"M={M:4d}  " + "  ".join(row), flush=True)
