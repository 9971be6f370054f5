"""Fused gate/up + SiLU*mul epilogue vs GEMM followed by the stand-alone operator (diagnostic)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from nanovllm import _qlinear as Q, ops
from nanovllm.utils import quantization as NQ
import bench_kernels as BK
lib = Q.load()
N, K = 6144, 1024
C = 100
ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(C)]
qs = [NQ.per_row_quant(w, torch.int8) for w in ws]
qi = [(NQ.interleave_gate_up(q), NQ.interleave_gate_up(s)) for q, s in qs]
tiles = [None] + [tuple(map(int, t.split(","))) for t in sys.argv[1:]]
for M in (16, 64, 128, 256):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    for tile in tiles:
        if tile: lib.ql_gemm_force_config(*tile, 1)
        a = BK.time_fn(lambda i: NQ.per_row_weight_only_matmul(x, qs[i][0], qs[i][1], None, torch.int8), C, 200)
        b = BK.time_fn(lambda i: ops.silu_mul(NQ.per_row_weight_only_matmul(x, qs[i][0], qs[i][1], None, torch.int8)), C, 200)
        c = BK.time_fn(lambda i: NQ.per_row_weight_only_matmul(x, qi[i][0], qi[i][1], None, torch.int8, silu=True), C, 200)
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
        print(f"M={M:4d} tile={tile}  gemm {a:6.2f}  gemm+silu {b:6.2f}  fused {c:6.2f} us", flush=True)
