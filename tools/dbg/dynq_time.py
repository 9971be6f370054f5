"""W8A8 o_proj / down_proj: separate activation quantization launch + GEMM against the GEMM that quantizes while staging."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from nanovllm.utils import quantization as NQ
import bench_kernels as BK
for tdt in (torch.int8, torch.float8_e4m3fn):
    for name, N, K, parts in [("o", 1024, 2048, 8), ("down", 1024, 3072, 96)]:
        nc = 64
        ws = []
        for i in range(nc):
            w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
            q, s = NQ.per_row_quant(w, tdt)
            ws.append((NQ.tile_weight(q), s))
        for M in (16, 64, 128, 256, 2048):
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            pm = NQ.row_absmax_parts(x, parts)
            qx, sx = NQ.per_row_quant(x, tdt)
            t_sep = BK.time_fn(lambda i: NQ.per_row_matmul(x, ws[i][0], ws[i][1], None, tdt), nc, 64)
            t_gemm = BK.time_fn(lambda i: NQ.w8a8_matmul(qx, sx, ws[i][0], ws[i][1], None, tdt), nc, 64)
            t_dyn = BK.time_fn(lambda i: NQ.w8a8_dynq_matmul(x, pm, ws[i][0], ws[i][1], None, tdt), nc, 64)
            print(f"{str(tdt)[6:]:14s} {name:5s} M={M:5d}  quant+gemm {t_sep:6.2f}  gemm alone {t_gemm:6.2f}  dynq gemm {t_dyn:6.2f} us", flush=True)
