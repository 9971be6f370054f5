import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
from nanovllm import _qlinear as Q, ops
from nanovllm.utils import quantization as NQ
import bench_kernels as BK
N = 1024
nw = torch.ones(N, device="cuda", dtype=torch.bfloat16)
sw = torch.rand(N, device="cuda")
for M in (8, 64, 128, 256):
    res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    y = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = BK.time_fn(lambda i: ops.add_rmsnorm(y, res, nw, 1e-6), 1, 200)
    line = f"M={M:4d} plain {t0:5.2f}"
    for S in (1, 2, 3, 6):
        ps = NQ.PartialSum(torch.randn(S, M, N, device="cuda"), 0, sw, None, (M, N))
        t = BK.time_fn(lambda i: ops.add_rmsnorm(ps, res, nw, 1e-6), 1, 200)
        line += f"  S={S}: {t:5.2f}"
    print(line, flush=True)
