"""Graph-timed cost of (row-parallel-shaped GEMM -> residual add + RMSNorm) at decode sizes: self-contained GEMM +
norm against the deferred split-K form (GEMM partial + norm that reduces), per forced tile / split (diagnostic)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
from nanovllm import _qlinear as Q, ops
from nanovllm.utils import quantization as NQ
import bench_kernels as BK
lib = Q.load()
Q.workspace.reserve("cuda", 64 << 20)
SHAPES = {"o": (1024, 2048), "down": (1024, 3072)}
CFGS = [None, (1, 1, 8, 4, 2), (1, 1, 8, 4, 3), (1, 2, 4, 8, 2), (1, 2, 4, 8, 3)]
names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(SHAPES)
for name in names:
    N, K = SHAPES[name]
    NC = max(2, int(400e6 // (N * K)))
    mods = [NQ.QLinear(torch.int8, (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16), None, weight_only=True).tile()
            for _ in range(NC)]
    nw = torch.ones(N, device="cuda", dtype=torch.bfloat16)
    for M in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8,32,64,128,192,256").split(",")]:
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
        plain = BK.time_fn(lambda i: ops.add_rmsnorm(mods[i](x), res, nw, 1e-6), NC, 60)
        gemm = BK.time_fn(lambda i: mods[i](x), NC, 60)
        out = []
        for cfg in CFGS:
            if cfg is not None:
                assert lib.ql_gemm_force_config(*cfg) == 0
            S = lib.ql_gemm_partial_slices(M, N, K, 0)
            t = BK.time_fn(lambda i: ops.add_rmsnorm(NQ.w8a16_partial_matmul(mods[i], x), res, nw, 1e-6), NC, 60)
            tg = BK.time_fn(lambda i: NQ.w8a16_partial_matmul(mods[i], x), NC, 60)
            lib.ql_gemm_force_config(0, 0, 0, 0, 0)
            out.append((cfg, S, t, tg))
        best = min(out, key=lambda o: o[2])
        print(f"{name:5s} M={M:4d} plain gemm+norm {plain:5.2f} (gemm {gemm:5.2f}) | deferred auto S={out[0][1]} {out[0][2]:5.2f} (gemm {out[0][3]:5.2f}) | best "
              f"{'.'.join(map(str, best[0])) if best[0] else 'auto'} S={best[1]} {best[2]:5.2f} (gemm {best[3]:5.2f})", flush=True)
        print("      " + "  ".join(f"{'.'.join(map(str, c)) if c else 'auto'}:{t:4.1f}" for c, S, t, tg in out), flush=True)
