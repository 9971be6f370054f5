"""(Experiment record: needs the parked kernel tools/dbg/ql_gemm_allrows.hip wired in as its header says.)  gate/up + SiLU on the all-rows-per-workgroup kernel (QL_WIDE=1) against the strip-per-wave kernels (forced tile)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["QL_WIDE"] = "1"
from nanovllm import _qlinear as Q
from nanovllm.utils import quantization as NQ
import bench_kernels as BK
lib = Q.load()
Q.workspace.reserve("cuda", 256 << 20)
torch.manual_seed(0)
shapes = [(2048, 512), (19456, 2560), (24576, 4096)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for N, K in shapes:
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    for mode in ("row", "group"):
        for dt in (torch.int8, torch.float8_e4m3fn):
            if mode == "row": q, s = NQ.per_row_quant(w, dt)
            else: q, s = NQ.per_group_quant(w, 128, dt)
            for tiled in (True, False):
                qq = NQ.tile_weight(q) if tiled else q
                for M in (129, 160, 200, 256):
                    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
                    f = (lambda: NQ.per_row_weight_only_matmul(x, qq, s, None, dt, silu=True)) if mode == "row" else \
                        (lambda: NQ.per_group_matmul(x, qq, s, None, 128, dt, silu=True))
                    y = f()
                    lib.ql_gemm_force_config(2, 4, 2, 4, 1)
                    y0 = f()
                    lib.ql_gemm_force_config(0, 0, 0, 0, 0)
                    d = (y.float() - y0.float()).abs()
                    tol = 2 ** -6 * y0.float().abs() + 6e-2
                    bad = int((d > tol).sum())
                    print(f"N={N} K={K} {mode} {dt} tiled={tiled} M={M}: max diff {float(d.max()):.4g}, bad {bad}/{d.numel()}, equal {float((y == y0).float().mean()):.4f}", flush=True)
                    assert bad == 0
# timing on the big shapes
for N, K in shapes:
    if N * K < 8e6: continue
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    n_copies = max(2, int(600e6 // (N * K)) + 1)
    for mode in ("row", "group"):
        dt = torch.int8
        q, s = NQ.per_row_quant(w, dt) if mode == "row" else NQ.per_group_quant(w, 128, dt)
        qs = [NQ.tile_weight(q.clone()) for _ in range(n_copies)]
        for M in (129, 160, 192, 224, 256):
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            f = (lambda i: NQ.per_row_weight_only_matmul(x, qs[i], s, None, dt, silu=True)) if mode == "row" else \
                (lambda i: NQ.per_group_matmul(x, qs[i], s, None, 128, dt, silu=True))
            t_new = BK.time_fn(f, n_copies, 40)
            lib.ql_gemm_force_config(-1, 0, 0, 0, 0) if False else None
            os.environ["X"] = "1"
            print(f"N={N} K={K} {mode} M={M}: all-rows kernel {t_new:.1f} us ({2*M*N*K/t_new/1e6:.0f} TFLOP/s)", flush=True)
