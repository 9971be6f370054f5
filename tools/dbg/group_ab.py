"""Digest of per-group GEMM outputs over shapes, group sizes, forms and forced tiles (to compare two library builds
bit for bit: QLINEAR_LIB=... python tools/dbg/group_ab.py), followed by timings of the Qwen3-4B g128 shapes."""
import hashlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["QLINEAR_NO_PLANS"] = "1"
from nanovllm import _qlinear as Q
from nanovllm.utils import quantization as NQ
import bench_kernels as BK
lib = Q.load()
Q.workspace.reserve("cuda", 256 << 20)
torch.manual_seed(0)
h = hashlib.sha256()
n = 0
for (N, K) in [(256, 512), (2560, 4096), (1024, 3072), (1088, 1152)]:
    for group in (32, 96, 128, 384, 512):
        if K % group:
            continue
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        q, s = NQ.per_group_quant(w, group, torch.int8)
        qt = NQ.tile_weight(q)
        for M in (1, 33, 130, 300):
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            for t in [None] + BK.TILES:
                for sp in (1, 2, 3):
                    if t is None and sp > 1:
                        continue
                    if t is not None:
                        if t[0] == 8 or (t[0] * 32 >= 2 * max(M, 32) and t[0] > 1) or sp > max(1, K // (32 * t[3] * t[2])):
                            continue
                        assert lib.ql_gemm_force_config(*t, sp) == 0
                    for silu in ((False, True) if N % 64 == 0 and (t is None or t[1] % 2 == 0) and sp == 1 else (False,)):
                        y = NQ.per_group_matmul(x, qt, s, None, group, torch.int8, silu=silu)
                        h.update(y.cpu().view(torch.int16).numpy().tobytes())
                        n += 1
                    lib.ql_gemm_force_config(0, 0, 0, 0, 0)
print("cases", n, "digest", h.hexdigest(), flush=True)
for name, N, K in BK.SHAPES["4b"]:
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    q, s = NQ.per_group_quant(w, 128, torch.int8)
    nc = max(2, int(600e6 // (N * K)) + 1)
    qs = [NQ.tile_weight(q) for _ in range(nc)]
    for M in (16, 64, 128, 256, 2048):
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        us = BK.time_fn(lambda i: NQ.per_group_matmul(x, qs[i], s, None, 128, torch.int8), nc, 60)
        print(f"{name:8s} M={M:5d} {us:8.2f} us", flush=True)
