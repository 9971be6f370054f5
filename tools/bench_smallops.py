"""Graph-timed cost of the small operators between the GEMMs at decode sizes (diagnostic)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from nanovllm import ops
from nanovllm.models.qwen3 import build_cos_sin
import bench_kernels as BK
H, Hq, Hkv, D, I = 1024, 16, 8, 128, 3072
w = torch.ones(H, device="cuda", dtype=torch.bfloat16)
cs = build_cos_sin(D, 4096, 1e6, "cuda")
nw = torch.ones(D, device="cuda", dtype=torch.bfloat16)
for M in (16, 64, 128, 256):
    x = torch.randn(M, H, device="cuda", dtype=torch.bfloat16); r = torch.randn_like(x)
    qkv = torch.randn(M, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.bfloat16)
    pos = torch.randint(0, 2000, (M,), device="cuda")
    kc = torch.zeros(64, 256, Hkv, D, device="cuda", dtype=torch.bfloat16); vc = torch.zeros_like(kc)
    slots = torch.randperm(64 * 256, device="cuda")[:M].to(torch.int32)
    gu = torch.randn(M, 2 * I, device="cuda", dtype=torch.bfloat16)
    a = BK.time_fn(lambda i: ops.add_rmsnorm(x, r, w, 1e-6), 1, 200)
    b = BK.time_fn(lambda i: ops.qknorm_rope_store(qkv, pos, cs, nw, nw, 1e-6, Hq, Hkv, D, kc, vc, slots), 1, 200)
    c = BK.time_fn(lambda i: ops.silu_mul(gu), 1, 200)
    print(f"M={M:4d}  add_rmsnorm {a:5.2f}  qknorm_rope_store {b:5.2f}  silu_mul {c:5.2f} us (per launch, back to back in a graph)", flush=True)
