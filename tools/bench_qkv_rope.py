"""Graph-timed cost of the qkv projection + q/k norm + RoPE + KV store at decode sizes: fused epilogue (per forced
tile / split) against GEMM + stand-alone operator (diagnostic)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
from nanovllm import _qlinear as Q, ops
from nanovllm.utils import quantization as NQ
from nanovllm.models.qwen3 import build_cos_sin
import bench_kernels as BK
lib = Q.load()
Hq, Hkv, K = 16, 8, 1024
N = (Hq + 2 * Hkv) * 128
NC = 40
mods = []
for i in range(NC):
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    mods.append(NQ.QLinear(torch.int8, w, None, weight_only=True).tile())
cs = build_cos_sin(128, 4096, 1e6, "cuda")
nw = torch.ones(128, device="cuda", dtype=torch.bfloat16)
kc = torch.zeros(64, 256, Hkv, 128, device="cuda", dtype=torch.bfloat16); vc = torch.zeros_like(kc)
CFGS = [None, (2, 4, 2, 4, 1)]
Q.workspace.reserve("cuda", 64 << 20)
for M in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8,16,32,48,64,96,128,160,192,224,256").split(",")]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    pos = torch.randint(0, 2000, (M,), device="cuda")
    slots = torch.randperm(64 * 256, device="cuda")[:M].to(torch.int32)
    def unfused(i):
        qkv = mods[i](x)
        ops.qknorm_rope_store(qkv, pos, cs, nw, nw, 1e-6, Hq, Hkv, 128, kc, vc, slots, False)
    rows = cs.index_select(0, pos)
    def fused(i):
        if os.environ.get("ROWS", "1") == "1":
            NQ.qkv_rope_matmul(mods[i], x, None, rows, nw, nw, 1e-6, Hq, Hkv, 128, kc, vc, slots, False)
        else:
            NQ.qkv_rope_matmul(mods[i], x, pos, cs, nw, nw, 1e-6, Hq, Hkv, 128, kc, vc, slots, False)
    base = BK.time_fn(unfused, NC, 80) 
    gemm_only = BK.time_fn(lambda i: mods[i](x), NC, 80)
    res = []
    for cfg in CFGS:
        if cfg is not None:
            assert lib.ql_gemm_force_config(*cfg) == 0
        res.append(BK.time_fn(fused, NC, 80))
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
    print(f"M={M:4d} unfused(2 launches) {base:5.2f}  gemm alone {gemm_only:5.2f} | fused: " +
          "  ".join(f"{'auto' if c is None else '.'.join(map(str, c))}={t:5.2f}" for c, t in zip(CFGS, res)), flush=True)
