"""PMC target: the prefill W8A16 GEMM at M = 16384 on the four Qwen3-0.6B layer shapes, 3 launches each (the first
of each shape warms the code path).  rocprofv3 --pmc <counters> -- python3 tools/pmc_prefill.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
from nanovllm.utils import quantization as NQ
torch.manual_seed(0)
M = 16384
for name, N, K in [("qkv", 4096, 1024), ("o", 1024, 2048), ("gate_up", 6144, 1024), ("down", 1024, 3072)]:
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    q, s = NQ.per_row_quant(w, torch.int8)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        NQ.per_row_weight_only_matmul(x, q, s, None, torch.int8)
        torch.cuda.synchronize()
