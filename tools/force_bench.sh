# usage: tools/force_bench.sh "<mt wn wk cp>" ... : time forced tiles at prefill sizes (qkv, gate_up)
for cfg in "$@"; do
python3 -u - "$cfg" <<'PY'
import sys, os, torch
sys.path.insert(0, os.path.join(os.getcwd(), "nano-vllm-hw3_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from nanovllm import _qlinear as Q
from nanovllm.utils import quantization as NQ
import bench_kernels as BK
mt, wn, wk, cp = map(int, sys.argv[1].split())
lib = Q.load()
for name, N, K in [("qkv", 4096, 1024), ("gate_up", 6144, 1024), ("down", 1024, 3072)]:
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    q, s = NQ.per_row_quant(w, torch.int8)
    for M in (2048, 16384):
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        lib.ql_gemm_force_config(mt, wn, wk, cp, 1)
        us = BK.time_fn(lambda i: NQ.per_row_weight_only_matmul(x, q, s, None, torch.int8), 1, 10)
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
        print(f"tile {sys.argv[1]:10s} {name:8s} M={M:6d} {us:8.1f} us {2*M*N*K/us/1e6:7.1f} TF", flush=True)
PY
done
