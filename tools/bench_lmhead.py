"""Output projection + greedy sampler at the Qwen3 vocabulary: torch's bf16 GEMM + ql_argmax_rows (two launches, logits
through memory) against ql_lmhead_argmax (sampler in the epilogue), per decode bucket, optionally per forced tile.

    python tools/bench_lmhead.py [--k 1024] [--ms 16,64,128,256] [--sweep]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from nanovllm import _qlinear as Q, ops  # noqa: E402
import bench_kernels as BK  # noqa: E402

TILES = [(1, 4, 1, 4), (2, 4, 1, 4), (4, 4, 1, 4), (4, 8, 1, 4), (5, 8, 1, 4), (6, 8, 1, 4), (7, 8, 1, 4), (8, 8, 1, 2), (5, 4, 1, 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--v", type=int, default=151936)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--ms", default="1,16,64,128,130,160,192,224,256")
    ap.add_argument("--sweep", action="store_true")
    a = ap.parse_args()
    lib = Q.load()
    V, K = a.v, a.k
    w = (torch.randn(V, K, device="cuda") * 0.02).to(torch.bfloat16)
    wbytes = V * K * 2
    for M in [int(m) for m in a.ms.split(",")]:
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        logits = torch.empty(M, V, device="cuda", dtype=torch.bfloat16)
        out = torch.empty(M, dtype=torch.int64, device="cuda")
        ops.reserve_head_workspace(M, V)

        def torch_path(i):
            torch.mm(x, w.t(), out=logits)
            ops.argmax_rows(logits)

        def plain_path(i):
            ops.argmax_rows(ops.lm_head_logits(x, w))

        t_torch = BK.time_fn(torch_path, 1, 20)
        t_plain = BK.time_fn(plain_path, 1, 20)
        t_fused = BK.time_fn(lambda i: ops.lm_head_argmax(x, w, out=out), 1, 20)
        print(f"M={M:4d}  torch mm + argmax {t_torch:7.1f} us | own logits + argmax {t_plain:7.1f} us | fused {t_fused:7.1f} us "
              f"({wbytes / t_fused / 1e6:5.2f} TB/s of table)", flush=True)
        if a.sweep:
            for t in TILES:
                if t[0] * 32 >= 2 * max(M, 32) and t[0] > 1:
                    continue
                lib.ql_gemm_force_config(*t, 1)
                us = BK.time_fn(lambda i: ops.lm_head_argmax(x, w, out=out), 1, 20)
                lib.ql_gemm_force_config(0, 0, 0, 0, 0)
                print(f"        tile {t}: {us:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
