"""Per-shape micro-benchmark of the fused dequant GEMMs on one GPU (not the headline bench).

Times ql_gemm_w8a16 / ql_gemm_w8a8 at the Qwen3 layer shapes over a range of M with HIP events on
the launch stream, cycling through enough distinct weight copies that the weights are streamed from
HBM (footprint > the 256 MiB Infinity Cache), and optionally sweeps the forced tile configurations.
"""
import argparse
import itertools
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
from nanovllm import _qlinear as Q  # noqa: E402
from nanovllm.utils import quantization as NQ  # noqa: E402

SHAPES = {
    "0.6b": [("qkv", 4096, 1024), ("o", 1024, 2048), ("gate_up", 6144, 1024), ("down", 1024, 3072)],
    "4b": [("qkv", 6144, 2560), ("o", 2560, 4096), ("gate_up", 19456, 2560), ("down", 2560, 9728)],
    "8b_tp8": [("qkv", 768, 4096), ("o", 4096, 512), ("gate_up", 3072, 4096), ("down", 4096, 1536)],
}
TILES = [(1, 1, 4, 2), (1, 1, 4, 4), (1, 1, 4, 8), (2, 1, 4, 2), (2, 1, 4, 4), (1, 4, 1, 4), (2, 4, 1, 4), (4, 4, 1, 4),
         (8, 4, 1, 4), (2, 2, 2, 2), (2, 2, 2, 4), (2, 2, 2, 8), (4, 2, 2, 2), (4, 2, 2, 4),
         (1, 2, 4, 8), (1, 2, 4, 4), (2, 2, 4, 8), (2, 2, 4, 4), (1, 1, 8, 4), (2, 4, 2, 4), (4, 4, 2, 4),
         (1, 4, 1, 32), (1, 4, 2, 16), (2, 4, 2, 16), (1, 2, 4, 16), (1, 1, 8, 8), (4, 8, 1, 2), (4, 8, 1, 4)]


def time_fn(fn, n_copies, iters):
    """Device time per call: the calls are captured into one hipGraph (no host launch cost between
    kernels; the ~1.5 us dependent-kernel boundary stays, as in the captured decode step)."""
    for i in range(min(n_copies, 3)):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(iters):
                fn(i % n_copies)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    del g
    return best   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="0.6b")
    ap.add_argument("--ms", default="1,16,32,64,128,256,512,2048,16384")
    ap.add_argument("--mode", default="w8a16", choices=["w8a16", "w8a8", "w8a8fp8", "group", "bf16"])
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--out", default="")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--copies", type=int, default=0, help="weight copies cycled (0 = enough to defeat the 256 MiB cache)")
    ap.add_argument("--shapes", default="", help="comma list of shape names to run")
    ap.add_argument("--tiles", default="", help="with --sweep: only these tiles, e.g. 4.8.1.2,4.8.1.4")
    ap.add_argument("--splits", default="1,2,4,8")
    ap.add_argument("--tiled", action="store_true", help="weights in the tiled layout (tile_weight)")
    args = ap.parse_args()
    lib = Q.load()
    dev = "cuda"
    rows = []
    for name, N, K in SHAPES[args.model]:
        if args.shapes and name not in args.shapes.split(","):
            continue
        n_copies = args.copies if args.copies > 0 else max(2, int(600e6 // (N * K)) + 1)
        w = (torch.randn(n_copies, N, K, device=dev) * 0.02).to(torch.bfloat16) if args.mode == "bf16" else None
        dt = torch.float8_e4m3fn if args.mode == "w8a8fp8" else torch.int8
        if args.mode != "bf16":
            w1 = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
            if args.mode == "group":
                q1, s1 = NQ.per_group_quant(w1, 128, dt)
            else:
                q1, s1 = NQ.per_row_quant(w1, dt)
            qw = q1.unsqueeze(0).repeat(n_copies, 1, 1).contiguous()
            if args.tiled:
                qw = [NQ.tile_weight(qw[i]) for i in range(n_copies)]
        for M in [int(m) for m in args.ms.split(",")]:
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            Q.workspace.reserve(dev, 256 << 20)
            if args.mode == "w8a16":
                fn = lambda i: NQ.per_row_weight_only_matmul(x, qw[i], s1, None, dt)
            elif args.mode == "group":
                fn = lambda i: NQ.per_group_matmul(x, qw[i], s1, None, 128, dt)
            elif args.mode in ("w8a8", "w8a8fp8"):
                qx, sx = NQ.per_row_quant(x, dt)
                fn = lambda i: NQ.w8a8_matmul(qx, sx, qw[i], s1, None, dt)
            else:
                fn = lambda i: torch.nn.functional.linear(x, w[i])
            iters = args.iters if M <= 2048 else max(10, args.iters // 10)
            cfgs = [None]
            if args.sweep and args.mode != "bf16":
                tiles = [tuple(int(v) for v in t.split(".")) for t in args.tiles.split(",")] if args.tiles else TILES
                cfgs += [(t, s) for t in tiles for s in [int(v) for v in args.splits.split(",")]
                         if not (args.mode == "group" and t[0] == 8)]
            best = None
            for cfg in cfgs:
                if cfg is not None:
                    (mt, wn, wk, cp), sp = cfg
                    if mt * 32 >= 2 * max(M, 32) and mt > 1:
                        continue
                    kb = 32 * cp * wk
                    if sp > max(1, K // kb):
                        continue
                    lib.ql_gemm_force_config(mt, wn, wk, cp, sp)
                    # (a negative split asks for the stream-K schedule: only on the tiles built for it)
                    form = 1 if args.mode == "group" else 8 if args.mode in ("w8a8", "w8a8fp8") else 0
                    if sp < 0 and Q.plan_query(form, M, N, K)[4] != sp:
                        continue
                else:
                    lib.ql_gemm_force_config(0, 0, 0, 0, 0)
                try:
                    us = time_fn(fn, n_copies, iters)
                except Exception as ex:  # workspace too small for a forced split etc.
                    print("skip", cfg, ex)
                    continue
                wbytes = N * K * (2 if args.mode == "bf16" else 1)
                abytes = wbytes + 2 * M * K + 2 * M * N + 4 * N
                rec = dict(shape=name, N=N, K=K, M=M, cfg=cfg, us=round(us, 2), GBps=round(abytes / us / 1e3, 1),
                           TFLOPs=round(2 * M * N * K / us / 1e6, 1))
                if cfg is None or args.verbose:
                    rows.append(rec)
                    print(json.dumps(rec), flush=True)
                if best is None or us < best["us"]:
                    best = rec
            lib.ql_gemm_force_config(0, 0, 0, 0, 0)
            if args.sweep and best is not None:
                best = dict(best, best=True)
                rows.append(best)
                print(json.dumps(best), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
