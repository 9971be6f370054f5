"""The four layer GEMMs of Qwen3-0.6B at a prefill-sized batch, as modules run them (gate/up with the SiLU epilogue),
per forced tile: graph-timed per shape, and their sum as TFLOP/s (what bench.py's roofline_prefill leg adds up).

    python tools/dbg/prefill_forms.py [M] [tiles, e.g. auto,4.8.1.2,4.8.1.4]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from nanovllm import _qlinear as Q  # noqa: E402
from nanovllm.utils import quantization as NQ  # noqa: E402
import bench_kernels as BK  # noqa: E402

lib = Q.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
tiles = (sys.argv[2] if len(sys.argv) > 2 else "auto,4.8.1.2,4.8.1.4").split(",")
dtype = torch.float8_e4m3fn if os.environ.get("FP8") else torch.int8
w8a8 = bool(os.environ.get("W8A8"))
SHAPES = [("qkv", 4096, 1024, False), ("o", 1024, 2048, False), ("gate_up", 6144, 1024, True), ("down", 1024, 3072, False)]
mods = []
for name, N, K, silu in SHAPES:
    m = NQ.QLinear(dtype, (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16), None, weight_only=not w8a8)
    if silu:
        m.enable_silu_mul()
    mods.append(m.tile())
Q.workspace.reserve("cuda", 64 << 20)
for t in tiles:
    if t != "auto":
        assert lib.ql_gemm_force_config(*[int(v) for v in t.split(".")], 1) == 0
    tot_us, tot_flop, row = 0.0, 0.0, []
    with torch.inference_mode():
        for (name, N, K, silu), m in zip(SHAPES, mods):
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            us = BK.time_fn(lambda i: m(x), 1, 10)
            fl = 2.0 * M * N * K
            tot_us += us
            tot_flop += fl
            row.append(f"{name} {us:7.1f} us {fl / us / 1e6:6.0f} TF")
    lib.ql_gemm_force_config(0, 0, 0, 0, 0)
    print(f"M={M} tile {t:9s} " + " | ".join(row) + f" | sum {tot_flop / tot_us / 1e6:6.0f} TF", flush=True)

# the same four launches back to back, eager, 5 rounds between two events (bench.py's roofline_prefill leg)
for t in tiles:
    if t != "auto":
        assert lib.ql_gemm_force_config(*[int(v) for v in t.split(".")], 1) == 0
    with torch.inference_mode():
        xs = [torch.randn(M, K, device="cuda", dtype=torch.bfloat16) for _, N, K, _ in SHAPES]
        res = []
        for rep in range(3):
            for x, m in zip(xs, mods):
                m(x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                for x, m in zip(xs, mods):
                    m(x)
            e.record()
            torch.cuda.synchronize()
            res.append(sum(2.0 * M * N * K for _, N, K, _ in SHAPES) / (s.elapsed_time(e) * 1e-3 / 5) / 1e12)
    lib.ql_gemm_force_config(0, 0, 0, 0, 0)
    print(f"M={M} tile {t:9s} eager leg: " + " ".join(f"{r:6.0f}" for r in res) + " TF", flush=True)

# sustained: the same eager rounds for SUSTAIN seconds per tile (clocks settle under load), rate over the second half
import time
sustain = float(os.environ.get("SUSTAIN", "0"))
if sustain > 0:
    for t in tiles:
        if t != "auto":
            assert lib.ql_gemm_force_config(*[int(v) for v in t.split(".")], 1) == 0
        with torch.inference_mode():
            rates, t0 = [], time.time()
            while time.time() - t0 < sustain:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    for x, m in zip(xs, mods):
                        m(x)
                e.record()
                torch.cuda.synchronize()
                rates.append(sum(2.0 * M * N * K for _, N, K, _ in SHAPES) / (s.elapsed_time(e) * 1e-3 / 20) / 1e12)
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
        h = rates[len(rates) // 2:]
        print(f"M={M} tile {t:9s} sustained {sustain:.0f} s: first {rates[0]:6.0f}, second half mean {sum(h) / len(h):6.0f} TF", flush=True)
