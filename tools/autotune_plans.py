"""Measure the decode-sized GEMM launches of a model's layer shapes over the built tiles and K splits, and write the
choices that beat the library's built-in heuristics into the plan table the library loads at start-up
(nano-vllm-hw3_amd/nanovllm/_plans/gfx950.json, see ql_gemm_plan_set in include/qlinear.h).

Every launch is timed in the form the decode graph issues it: the qkv projection with its q/k norm + RoPE + KV-store
epilogue, gate/up with the SiLU epilogue, o_proj / down_proj both self-contained and with the split-K reduction left
to the residual add + RMSNorm launch (timed together with that launch).  Weights cycle through more copies than the
256 MiB memory-side cache holds; calls are captured into one hipGraph (the dependent-launch boundary stays in).

    python tools/autotune_plans.py --model 4b --mode group --out nano-vllm-hw3_amd/nanovllm/_plans/gfx950.json --merge
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("QLINEAR_NO_PLANS", "1")           # measure against the heuristics, not an older table
from nanovllm import _qlinear as Q, ops  # noqa: E402
from nanovllm.models.qwen3 import build_cos_sin  # noqa: E402
from nanovllm.utils import quantization as NQ  # noqa: E402
import bench_kernels as BK  # noqa: E402

# name -> (heads, kv heads, hidden, intermediate) of the rank-local layer
MODELS = {
    "0.6b": (16, 8, 1024, 3072),
    "1.7b": (16, 8, 2048, 6144),
    "4b": (32, 8, 2560, 9728),
    "8b": (32, 8, 4096, 12288),
    "8b_tp8": (4, 1, 4096, 1536),
    # the headline model's rank-local layers under tensor parallelism (bench.py --gpus 2 / 4 / 8)
    "0.6b_tp2": (8, 4, 1024, 1536),
    "0.6b_tp4": (4, 2, 1024, 768),
    "0.6b_tp8": (2, 1, 1024, 384),
}
TILES = BK.TILES
ROPE_TILES = [(1, 4, 2, 16), (1, 4, 1, 4), (2, 4, 2, 4), (2, 4, 1, 4)]
PARTIAL_TILES = [(1, 1, 8, 4), (1, 1, 8, 8), (1, 2, 4, 8), (1, 2, 4, 4), (2, 2, 4, 8), (2, 2, 4, 4), (1, 4, 1, 4),
                 (1, 4, 2, 16), (2, 4, 1, 4), (2, 4, 2, 4), (1, 2, 4, 16)]
# tiles built with the stream-K schedule (QL_TILE_LIST_SK* in csrc/ql_gemm_kernel.h) and the grid sizes tried on them:
# one to four workgroups per CU (a tile's registers decide how many of them are resident at once)
SK_TILES = [(1, 1, 4, 4), (1, 2, 4, 4), (2, 2, 4, 4), (2, 2, 2, 4), (1, 4, 1, 4), (2, 4, 1, 4), (4, 4, 1, 4), (2, 4, 2, 4),
            (4, 4, 2, 4), (4, 8, 1, 2), (4, 8, 1, 4), (1, 1, 8, 4), (1, 2, 4, 8), (1, 4, 2, 16), (8, 4, 1, 4)]
SK_ROPE_TILES = [(1, 4, 1, 4), (2, 4, 2, 4), (2, 4, 1, 4), (4, 8, 1, 2)]


def sk_grids(t):
    return (256, 512) if t[1] * t[2] == 8 else (256, 512, 768, 1024)


MS = [16, 64, 96, 128, 160, 192, 224, 256, 320, 384, 448, 512]
PREFILL_MS = [1024, 2048, 4096, 8192, 16384]
# prefill sizes: the double-buffered tiles of 64 rows and more, never split (thousands of tiles fill the chip)
PREFILL_TILES = [(2, 4, 1, 4), (4, 4, 1, 4), (8, 4, 1, 4), (2, 4, 2, 4), (4, 4, 2, 4), (4, 8, 1, 2), (4, 8, 1, 4), (2, 2, 2, 4), (4, 2, 2, 4)]
PREFILL_ROPE_TILES = [(2, 4, 1, 4), (2, 4, 2, 4), (4, 8, 1, 2), (4, 8, 1, 4)]
MIN_GAIN = 0.03


def ranges(ms, lo=1):
    for m in ms:
        hi = (m + 31) // 32 * 32
        yield m, lo, hi
        lo = hi + 1


def make_mod(N, K, mode, dt, silu=False):
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    m = NQ.QGroupLinear(dt, w, None, 128) if mode == "group" else NQ.QLinear(dt, w, None, weight_only=(mode != "w8a8"))
    if silu:
        m.enable_silu_mul()
    return m.tile()


def tune_one(lib, fn, n_copies, cands, iters):
    """cands: list of (tile, split) with None = the heuristic; returns {cand: us} after a confirming second pass."""
    res = {}
    for c in cands:
        if c is None:
            lib.ql_gemm_force_config(0, 0, 0, 0, 0)
        elif lib.ql_gemm_force_config(*c[0], c[1]) != 0:
            continue
        try:
            res[c] = BK.time_fn(fn, n_copies, iters)
        except Exception:  # noqa: BLE001  (a split the form cannot run, a workspace a forced split outgrows)
            pass
        finally:
            lib.ql_gemm_force_config(0, 0, 0, 0, 0)
    top = sorted((c for c in res if c is not None), key=lambda c: res[c])[:3]
    for c in ([None] if None in res else []) + top:       # confirm the leaders and the baseline with a longer run
        if c is None:
            lib.ql_gemm_force_config(0, 0, 0, 0, 0)
        else:
            lib.ql_gemm_force_config(*c[0], c[1])
        res[c] = min(res[c], BK.time_fn(fn, n_copies, iters * 3))
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
    return res, top


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="4b", choices=sorted(MODELS))
    ap.add_argument("--mode", default="row", choices=["row", "group", "w8a8"],
                    help="row / group: W8A16 with per-row / g128 scales; w8a8: per-row W8A8 (quantized activations in)")
    ap.add_argument("--dtype", default="int8", choices=["int8", "fp8"])
    ap.add_argument("--ms", default="")
    ap.add_argument("--prefill", action="store_true", help="tune the prefill row counts (513 ... 16384) instead of the decode ones")
    ap.add_argument("--forms", default="rope,silu,plain,partial")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--out", default="")
    ap.add_argument("--merge", action="store_true", help="keep the entries of other shapes already in --out")
    ap.add_argument("--log", default="")
    args = ap.parse_args()
    lib = Q.load()
    lib.ql_gemm_plan_clear()
    Q.workspace.reserve("cuda", 512 << 20)
    if args.prefill:
        args.iters = min(args.iters, 8)
    dt = torch.int8 if args.dtype == "int8" else torch.float8_e4m3fn
    Hq, Hkv, H, I = MODELS[args.model]
    gbit = {"row": 0, "group": Q.QL_PLAN_GROUP, "w8a8": Q.QL_PLAN_W8A8}[args.mode]
    w8a8 = args.mode == "w8a8"
    ms = [int(v) for v in args.ms.split(",")] if args.ms else (PREFILL_MS if args.prefill else MS)
    first_row = 513 if args.prefill else 1
    forms = [f for f in args.forms.split(",") if not (args.prefill and f == "partial")]
    if w8a8:
        forms = [f for f in forms if f != "partial"]     # (the deferred W8A8 form shares the W8A16 entries)

    def act(M, K):
        """The activation operand of a launch: bf16 rows, or (qx, sx) for the W8A8 kernels."""
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        return NQ.per_row_quant(x, dt) if w8a8 else x
    plans, log = [], []
    t_start = time.time()

    def record(form, name, N, K, M, lo, hi, res, top, extra=None):
        base = res[None]
        best = top[0] if top else None
        gain = 1.0 - res[best] / base if best is not None else 0.0
        row = dict(shape=name, form=form, N=N, K=K, M=M, auto_us=round(base, 2), best=best,
                   best_us=round(res[best], 2) if best is not None else None, gain=round(gain, 3))
        if extra:
            row.update(extra)
        log.append(row)
        print(json.dumps(row), flush=True)
        if best is not None and gain >= MIN_GAIN:
            plans.append(dict(form=form, N=N, K=K, m_lo=lo, m_hi=hi, tile=list(best[0]), split=best[1], shape=name,
                              model=args.model, us=round(res[best], 2), heuristic_us=round(base, 2)))

    def copies(N, K):
        return max(2, int(600e6 // (N * K)) + 1)

    # ---- qkv projection with its epilogue
    if "rope" in forms:
        N, K = (Hq + 2 * Hkv) * 128, H
        nc = min(copies(N, K), 40)
        mods = [make_mod(N, K, args.mode, dt) for _ in range(nc)]
        cs = build_cos_sin(128, 4096, 1e6, "cuda")
        nw = torch.ones(128, device="cuda", dtype=torch.bfloat16)
        kc = torch.zeros(64, 256, Hkv, 128, device="cuda", dtype=torch.bfloat16)
        vc = torch.zeros_like(kc)
        for M, lo, hi in ranges(ms, first_row):
            x = act(M, K)
            pos = torch.randint(0, 2000, (M,), device="cuda")
            rows = cs.index_select(0, pos)
            slots = torch.randperm(64 * 256, device="cuda")[:M].to(torch.int32)
            if w8a8:
                fn = lambda i: NQ.qkv_rope_matmul(mods[i], None, None, rows, nw, nw, 1e-6, Hq, Hkv, 128, kc, vc, slots, False, quantized=x)  # noqa: E731
            else:
                fn = lambda i: NQ.qkv_rope_matmul(mods[i], x, None, rows, nw, nw, 1e-6, Hq, Hkv, 128, kc, vc, slots, False)  # noqa: E731
            cands = [None] + [(t, s) for t in ROPE_TILES for s in (1, 2, 4) if not (t[0] > 1 and t[0] * 32 >= 2 * max(M, 32))
                              and s <= max(1, K // (32 * t[3] * t[2]))]
            if not w8a8:
                cands += [(t, -g) for t in SK_ROPE_TILES for g in sk_grids(t) if not (t[0] > 1 and t[0] * 32 >= 2 * max(M, 32))]
            if args.prefill:
                cands = [None] + [(t, 1) for t in PREFILL_ROPE_TILES]
            res, top = tune_one(lib, fn, nc, cands, args.iters)
            record(gbit | Q.QL_PLAN_ROPE, "qkv", N, K, M, lo, hi, res, top)
        del mods

    # ---- gate/up with the SiLU epilogue
    if "silu" in forms:
        N, K = 2 * I, H
        nc = min(copies(N, K), 40)
        mods = [make_mod(N, K, args.mode, dt, silu=True) for _ in range(nc)]
        for M, lo, hi in ranges(ms, first_row):
            x = act(M, K)
            fn = (lambda i: mods[i].forward_quantized(*x)) if w8a8 else (lambda i: mods[i](x))  # noqa: E731
            cands = [None] + [(t, 1) for t in (PREFILL_TILES if args.prefill else TILES)
                              if t[1] % 2 == 0 and not (t[0] > 1 and t[0] * 32 >= 2 * max(M, 32))
                              and not (args.mode == "group" and t[0] == 8)]
            if not args.prefill:
                # K split across workgroups (the tile's last workgroup runs the gate/up exchange), and the stream-K schedule
                cands += [(t, sp) for t in TILES for sp in (2, 4) if t[1] % 2 == 0 and not (t[0] > 1 and t[0] * 32 >= 2 * max(M, 32))
                          and not (args.mode == "group" and t[0] == 8) and sp <= max(1, K // (32 * t[3] * t[2]))]
                if not w8a8:
                    cands += [(t, -g) for t in SK_TILES for g in sk_grids(t) if t[1] % 2 == 0
                              and not (t[0] > 1 and t[0] * 32 >= 2 * max(M, 32)) and not (args.mode == "group" and t[0] == 8)]
            res, top = tune_one(lib, fn, nc, cands, args.iters)
            record(gbit | Q.QL_PLAN_SILU, "gate_up", N, K, M, lo, hi, res, top)
        del mods

    # ---- o_proj / down_proj: self-contained, and deferred into the residual add + RMSNorm
    for name, N, K in (("o", H, Hq * 128), ("down", H, I)):
        if "plain" not in forms and "partial" not in forms:
            break
        nc = min(copies(N, K), 60)
        mods = [make_mod(N, K, args.mode, dt) for _ in range(nc)]
        nw = torch.ones(N, device="cuda", dtype=torch.bfloat16)
        for M, lo, hi in ranges(ms, first_row):
            x = act(M, K)
            resid = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
            if w8a8:
                plain_fn = lambda i: ops.add_rmsnorm(mods[i].forward_quantized(*x), resid, nw, 1e-6)  # noqa: E731
            else:
                plain_fn = lambda i: ops.add_rmsnorm(mods[i](x), resid, nw, 1e-6)  # noqa: E731
            cands = [None] + [(t, s) for t in TILES for s in (1, 2, 4, 8) if not (t[0] > 1 and t[0] * 32 >= 2 * max(M, 32))
                              and not (args.mode == "group" and t[0] == 8) and s <= max(1, K // (32 * t[3] * t[2]))]
            if not w8a8:
                cands += [(t, -g) for t in SK_TILES for g in sk_grids(t) if not (t[0] > 1 and t[0] * 32 >= 2 * max(M, 32))
                          and not (args.mode == "group" and t[0] == 8)]
            if args.prefill:
                cands = [None] + [(t, 1) for t in PREFILL_TILES if not (args.mode == "group" and t[0] == 8)]
            res, top = tune_one(lib, plain_fn, nc, cands, args.iters)
            if "plain" in forms:
                record(gbit, name, N, K, M, lo, hi, res, top)
            if "partial" not in forms:
                continue
            plain_best = min(res.values())
            # the deferred form as the engine would run it now (heuristic: it may decline, i.e. one slice)
            S0 = lib.ql_gemm_partial_slices(M, N, K, Q.QL_PER_GROUP if args.mode == "group" else Q.QL_PER_ROW)
            part_fn = lambda i: ops.add_rmsnorm(NQ.w8a16_partial_matmul(mods[i], x), resid, nw, 1e-6)  # noqa: E731
            engine_now = BK.time_fn(part_fn, nc, args.iters * 3) if S0 > 1 else res[None]
            pc = [(t, s) for t in PARTIAL_TILES for s in (2, 3, 4, 6, 8, 12, 16) if s <= max(1, K // (32 * t[3] * t[2]))
                  and not (t[0] > 1 and t[0] * 32 >= 2 * max(M, 32))]
            pres, ptop = tune_one(lib, part_fn, nc, [c for c in pc], args.iters) if pc else ({}, [])
            pbest = ptop[0] if ptop else None
            row = dict(shape=name, form="partial", N=N, K=K, M=M, engine_now_us=round(engine_now, 2), heuristic_slices=S0,
                       plain_best_us=round(plain_best, 2), partial_best=pbest,
                       partial_best_us=round(pres[pbest], 2) if pbest else None)
            log.append(row)
            print(json.dumps(row), flush=True)
            form = gbit | Q.QL_PLAN_PARTIAL
            if pbest is not None and pres[pbest] < min(plain_best, engine_now) * (1.0 - MIN_GAIN):
                plans.append(dict(form=form, N=N, K=K, m_lo=lo, m_hi=hi, tile=list(pbest[0]), split=pbest[1], shape=name,
                                  model=args.model, us=round(pres[pbest], 2), heuristic_us=round(engine_now, 2)))
            elif S0 > 1 and plain_best < engine_now * (1.0 - MIN_GAIN):
                # deferring loses here: one slice = the caller keeps the self-contained GEMM
                t = top[0][0] if top else (1, 1, 8, 4)
                plans.append(dict(form=form, N=N, K=K, m_lo=lo, m_hi=hi, tile=list(t), split=1, shape=name, model=args.model,
                                  us=round(plain_best, 2), heuristic_us=round(engine_now, 2)))
        del mods

    print(f"# {len(plans)} plan entries in {time.time() - t_start:.0f} s", flush=True)
    if args.log:
        with open(args.log, "w") as f:
            for r in log:
                f.write(json.dumps(r) + "\n")
    if args.out:
        doc = {"plans": []}
        if args.merge and os.path.exists(args.out):
            doc = json.load(open(args.out))
        mine = {(p["form"], p["N"], p["K"]) for p in plans} | {
            (f, n, k) for f in range(32) for (n, k) in [((Hq + 2 * Hkv) * 128, H), (2 * I, H), (H, Hq * 128), (H, I)]
            if (f & (Q.QL_PLAN_GROUP | Q.QL_PLAN_W8A8)) == gbit}
        doc["plans"] = [p for p in doc.get("plans", []) if (p["form"], p["N"], p["K"]) not in mine] + plans
        doc["device"] = torch.cuda.get_device_name(0)
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=0)
            f.write("\n")


if __name__ == "__main__":
    main()
