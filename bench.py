"""Headline benchmark: tokens/s of Qwen3-0.6B with INT8 weight-only (W8A16, per-output-channel
scales) quantized linears on the reference bench.py's synthetic 256-sequence workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over one batch of synthetic input = one generate() of the
256 sequences (random token prompts of 100-1024 tokens, 100-1024 new tokens each, seed 0: exactly
the reference bench.py's workload; sum(max_tokens) = 133,966 output tokens per step).  Weights are
random-initialised (no checkpoint on the box); data: synthetic.  With N > 1 the model runs tensor-
parallel over N GPUs (one process per GPU, RCCL all-reduce in the row-parallel linears): the total
work is fixed, so scaling is "strong".

Rank 0 prints one JSON line with the metric, a `roofline` object for the kernel of the hot path (the decode
W8A16 GEMM, HBM-bound: algorithmic bytes / launch duration measured with HIP events), `roofline_prefill`
(the same kernel at prefill size, MFMA-bound), `roofline_attention` (decode attention, the largest share of
the bench's GPU time, HBM-bound) and a `cpu_baseline` object (the reference's CPU arithmetic for the same linears, timed on host cores).
"""
import argparse
import json
import os
import sys
import tempfile
import time
from random import randint, seed

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
sys.path.insert(0, ROOT)

QWEN3_06B = dict(architectures=["Qwen3ForCausalLM"], hidden_size=1024, num_hidden_layers=28, num_attention_heads=16,
                 num_key_value_heads=8, head_dim=128, intermediate_size=3072, vocab_size=151936, rms_norm_eps=1e-6,
                 rope_theta=1000000.0, max_position_embeddings=40960, tie_word_embeddings=True, attention_bias=False,
                 hidden_act="silu", torch_dtype="bfloat16", eos_token_id=151645)
# the other models BASELINE.json's configurations name (experiments: `--model 4b --quant per_group`); the headline
# line is always Qwen3-0.6B
QWEN3_4B = dict(QWEN3_06B, hidden_size=2560, num_hidden_layers=36, num_attention_heads=32, intermediate_size=9728)
QWEN3_8B = dict(QWEN3_06B, hidden_size=4096, num_hidden_layers=36, num_attention_heads=32, intermediate_size=12288,
                tie_word_embeddings=False)
MODELS = {"0.6b": ("Qwen3-0.6B", QWEN3_06B), "4b": ("Qwen3-4B", QWEN3_4B), "8b": ("Qwen3-8B", QWEN3_8B)}
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak


def workload():
    """The reference bench.py's inputs (bench.py:9-18 there)."""
    seed(0)
    prompts = [[randint(0, 10000) for _ in range(randint(100, 1024))] for _ in range(256)]
    max_tokens = [randint(100, 1024) for _ in range(256)]
    return prompts, max_tokens


def gemm_bytes(M, N, K, n_out=None):
    """Algorithmic bytes of one W8A16 linear call: 1-byte weights + f32 row scales + bf16 x in + bf16 y out
    (n_out = N/2 columns leave the gate/up GEMM whose epilogue applies SiLU * mul)."""
    return N * K + 4 * N + 2 * M * K + 2 * M * (N if n_out is None else n_out)


TRAFFIC_TABLE = os.path.join(ROOT, "profiles", "r03_decode_gemm_traffic.json")


def pmc_traffic(mods4, buckets):
    """HBM-side bytes per launch, weighted over the same (batch bucket, layer shape) mix as `achieved`, from
    the committed rocprofv3 PMC passes (FETCH_SIZE doubled as gfx950 requires, + WRITE_SIZE; see
    tools/pmc_traffic.py).  Counters cannot be read from inside this process, hence the table; null when
    the table is missing, was taken on other kernel sources than this tree's, or the model is not the one it was
    measured on."""
    if not os.path.exists(TRAFFIC_TABLE):
        return None
    doc = json.load(open(TRAFFIC_TABLE))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_traffic import kernel_fingerprint
    if doc.get("kernel_sources_sha256") != kernel_fingerprint():
        return None                  # measured on other kernel sources than the ones this tree builds: stale
    table = doc["traffic_bytes"]
    names = ["qkv", "o", "gate_up_silu", "down"]
    want = {"qkv": (4096, 1024), "o": (1024, 2048), "gate_up_silu": (6144, 1024), "down": (1024, 3072)}
    for nm, m in zip(names, mods4):
        if tuple(m.qw.shape) != want[nm] or (nm == "gate_up_silu") != bool(getattr(m, "silu_fused", False)):
            return None
    tot = n = 0
    for M, cnt in buckets.items():
        for nm in names:
            keys = sorted(int(k) for k in table[nm])
            k = min(keys, key=lambda v: (abs(v - M), v))
            tot += table[nm][str(k)] * cnt
            n += cnt
    return int(tot / n) if n else None


def _graph_time(fn, side):
    """Seconds for one replay of fn() captured into a hipGraph on `side` (best of 3, HIP events on that stream)."""
    g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            fn()
        g.replay()
        side.synchronize()
        best = None
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(side)
            g.replay()
            e.record(side)
            side.synchronize()
            t = s.elapsed_time(e) * 1e-3
            best = t if best is None else min(best, t)
    del g
    return best


MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp8": 5000.0, "int8": 5000.0}    # dense peaks of the MI355X matrix cores


def measure_decode_gemm(llm, step_batches, model_key="0.6b"):
    """Average duration of the decode GEMM launches with HIP events on the launch stream, over the
    (batch bucket, layer shape) mix the timed region actually ran; returns the roofline object.

    The linear launches of one decode step (4 per layer) are issued exactly as the decode graph issues them: the qkv
    projection with its q/k norm + RoPE + KV-store epilogue, gate/up with the SiLU epilogue, o_proj / down_proj
    with their split-K reduction deferred to the residual add + RMSNorm launch where the plan does that; W8A8 modules
    take the quantized activations the preceding norm / attention launch hands them (down_proj quantizes its own
    input, a launch that is part of the timed graph).  What the deferral moves into the norm launches is charged to the
    GEMMs: the norms that finish a deferred GEMM are part of the timed graph, and a second graph with the same number
    of plain norms is subtracted.

    The bound of a launch is the larger of its algorithmic bytes / HBM peak and its flops / MFMA peak (the 200+ row
    buckets of the weight-only kernels sit on the MFMA side of the ridge); `frac_of_roofline` = sum of those bounds /
    measured time.  `bound` names the side that holds more of that sum, `achieved` / `peak` / `frac` are on that side."""
    from nanovllm import ops
    from nanovllm.utils.quantization import PartialSum, QLinear, _QBase, per_row_quant, qkv_rope_matmul
    runner = llm.model_runner
    buckets = {}
    for bs in step_batches:
        g = next(x for x in runner.graph_bs if x >= bs) if not runner.enforce_eager else bs
        buckets[g] = buckets.get(g, 0) + 1
    layers = runner.model.model.layers
    mods = [[l.self_attn.qkv_proj, l.self_attn.o_proj, l.mlp.gate_up_proj, l.mlp.down_proj] for l in layers]
    if not isinstance(mods[0][0], _QBase):
        return None
    w8a8 = isinstance(mods[0][0], QLinear) and not mods[0][0].weight_only
    qdt = mods[0][0].dtype
    group = getattr(mods[0][0], "group_size", 0) or 0
    mfma_peak = MFMA_PEAK_TFLOPS["bf16"] if not w8a8 else MFMA_PEAK_TFLOPS["fp8" if qdt == torch.float8_e4m3fn else "int8"]
    hidden = mods[0][1].qw.shape[0]
    nw = torch.ones(hidden, device="cuda", dtype=torch.bfloat16)
    tot_t = tot_b = tot_n = tot_plain = tot_raw = tot_rope = tot_f = tot_roof = roof_hbm = roof_mfma = 0.0
    per_bucket = []
    side = torch.cuda.Stream()
    with torch.inference_mode():
        for M, cnt in sorted(buckets.items()):
            xs = [torch.randn(M, m.qw.shape[1], device="cuda", dtype=torch.bfloat16) for m in mods[0]]
            qs = [per_row_quant(x, qdt) for x in xs] if w8a8 else None
            res = torch.randn(M, hidden, device="cuda", dtype=torch.bfloat16)
            pos = torch.randint(0, 2000, (M,), device="cuda")
            slots = torch.arange(M, device="cuda", dtype=torch.int32)
            deferred = [0]

            def gemms(with_norms=True):
                for li, m4 in enumerate(mods):
                    at = layers[li].self_attn
                    qkv_rope_matmul(m4[0], xs[0], pos, at.cos_sin, at.q_norm.weight, at.k_norm.weight, at.eps, at.num_heads,
                                    at.num_kv_heads, at.head_dim, at.k_cache, at.v_cache, slots, want_kv=False,
                                    quantized=qs[0] if w8a8 else None)
                    for k in (1, 3):
                        if w8a8:      # o_proj: quantized by the attention merge launch; down_proj: its own launch
                            y = m4[k].forward_partial_quantized(*(qs[1] if k == 1 else per_row_quant(xs[3], qdt)))
                        else:
                            y = m4[k].forward_partial(xs[k])
                        if isinstance(y, PartialSum) and with_norms:
                            ops.add_rmsnorm(y, res, nw, 1e-6)
                            deferred[0] += 1
                    if w8a8:
                        m4[2].forward_quantized(*qs[2])
                    else:
                        m4[2](xs[2])

            def norms():
                for _ in range(deferred[0]):
                    ops.add_rmsnorm(res, res, nw, 1e-6)

            qkv_out = torch.randn(M, mods[0][0].qw.shape[0], device="cuda", dtype=torch.bfloat16)

            def ropes():                # the stand-alone operator the qkv epilogue replaces, once per layer
                for li in range(len(mods)):
                    at = layers[li].self_attn
                    ops.qknorm_rope_store(qkv_out, pos, at.cos_sin, at.q_norm.weight, at.k_norm.weight, at.eps, at.num_heads,
                                          at.num_kv_heads, at.head_dim, at.k_cache, at.v_cache, slots, want_kv=False)
            def plain():                # the self-contained form of every launch (every launch with its own reduction and a separate RoPE operator)
                for m4 in mods:
                    for x, m in zip(xs, m4):
                        m(x)
            gemms()                     # untimed: touches the code path for this M
            plain()
            torch.cuda.synchronize()
            deferred[0] = 0
            t1 = _graph_time(gemms, side)
            t0 = _graph_time(norms, side) if deferred[0] else 0.0
            if model_key == "0.6b":     # (the other accountings of the headline object; the larger models skip them)
                tot_plain += _graph_time(plain, side) * cnt
                tot_raw += _graph_time(lambda: gemms(False), side) * cnt
                tot_rope += _graph_time(ropes, side) * cnt
            nbytes = nflop = roof = 0.0
            for m4 in mods:
                for m in m4:
                    N, K = m.qw.shape
                    b = gemm_bytes(M, N, K, N // 2 if getattr(m, "silu_fused", False) else None)
                    if group:
                        b += 4 * N * (K // group) - 4 * N           # [N, K/g] group scales instead of [N] row scales
                    if w8a8:
                        b -= M * K                                  # 1-byte activations in
                    f = 2.0 * M * N * K
                    tb, tf = b / (HBM_PEAK_GBPS * 1e9), f / (mfma_peak * 1e12)
                    nbytes += b
                    nflop += f
                    roof += max(tb, tf)
                    if tb >= tf:
                        roof_hbm += tb * cnt
                    else:
                        roof_mfma += tf * cnt
            nlaunch = sum(len(m4) for m4 in mods)
            tot_t += (t1 - t0) * cnt
            tot_b += nbytes * cnt
            tot_f += nflop * cnt
            tot_roof += roof * cnt
            tot_n += nlaunch * cnt
            per_bucket.append({"rows": M, "steps": cnt, "us_per_launch": round((t1 - t0) / nlaunch * 1e6, 2),
                               "frac_of_roofline": round(roof / (t1 - t0), 4)})
            del xs, qs, res, qkv_out
    if tot_n == 0:
        return None
    avg_us = tot_t / tot_n * 1e6
    hbm_side = roof_hbm >= roof_mfma
    gbps, tflops = tot_b / tot_t / 1e9, tot_f / tot_t / 1e12
    achieved, peak, unit = (gbps, HBM_PEAK_GBPS, "GB/s") if hbm_side else (tflops, mfma_peak, "TFLOP/s")
    kind = "W8A8" if w8a8 else "W8A16"
    out = {"bound": "hbm" if hbm_side else "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": unit,
           "frac": round(achieved / peak, 4),
           "traffic": pmc_traffic(mods[0], buckets) if model_key == "0.6b" and not w8a8 and not group else None,
           "kernel": f"ql::qgemm_kernel (decode {kind} GEMM, all four layer shapes as the decode graph launches them, "
                     "M = graph bucket)",
           "avg_launch_us": round(avg_us, 2),
           # per launch: max(bytes / HBM peak, flops / MFMA peak) summed over the mix / measured time; and the two
           # sides separately (what share of that bound sits on launches whose larger term is the byte one)
           "frac_of_roofline": round(tot_roof / tot_t, 4),
           "roofline_share_hbm_bound_launches": round(roof_hbm / (roof_hbm + roof_mfma), 3),
           "achieved_GBps": round(gbps, 1), "achieved_TFLOPs": round(tflops, 1),
           "avg_algorithmic_bytes": int(tot_b / tot_n), "avg_flop": int(tot_f / tot_n),
           "launches_weighted": int(tot_n), "buckets": per_bucket}
    if model_key == "0.6b":
        # the same launches without the norms that finish the deferred reductions (what a kernel trace shows
        # for these kernels, plus the launch gap), their self-contained form (every launch with its own reduction and a separate RoPE operator), and
        # the launches' cost net of the stand-alone RoPE operator their epilogue replaced
        out.update({"avg_launch_us_gemm_launches_only": round(tot_raw / tot_n * 1e6, 2),
                    "avg_launch_us_self_contained_form": round(tot_plain / tot_n * 1e6, 2),
                    "avg_launch_us_net_of_replaced_rope_launch": round((tot_t - tot_rope) / tot_n * 1e6, 2)})
    return out


def measure_prefill_gemm(llm, M=16384):
    """MFMA-bound side of the same kernel: the four layer GEMMs at a prefill-sized batch (M = the scheduler's
    max_num_batched_tokens), HIP events around 20 back-to-back rounds after 40 untimed ones (the leg follows the
    latency-bound decode legs, and the first ~10 ms of MFMA-bound work run at the clocks those left behind: a
    5-round measurement read 5-10 % low); achieved = 2*M*N*K flop / time against the dense bf16 MFMA peak (the W8A16
    kernel contracts on bf16 MFMA after widening the weights)."""
    from nanovllm.utils.quantization import _QBase
    l0 = llm.model_runner.model.model.layers[0]
    mods = [l0.self_attn.qkv_proj, l0.self_attn.o_proj, l0.mlp.gate_up_proj, l0.mlp.down_proj]
    if not isinstance(mods[0], _QBase):
        return None
    with torch.inference_mode():
        xs = [torch.randn(M, m.qw.shape[1], device="cuda", dtype=torch.bfloat16) for m in mods]
        for _ in range(40):
            for x, m in zip(xs, mods):
                m(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            for x, m in zip(xs, mods):
                m(x)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) * 1e-3 / 20
        # the same-box ceiling this figure is read against: the vendor library's bf16 x bf16 GEMM (F.linear ->
        # hipBLASLt) at the same four shapes and row count, same warm-up, no dequantization and no epilogue
        ws = [(torch.randn(m.qw.shape[0], m.qw.shape[1], device="cuda") * 0.02).to(torch.bfloat16) for m in mods]
        for _ in range(40):
            for x, w in zip(xs, ws):
                torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        s.record()
        for _ in range(20):
            for x, w in zip(xs, ws):
                torch.nn.functional.linear(x, w)
        e.record()
        torch.cuda.synchronize()
        t_lib = s.elapsed_time(e) * 1e-3 / 20
        del ws
    flop = sum(2.0 * M * m.qw.shape[0] * m.qw.shape[1] for m in mods)
    tf = flop / t / 1e12
    # the W8A16 kernels contract on bf16 MFMA (2.5 PFLOP/s dense); the fp8 W8A8 kernel on the K = 64 block-scaled fp8
    # instruction (5 PFLOP/s dense); the timed region of a W8A8 module includes its activation-quantization launch
    w8a8 = not getattr(mods[0], "weight_only", True)
    peak = 5000.0 if (w8a8 and mods[0].dtype == torch.float8_e4m3fn) else 2500.0
    form = "W8A8, activation quantization launch included" if w8a8 else "W8A16"
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
            "traffic": None, "kernel": f"ql::qgemm_kernel (prefill {form} GEMM, four layer shapes, M = {M})",
            "hipblaslt_bf16_tflops": round(flop / t_lib / 1e12, 1), "vs_hipblaslt_bf16": round(t_lib / t, 3)}


def measure_decode_attention(llm):
    """HBM-bound neighbour of the GEMMs, and the largest share of the bench's GPU time: the decode attention launch
    (ql_paged_attn_decode + its merge launch) on a mid-run state of the workload - 131 running sequences (the mean
    over the 1023 decode steps) with 300..1800 cached tokens each - over 8 distinct caches so that K/V stream from
    HBM, captured into a hipGraph and timed with HIP events on the launch stream.  achieved = K/V bytes / time."""
    from nanovllm import ops
    hf = llm.model_runner.config.hf_config
    Hq, Hkv, D, bs = hf.num_attention_heads, hf.num_key_value_heads, hf.head_dim, llm.model_runner.block_size
    g = torch.Generator().manual_seed(0)
    B, L = 131, 8
    lens = torch.randint(300, 1800, (B,), generator=g)
    nblk = int(((lens + bs - 1) // bs).sum()) + 1
    with torch.inference_mode():
        kc = [torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16) for _ in range(L)]
        vc = [torch.randn(nblk, bs, Hkv, D, device="cuda", dtype=torch.bfloat16) for _ in range(L)]
        tables = torch.zeros(B, 16, dtype=torch.int32)
        nxt = 1
        for b in range(B):
            for j in range((int(lens[b]) + bs - 1) // bs):
                tables[b, j] = nxt
                nxt += 1
        tables = tables.cuda()
        q = torch.randn(B, Hq, D, device="cuda", dtype=torch.bfloat16)
        ctx = lens.to(torch.int32).cuda()
        ops.reserve_attn_workspace(B, Hq, Hkv, bs, 16)

        def run():
            for _ in range(4):
                for i in range(L):
                    ops.paged_attn_decode(q, kc[i], vc[i], tables, ctx, D ** -0.5)
        run()
        torch.cuda.synchronize()
        t = _graph_time(run, torch.cuda.Stream()) / (4 * L)
    nbytes = float(lens.sum()) * Hkv * D * 2 * 2
    gbps = nbytes / t / 1e9
    return {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
            "traffic": None, "kernel": "ql::paged_attn_decode_kernel + merge launch (131 sequences, 300..1800 cached tokens, "
                                       f"{Hq} query / {Hkv} kv heads)",
            "avg_launch_us": round(t * 1e6, 1), "avg_algorithmic_bytes": int(nbytes)}


def cpu_baseline():
    """The reference's CPU-runnable forward arithmetic for the same linears (x.float() @ (qw.float()*sw).T,
    torch.matmul) on this box's host cores: all 112 quantized linears of one Qwen3-0.6B decode step at
    M = 128 rows.  A reported baseline only."""
    from oracle.qlinear_oracle import cpu_reference_linear_torch
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    M = 128
    shapes = [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072)]
    g = torch.Generator().manual_seed(0)
    ws = [(torch.randint(-127, 128, (n, k), dtype=torch.int8, generator=g), torch.rand(n, generator=g) * 1e-3 + 1e-4)
          for n, k in shapes]
    xs = [torch.randn(M, k, generator=g).to(torch.bfloat16) for _, k in shapes]
    for (q, s), x in zip(ws, xs):
        cpu_reference_linear_torch(x, q, s)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 10.0:
        for _ in range(28):
            for (q, s), x in zip(ws, xs):
                cpu_reference_linear_torch(x, q, s)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(M / dt, 1), "unit": "tokens/s (quantized linears of one decode step only)", "cores": threads,
            "kind": "port",
            "sample": f"{reps} x [28 layers x 4 Qwen3-0.6B linears at M=128]: x.float() @ (qw.float()*sw).T via torch.matmul, "
                      f"{threads} threads, ~10 s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--quant", default="per_row_weight_only",
                    choices=["per_row_weight_only", "per_row", "per_group", "per_tensor", "bf16"])
    ap.add_argument("--dtype", default="int8", choices=["int8", "fp8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the FP8 W8A8 and Qwen3-4B g128 passes that follow the headline pass")
    ap.add_argument("--quant-lm-head", action="store_true", help="opt-in int8 LM head (not the reference's numerics)")
    ap.add_argument("--max-model-len", type=int, default=4096, help="4096 = the reference bench; other values are experiments")
    ap.add_argument("--model", default="0.6b", choices=sorted(MODELS),
                    help="0.6b = the headline benchmark; the larger models are experiments on the same workload")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank path on fewer GPUs than ranks (ranks share devices)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU) before anything here
        # has touched the GPU, relay their output (rank 0 prints the JSON line) and exit with their status
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local if args.backend == "nccl" else local % torch.cuda.device_count())
    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    from nanovllm import LLM, SamplingParams
    prompts, max_tokens = workload()
    sps = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=m) for m in max_tokens]
    total_tokens = sum(max_tokens)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_config(model, quant, dtype, steps, warmup):
        """Build the engine for one configuration, run `warmup` short passes and `steps` timed passes of the workload;
        returns (llm, seconds per pass, decode batch sizes of one pass, decode mode, captured?)."""
        model_dir = tempfile.mkdtemp(prefix=f"qwen3_{model}_r{rank}_")
        with open(os.path.join(model_dir, "config.json"), "w") as f:
            json.dump(MODELS[model][1], f)
        kw = dict(enforce_eager=False, max_model_len=args.max_model_len, load_format="dummy", tensor_parallel_size=world)
        if quant != "bf16":
            kw.update(quant_type=quant, linear_dtype=torch.int8 if dtype == "int8" else torch.float8_e4m3fn, group_size=128)
        if args.quant_lm_head:
            kw.update(quantize_lm_head=True)
        decode_mode, graph = "hipGraph decode", True
        llm, capture_error = None, None
        try:
            llm = LLM(model_dir, **kw)
        except Exception as ex:  # noqa: BLE001
            if world == 1:
                raise
            # Tensor-parallel ranks only: a collective that this RCCL build cannot capture into a hipGraph fails the
            # capture on every rank alike; the curve is then measured with eager decode steps, and the line says so.
            capture_error = type(ex).__name__
            print(f"[rank {rank}] graph capture failed ({capture_error}: {ex}); falling back to eager decode",
                  file=sys.stderr, flush=True)
        if llm is None:
            # (outside the handler: the exception's traceback keeps the first engine - weights, KV cache - alive)
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            kw["enforce_eager"] = True
            decode_mode, graph = f"EAGER decode (graph capture failed: {capture_error})", False
            llm = LLM(model_dir, **kw)
        # record the decode batch size of every step of one pass (for the roofline mix)
        step_batches = []
        orig_launch = llm.model_runner.launch

        def logged_launch(seqs, is_prefill, prev=None):
            if not is_prefill:
                step_batches.append(len(seqs))
            return orig_launch(seqs, is_prefill, prev=prev)
        llm.model_runner.launch = logged_launch
        for _ in range(warmup):
            # same shapes as the timed pass but one tenth of the decode length: warms graphs and allocator
            llm.generate(prompts, [SamplingParams(ignore_eos=True, max_tokens=max(8, m // 10)) for m in max_tokens])
        step_batches.clear()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            llm.generate(prompts, sps)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return llm, dt / steps, step_batches[: len(step_batches) // max(steps, 1)], decode_mode, graph

    def workload_text(model_name, quant, dtype, decode_mode):
        return ("reference bench.py: 256 seqs, prompts 100-1024 random tokens, 100-1024 new tokens "
                f"(133,966 output tokens), {model_name} random-init, "
                f"quant={quant}/{dtype if quant != 'bf16' else 'none'}, {decode_mode}")

    def arith(quant, dtype):
        return "bf16" if quant in ("bf16", "per_row_weight_only", "per_group", "per_tensor") else dtype

    llm, spp, batches, decode_mode, graph = run_config(args.model, args.quant, args.dtype, args.steps, args.warmup)
    roof = roof_prefill = roof_attn = None
    # (one GPU only: under tensor parallelism the row-parallel modules' forward contains a collective, which one
    # rank must not issue alone)
    model_name = MODELS[args.model][0]
    if rank == 0 and world == 1 and not args.no_roofline and args.quant != "bf16":
        roof = measure_decode_gemm(llm, batches, args.model)
        if args.model == "0.6b":
            roof_prefill = measure_prefill_gemm(llm)
            roof_attn = measure_decode_attention(llm)
    line = None
    if rank == 0:
        line = {
            "metric": f"tokens/s {model_name} INT8, 256 seqs (bench.py)", "value": round(total_tokens / spp, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(spp * 1e3, 2), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": arith(args.quant, args.dtype),
            "data": "synthetic", "graph": graph,
            "config": {"workload": workload_text(model_name, args.quant, args.dtype, decode_mode),
                       "parallelism": f"tp{world}", "decode_steps": len(batches)},
            "roofline": roof, "roofline_prefill": roof_prefill, "roofline_attention": roof_attn,
        }

    # BASELINE's other single-GPU configurations on the same workload, one timed pass each after the headline's:
    # Qwen3-0.6B FP8-e4m3 weights + activations on the fp8 MFMA, and Qwen3-4B INT8 g128 weight-only
    extra = []
    headline = args.model == "0.6b" and args.quant == "per_row_weight_only" and args.dtype == "int8"
    if world == 1 and headline and not args.no_extra_configs:
        import gc
        for key, quant, dtype, label in (("0.6b", "per_row", "fp8", "Qwen3-0.6B FP8-e4m3 weight+activation (W8A8), TP=1"),
                                         ("4b", "per_group", "int8", "Qwen3-4B INT8 group-size-128 weight-only (W8A16-g128), TP=1")):
            del llm
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            llm, spp2, batches2, mode2, graph2 = run_config(key, quant, dtype, 1, 1)
            r2 = None if args.no_roofline else measure_decode_gemm(llm, batches2, key)
            extra.append({"config": label, "workload": workload_text(MODELS[key][0], quant, dtype, mode2),
                          "value": round(total_tokens / spp2, 1), "unit": "tokens/s", "steps": 1, "warmup": 1,
                          "ms_per_step": round(spp2 * 1e3, 2), "dtype": arith(quant, dtype), "graph": graph2,
                          "decode_steps": len(batches2), "roofline": r2})
    if rank == 0:
        line["configs"] = extra
        line["cpu_baseline"] = None if (world > 1 or args.no_cpu_baseline) else cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
