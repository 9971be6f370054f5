"""Where a decode step's wall time goes: host-side preparation vs GPU execution (diagnostic)."""
import json, os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, ROOT)
import bench
from nanovllm import LLM, SamplingParams

d = tempfile.mkdtemp()
json.dump(bench.QWEN3_06B, open(os.path.join(d, "config.json"), "w"))
llm = LLM(d, enforce_eager=False, max_model_len=4096, load_format="dummy", quant_type="per_row_weight_only", linear_dtype=torch.int8)
prompts, mt = bench.workload()
llm.generate(prompts, [SamplingParams(ignore_eos=True, max_tokens=max(8, m // 10)) for m in mt])
r = llm.model_runner
T = dict(sched=0.0, prep=0.0, gpu=0.0, post=0.0, n=0, prefill=0.0)
llm.pipeline_decode = False          # step by step, so that host and GPU time can be told apart
orig_launch = r.launch
@torch.inference_mode()
def launch(seqs, is_prefill, prev=None):
    if is_prefill:
        t0 = time.perf_counter(); h = orig_launch(seqs, True); r.wait(h); T["prefill"] += time.perf_counter() - t0; return h
    t0 = time.perf_counter()
    ids, pos, slots, ctx_lens, tables = r.prepare_decode(seqs)      # fills the pinned mirror + one H2D copy
    bs = ids.numel(); gbs = next(x for x in r.graph_bs if x >= bs); gv = r.graph_vars
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t1b = time.perf_counter()
    r.graphs[gbs].replay(); tok = gv["logits"][:bs].argmax(dim=-1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out = tok.tolist(); t3 = time.perf_counter()
    T["prep"] += t1 - t0; T["gpu"] += t2 - t1b; T["post"] += t3 - t2 + (t1b - t1); T["n"] += 1
    return dict(seqs=list(seqs), tokens=tok, host=None, event=None)
r.launch = launch
t0 = time.perf_counter()
orig_sched = llm.scheduler.schedule
def sched():
    a = time.perf_counter(); o = orig_sched(); T["sched"] += time.perf_counter() - a; return o
llm.scheduler.schedule = sched
orig_commit = llm.scheduler.commit
def commit(s, i, t):
    a = time.perf_counter(); o = orig_commit(s, i, t); T["sched"] += time.perf_counter() - a; return o
llm.scheduler.commit = commit
llm.generate(prompts, [SamplingParams(ignore_eos=True, max_tokens=m) for m in mt])
tot = time.perf_counter() - t0
print(json.dumps({k: round(v, 3) for k, v in T.items()}), "total", round(tot, 3))
