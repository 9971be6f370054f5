"""PMC target: launches the decode W8A16 GEMM at every (Qwen3-0.6B layer shape, batch bucket) REPS times in
a fixed order, so that a rocprofv3 --pmc pass over it can be mapped back to (shape, M) by dispatch order.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/pmc_traffic.py
    rocprofv3 --pmc WRITE_SIZE ...                                    (separate pass)
    python3 tools/pmc_traffic.py --parse <fetch_dir> <write_dir> > profiles/<name>.json
"""
import csv, glob, json, os, statistics, sys

SHAPES = [("qkv", 4096, 1024), ("o", 1024, 2048), ("gate_up_silu", 6144, 1024), ("down", 1024, 3072)]
MS = [1, 8, 16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 256]
REPS = 5


def run():
    """The four linears as the decode graph launches them: qkv with the q/k norm + RoPE + KV-store epilogue, gate/up
    with the SiLU epilogue, o_proj / down_proj with the split-K reduction deferred wherever the plan defers it."""
    import types
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))
    from nanovllm import _qlinear as Q
    from nanovllm.utils import quantization as NQ
    lib = Q.load()
    torch.manual_seed(0)
    cos_sin = torch.randn(4096, 128, device="cuda", dtype=torch.float32)
    nw = torch.ones(128, device="cuda", dtype=torch.bfloat16)
    kc = torch.zeros(2, 256, 8, 128, device="cuda", dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    for name, N, K in SHAPES:
        ws = []
        for _ in range(REPS):
            w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
            q, s = NQ.per_row_quant(w, torch.int8)
            if name.endswith("silu"):
                q, s = NQ.interleave_gate_up(q), NQ.interleave_gate_up(s)
            # the layout the engine's modules use, behind the attributes the launch helpers read
            ws.append(types.SimpleNamespace(qw=NQ.tile_weight(q), sw=s, dtype=torch.int8, _bias_for_rank=lambda: None))
        for M in MS:
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            pos = torch.randint(0, 4096, (M,), device="cuda")
            slots = torch.arange(M, device="cuda", dtype=torch.int32)
            for mod in ws:
                if name == "qkv":
                    NQ.qkv_rope_matmul(mod, x, pos, cos_sin, nw, nw, 1e-6, 16, 8, 128, kc, vc, slots, want_kv=False)
                elif name.endswith("silu"):
                    NQ.per_row_weight_only_matmul(x, mod.qw, mod.sw, None, torch.int8, silu=True)
                elif M <= NQ._QBase.PARTIAL_MAX_ROWS and lib.ql_gemm_partial_slices(M, N, K, Q.QL_PER_ROW) > 1:
                    NQ.w8a16_partial_matmul(mod, x)
                else:
                    NQ.per_row_weight_only_matmul(x, mod.qw, mod.sw, None, torch.int8)
                torch.cuda.synchronize()          # one dispatch at a time: the counter pass drops records otherwise


def kernel_fingerprint():
    """sha256 over the sources the decode GEMM is built from: a traffic table is only quoted for the kernels it
    was measured on (bench.py compares this with the tree it runs from)."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano-vllm-hw3_amd", "csrc")
    h = hashlib.sha256()
    for f in ("ql_common.h", "ql_rope.h", "ql_gemm_kernel.h", "ql_gemm.hip"):
        h.update(open(os.path.join(root, f), "rb").read())
    # ... and the measured plans the dispatch consults before its heuristics (which tile a launch runs)
    plans = os.path.join(os.path.dirname(root), "nanovllm", "_plans", "gfx950.json")
    if os.path.exists(plans):
        h.update(open(plans, "rb").read())
    return h.hexdigest()


def read(d, counter):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "qgemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    assert len(vals) == len(SHAPES) * len(MS) * REPS, (len(vals), counter)
    return vals


def parse(fetch_dir, write_dir):
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    out = {"note": "HBM-side bytes per launch of the decode W8A16 GEMM from rocprofv3 PMC passes (tools/pmc_traffic.py): "
                   "traffic = 2 * FETCH_SIZE (gfx950 reports half the bytes of 16-B-per-lane streaming reads) + WRITE_SIZE, "
                   "both in KiB units; median of 5 launches on distinct weight copies",
           "kernel_sources_sha256": kernel_fingerprint(), "traffic_bytes": {}}
    i = 0
    for name, N, K in SHAPES:
        out["traffic_bytes"][name] = {}
        for M in MS:
            f = statistics.median(fetch[i:i + REPS]) * 1024 * 2
            w = statistics.median(write[i:i + REPS]) * 1024
            out["traffic_bytes"][name][str(M)] = int(f + w)
            i += REPS
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        run()
