"""Merge plan files written by tools/autotune_plans.py (one per model and mode, e.g. brought back from the GPU box under
gpurun_out/) into the table the library loads at start-up.  Entries of one (form, N, K) come from ONE input file: a
later file replaces an earlier file's entries for the shapes it tuned.

    python tools/merge_plans.py gpurun_out/plans_*.json --out nano-vllm-hw3_amd/nanovllm/_plans/gfx950.json
"""
import argparse
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    by_shape, device = {}, None
    for path in args.files:
        doc = json.load(open(path))
        device = doc.get("device", device)
        mine = {}
        for p in doc["plans"]:
            mine.setdefault((p["form"], p["N"], p["K"]), []).append(p)
        by_shape.update(mine)
    plans = [p for key in sorted(by_shape) for p in sorted(by_shape[key], key=lambda p: p["m_lo"])]
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        f.write('{"device": %s,\n "plans": [\n' % json.dumps(device))
        f.write(",\n".join("  " + json.dumps(p) for p in plans))
        f.write("\n ]}\n")
    print(f"{len(plans)} entries for {len(by_shape)} (form, N, K) keys -> {args.out}")


if __name__ == "__main__":
    main()
