"""Merge plan files written by tools/autotune_plans.py (one per model and mode, e.g. brought back from the GPU box under
gpurun_out/) into the table the library loads at start-up.  Where two files hold entries of one (form, N, K), the
later file replaces the earlier file's entries inside the row span it tuned.

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
        # a later file wins where row ranges collide: its tuner run covered the range its entries span, so earlier
        # entries of the same (form, N, K) inside that span go, those outside it (e.g. decode sizes next to a
        # prefill-size run) stay
        span = {}
        for p in doc["plans"]:
            key = (p["form"], p["N"], p["K"])
            lo, hi = span.get(key, (p["m_lo"], p["m_hi"]))
            span[key] = (min(lo, p["m_lo"]), max(hi, p["m_hi"]))
        for key, (lo, hi) in span.items():
            by_shape[key] = [q for q in by_shape.get(key, []) if q["m_hi"] < lo or q["m_lo"] > hi]
        for p in doc["plans"]:
            by_shape[(p["form"], p["N"], p["K"])].append(p)
    plans = [p for key in sorted(by_shape) for p in sorted(by_shape[key], key=lambda p: p["m_lo"])]
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        f.write('{"device": %s,\n "plans": [\n' % json.dumps(device))
        f.write(",\n".join("  " + json.dumps(p) for p in plans))
        f.write("\n ]}\n")
    print(f"{len(plans)} entries for {len(by_shape)} (form, N, K) keys -> {args.out}")


if __name__ == "__main__":
    main()
