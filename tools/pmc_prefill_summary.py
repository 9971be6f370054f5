"""Text summary of the two --pmc passes over tools/pmc_prefill.py (tools/profile_round.sh): per prefill-GEMM dispatch
(the last launch of each shape) the raw counters and the derived fractions.

    python3 tools/pmc_prefill_summary.py gpurun_out/prof_<tag>/pf_sq gpurun_out/prof_<tag>/pf_lds
"""
import collections
import csv
import glob
import sys

SHAPE = {1048576: "qkv  N=4096 K=1024", 262144: "o/down N=1024 (K=2048 then K=3072)", 1572864: "gate_up N=6144 K=1024"}


def load(d):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "qgemm_kernel" not in r["Kernel_Name"]:
            continue
        key = (int(r["Dispatch_Id"]), int(r["Grid_Size"]), r["Kernel_Name"])
        per.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    # the last launch of every (grid, kernel) run of consecutive dispatches
    out, prev = [], None
    for key, c in per.items():
        if prev is not None and (prev[0][1], prev[0][2]) != (key[1], key[2]):
            out.append(prev)
        prev = (key, c)
    if prev is not None:
        out.append(prev)
    return out


for d in sys.argv[1:]:
    tag = d.rstrip("/").split("/")[-1]
    for (disp, grid, name), c in load(d):
        tile = name[name.index("<"):name.index(">") + 1] if "<" in name else name[:60]
        print(f"[{tag}] grid {grid} ({SHAPE.get(grid, '?')}) dispatch {disp} tile {tile}")
        for k in sorted(c):
            print(f"    {k:30s} {c[k]:14.0f}")
        if "SQ_WAVE_CYCLES" in c and "SQ_INSTS_MFMA" in c:
            w = c["SQ_WAVE_CYCLES"]
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines
            util = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (c["SQ_BUSY_CYCLES"] / 32)
            print(f"    -> parked (WAIT_ANY) {c['SQ_WAIT_ANY'] / w:.2f}, issue-stalled (WAIT_INST_ANY) {c['SQ_WAIT_INST_ANY'] / w:.2f}, "
                  f"issuing {c['SQ_ACTIVE_INST_ANY'] / w:.2f} of wave time; VALU/MFMA instructions "
                  f"{c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']:.1f}; MFMA pipe busy cycles per MFMA "
                  f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_MFMA']:.1f}; MFMA pipe busy {util:.2f} of the kernel's cycles")
