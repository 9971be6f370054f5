"""GPU-busy share of the bench's timed pass from a rocprofv3 kernel trace (diagnostic).

usage: python tools/gpu_busy.py <kernel_trace.csv> <ms_per_step from the bench line>
The timed pass is the last `ms_per_step` of kernel activity when the bench ran without its roofline legs."""
import csv
import sys
from collections import defaultdict

path, ms = sys.argv[1], float(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = max(e for _, e, _ in rows)
t0 = t_end - int(ms * 1e6)
busy = 0
last = t0
by = defaultdict(int)
gaps = []
for s, e, n in rows:
    if e <= t0:
        continue
    s2 = max(s, last)
    if s > last:
        gaps.append(s - last)
    if e > s2:
        busy += e - s2
        by[n.split("(")[0][:90]] += e - s2
        last = e
win = t_end - t0
print(f"window {win/1e9:.3f} s, busy {busy/1e9:.3f} s = {busy/win:.3f}; kernels in window {sum(1 for s,e,n in rows if e>t0)}")
gaps.sort()
n = len(gaps)
print(f"gaps: n {n}, sum {sum(gaps)/1e9:.3f} s, median {gaps[n//2]/1e3:.2f} us, p90 {gaps[int(n*.9)]/1e3:.2f} us, p99 {gaps[int(n*.99)]/1e3:.2f} us, max {gaps[-1]/1e6:.2f} ms")
big = [g for g in gaps if g > 20e3]
print(f"gaps > 20 us: {len(big)}, sum {sum(big)/1e9:.3f} s;  gaps > 200 us: {len([g for g in gaps if g > 200e3])}, sum {sum(g for g in gaps if g > 200e3)/1e9:.3f} s")
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v/1e9:.3f} s  {k}")
