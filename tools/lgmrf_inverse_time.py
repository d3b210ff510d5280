"""Dev tool: the blocked Gauss-Jordan inversion's time end to end (first k_bgj_pivot0 start -> last k_bgj_update end) and
per kernel, from a rocprofv3 --kernel-trace CSV of tools/lgmrf_node_time.py.
usage: python tools/lgmrf_inverse_time.py <..._kernel_trace.csv>"""
import csv, sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
inv, cur, per = [], None, defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"]
    if "k_bgj" not in nm:
        continue
    short = nm.split("::")[-1].split("(")[0]
    per[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if "k_bgj_pivot0" in nm:
        cur = [int(r["Start_Timestamp"]), 0, 0]
        inv.append(cur)
    if cur:
        cur[1], cur[2] = int(r["End_Timestamp"]), cur[2] + 1
for c in inv:
    print(f"inversion: {(c[1] - c[0]) / 1e3:8.1f} us end to end, {c[2]} launches")
for k, v in per.items():
    print(f"{k:14s} {len(v):4d} launches  mean {sum(v) / len(v):7.2f} us  min {min(v):7.2f}  max {max(v):7.2f}")
