"""Launches of a rocprofv3 `--kernel-trace --output-format csv` run that cannot fill the chip:
per (kernel, grid) group with fewer than `maxwg` workgroups, the calls, average time and total per
step -- where a cheap re-partition pays (the split-K fold ran on 32 blocks: 0.34 ms per step).
usage: trace_underfilled.py <output dir> [steps in the trace=1] [maxwg=512] [min avg us=6]"""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
maxwg = int(sys.argv[3]) if len(sys.argv) > 3 else 512
minus = float(sys.argv[4]) if len(sys.argv) > 4 else 6.0
agg = {}
with open(f) as fh:
    for row in csv.DictReader(fh):
        name = re.sub(r"\(.*", "", row["Kernel_Name"])[:70]
        wg = 1
        for ax in "XYZ":
            g, w = int(row[f"Grid_Size_{ax}"]), max(int(row[f"Workgroup_Size_{ax}"]), 1)
            wg *= max(g // w, 1)
        thr = int(row["Workgroup_Size_X"]) * int(row["Workgroup_Size_Y"]) * int(row["Workgroup_Size_Z"])
        dt = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        a = agg.setdefault((name, wg, thr), [0, 0])
        a[0] += 1
        a[1] += dt
rows = [(k, v) for k, v in agg.items() if k[1] < maxwg and v[1] / v[0] / 1e3 >= minus]
rows.sort(key=lambda kv: -kv[1][1])
print(f"{'kernel':70s} {'blocks':>7s} {'thr':>5s} {'calls':>6s} {'avg_us':>8s} {'ms/step':>8s}")
tot = 0.0
for (name, wg, thr), (n, t) in rows[:60]:
    tot += t / 1e6 / steps
    print(f"{name:70s} {wg:7d} {thr:5d} {n:6d} {t / n / 1e3:8.1f} {t / 1e6 / steps:8.3f}")
print(f"total {tot:.3f} ms/step in launches of < {maxwg} workgroups and >= {minus} us")
