"""Per-kernel calls / total / average time from a rocprofv3 `--kernel-trace --output-format csv` run.
usage: trace_stats.py <output dir> [header line]"""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
agg = {}
with open(f) as fh:
    for row in csv.DictReader(fh):
        name = re.sub(r"\(.*", "", row["Kernel_Name"])[:90]
        dt = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dt
        a[2] = min(a[2], dt)
        a[3] = max(a[3], dt)
tot = sum(a[1] for a in agg.values())
if len(sys.argv) > 2:
    print(sys.argv[2])
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for n, (cnt, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:90s} {cnt:6d} {t/1e6:10.3f} {t/cnt/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {100*t/tot:6.2f}")
print(f"TOTAL kernel time {tot/1e6:.3f} ms")
