"""Two-stream timeline of a rocprofv3 `--kernel-trace --output-format csv` run: per queue the summed
kernel time, the time at least one / at least two kernels were running, and which main-queue
kernels ran alone (nothing from the other queue beside them) -- where the next overlap could come from.
usage: trace_overlap.py <output dir> [steps in the trace=1] [skip first fraction=0.5]"""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"],
                     re.sub(r"\(.*", "", r["Kernel_Name"])[:60]))
rows.sort()
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip      # drop warm-up / start-up
rows = [r for r in rows if r[0] >= t0]
queues = {}
for s, e, q, n in rows:
    queues.setdefault(q, [0, 0])
    queues[q][0] += 1
    queues[q][1] += e - s
main = max(queues, key=lambda q: queues[q][1])
ev = []
for s, e, q, n in rows:
    ev.append((s, 1, q))
    ev.append((e, -1, q))
ev.sort()
busy1 = busy2 = 0
depth, last = 0, ev[0][0]
for t, d, q in ev:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth += d
    last = t
span = rows[-1][1] - rows[0][0]
print(f"span {span / 1e6:.2f} ms; queues: " + ", ".join(
    f"{q}{' (main)' if q == main else ''}: {c} launches, {t / 1e6:.2f} ms" for q, (c, t) in queues.items()))
print(f">= 1 kernel running {busy1 / 1e6:.2f} ms, >= 2 running {busy2 / 1e6:.2f} ms, idle "
      f"{(span - busy1) / 1e6:.2f} ms")
# main-queue kernels by how much of their time a side-queue kernel was running beside them
side = sorted((s, e) for s, e, q, n in rows if q != main)
import bisect
starts = [s for s, e in side]
alone = {}
for s, e, q, n in rows:
    if q != main:
        continue
    i = max(bisect.bisect_left(starts, s) - 64, 0)
    cov = 0
    cur = s
    for ss, ee in side[i:]:
        if ss >= e:
            break
        lo, hi = max(ss, cur), min(ee, e)
        if hi > lo:
            cov += hi - lo
            cur = hi
    a = alone.setdefault(n, [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    a[2] += (e - s) - cov
print(f"{'main-queue kernel':60s} {'calls':>6s} {'ms/step':>8s} {'alone ms/step':>14s}")
for n, (c, t, al) in sorted(alone.items(), key=lambda kv: -kv[1][2])[:25]:
    print(f"{n:60s} {c:6d} {t / 1e6 / steps:8.3f} {al / 1e6 / steps:14.3f}")
