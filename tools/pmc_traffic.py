#!/usr/bin/env python
"""Per-launch HBM traffic of each kernel from two rocprofv3 PMC passes (CSV output):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirW> -- python3 bench.py ...
    python tools/pmc_traffic.py <dirF> <dirW> profiles/rNN_pmc_traffic.json <bench_line.json>

The last argument is the JSON line bench.py printed in the profiled run: its
``config.workload`` (shape, batch, loss, optimiser) and ``dtype`` are stored with the
figures, and bench.py attaches a traffic figure only to a run of the same workload.

Units / corrections as MI355X_MICROARCH.md "HBM" prescribes: the counters are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so the read side
is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores. traffic = 2*FETCH + WRITE.
"""
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"])
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"<.*", "", name)
            a = agg.setdefault(name, [0.0, 0, 0.0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    return agg


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
line = {}
if len(sys.argv) > 4:
    with open(sys.argv[4]) as fh:
        for ln in fh:
            if ln.startswith("{"):
                line = json.loads(ln)
out = {"unit": "bytes per launch (mean)",
       "workload": line.get("config", {}).get("workload"), "dtype": line.get("dtype"),
       "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950: FETCH_SIZE counts 128-B requests "
                  "as 64 B; separate --pmc passes]", "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k][0]):
    if k not in write:
        continue
    fr, n, us = fetch[k]
    wr, n2, _ = write[k]
    out["kernels"][k] = {"launches": n, "fetch_size_kib_mean": fr / n,
                         "write_size_kib_mean": wr / max(n2, 1),
                         "traffic_bytes_mean": (2 * fr / n + wr / max(n2, 1)) * 1024,
                         "avg_us_under_pmc": us / n}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:12]:
    print(f"{k[:60]:60s} n={v['launches']:5d} traffic/launch {v['traffic_bytes_mean'] / 1e6:9.2f} MB")
