"""Matrix-pipe busy fraction and the clock the chip held, per kernel, from a rocprofv3
`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv`
run of bench.py. Counter values are sums over the chip: MFMA_BUSY over 1024 SIMDs (32 cycles per
v_mfma_f32_32x32x16_f16), BUSY_CU over 256 CUs, GRBM_GUI_ACTIVE over the 8 XCDs
(MI355X_MICROARCH.md, DVFS give-back: clock ~ GRBM_GUI_ACTIVE / 8 / duration; it reads high on
dispatches shorter than ~0.3 ms). Only launches of at least `min_us` are averaged.
usage: pmc_mfma.py <output dir> [min_us=150]"""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
per = {}
with open(f) as fh:
    for row in csv.DictReader(fh):
        key = (row["Dispatch_Id"], re.sub(r"\(.*", "", row["Kernel_Name"])[:70])
        d = per.setdefault(key, {"us": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3})
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
agg = {}
for (_, name), d in per.items():
    if d["us"] < min_us or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
        continue
    a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
    a[0] += 1
    a[1] += d["us"]
    a[2] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a[3] += d.get("SQ_BUSY_CU_CYCLES", 0.0)
    a[4] += d.get("GRBM_GUI_ACTIVE", 0.0)
print(f"launches >= {min_us:.0f} us only; MFMA_BUSY / 1024 SIMDs, BUSY_CU / 256 CUs, clock = GRBM_GUI_ACTIVE / 8 / duration")
print(f"{'kernel':70s} {'n':>4s} {'avg_us':>9s} {'mfma_busy/SIMD Mcyc':>20s} {'busy_cu/CU Mcyc':>16s} {'clock GHz':>10s} "
      f"{'pipe busy of clock':>19s} {'of 2.4 GHz':>11s}")
for name, (n, us, mf, cu, gui) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    mf_s, cu_c = mf / n / 1024, cu / n / 256
    dur = us / n * 1e-6
    clock = gui / n / 8 / dur if gui > 0 else 0.0
    busy_clock = mf_s / (clock * dur) if clock > 0 else 0.0
    busy_nom = mf_s / (2.4e9 * dur)
    print(f"{name:70s} {n:4d} {us / n:9.1f} {mf_s / 1e6:20.3f} {cu_c / 1e6:16.3f} {clock / 1e9:10.2f} "
          f"{100 * busy_clock:18.1f}% {100 * busy_nom:10.1f}%")
