"""Mean of each PMC counter per kernel name from a rocprofv3 --pmc CSV directory."""
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for row in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", row["Kernel_Name"])
    if pat not in name:
        continue
    a = agg.setdefault((name[:70], row["Counter_Name"]), [0.0, 0])
    a[0] += float(row["Counter_Value"]); a[1] += 1
for (n, c), (v, k) in sorted(agg.items()):
    print(f"{n:70s} {c:28s} {v / k:16.1f}  (n={k})")
