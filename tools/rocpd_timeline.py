"""One steady-state step out of a rocprofv3 rocpd kernel trace: every launch in order with its
duration and the idle gap before it, then totals (busy, idle, idle by the kernel that follows).
usage: rocpd_timeline.py trace.db [marker-kernel-substring] [step index from the end]"""
import re, sqlite3, sys
from collections import defaultdict

db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "dice_focal_fwd"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, "
                 f"d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
assert len(marks) > back, (len(marks), "marker launches")
lo, hi = marks[-back - 1], marks[-back]
step = rows[lo:hi]
short = lambda n: re.sub(r"^_Z\d+|Ev\d*\w*Args.*|\.kd$", "", n)[:70]
busy = idle = 0
by_next = defaultdict(lambda: [0, 0.0])
by_kernel = defaultdict(lambda: [0, 0.0])
prev_end = None
for n, s, e, gx, gy, gz, wx in step:
    gap = 0 if prev_end is None else max(0, s - prev_end)
    prev_end = max(e, prev_end or e)
    busy += e - s
    idle += gap
    by_next[short(n)][0] += 1
    by_next[short(n)][1] += gap
    key = (short(n), gx // max(wx, 1), gy, gz)
    by_kernel[key][0] += 1
    by_kernel[key][1] += e - s
    if "-v" in sys.argv:
        print(f"{(s - step[0][1]) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {short(n)} ({gx // max(wx,1)},{gy},{gz})")
span = step[-1][2] - step[0][1]
print(f"step span {span / 1e6:.3f} ms, {len(step)} launches, busy {busy / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms")
print("idle before (top 12):")
for k, (cnt, g) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {g / 1e3:8.1f} us over {cnt:3d} launches  {k}")
print("busy by kernel / grid (top 45):")
for k, (cnt, t) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {t / 1e3:8.1f} us  x{cnt:2d}  avg {t / cnt / 1e3:8.1f}  {k[0]} grid=({k[1]},{k[2]},{k[3]})")
