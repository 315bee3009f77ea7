"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / share."""
import re, sqlite3, sys

db = sys.argv[1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
scol = [r[1] for r in c.execute(f"pragma table_info({ks})")]
namecol = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
rows = c.execute(f"select s.{namecol}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                 f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for n, cnt, t, mn, mx in rows:
    n = re.sub(r"\(.*", "", n)[:90]
    print(f"{n:90s} {cnt:6d} {t/1e6:10.3f} {t/cnt/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {100*t/tot:6.2f}")
print(f"TOTAL kernel time {tot/1e6:.3f} ms")
