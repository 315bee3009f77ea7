"""Per-launch-shape breakdown of the conv kernels in a rocprofv3 rocpd trace."""
import re, sqlite3, sys
db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(
    f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, "
    f"d.group_segment_size, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id "
    f"where s.kernel_name like '%igemm%' or s.kernel_name like '%wgrad_kernel%' "
    f"group by 1,2,3,4,6 order by count(*)*avg(d.end-d.start) desc").fetchall()
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    n = re.sub(r"_Z23adell_conv_|Ev8ConvArgs.kd|Ev9WgradArgs.kd", "", r[0])
    print(f"{n:28s} grid=({r[1]//r[4]},{r[2]},{r[3]}) lds={r[5]:6d} calls={r[6]:3d} "
          f"avg_us={r[7]/1e3:8.1f} ms/step={r[6]*r[7]/1e6/steps:6.2f}")
