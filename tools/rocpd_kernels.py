"""Per-kernel totals of a rocprofv3 run stored as a rocpd sqlite database (the default output format of rocprofv3 here)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, s.arch_vgpr_count, s.group_segment_size from {kd} d join {ks} s "
     "on d.kernel_id=s.id group by 1 order by 3 desc limit 12")
for r in c.execute(q):
    print(f"{r[2]:10.1f} ms {r[1]:5d} calls  vgpr {r[3]}  {r[0][:110]}")
