"""Per-kernel average / minimum duration (us) from a rocprofv3 --kernel-trace results .db, grouped by grid size."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'rocpd_kernel_dispatch' in t][0]; ks = [t for t in tabs if 'rocpd_info_kernel_symbol' in t][0]
pat = sys.argv[2] if len(sys.argv) > 2 else "blh"
q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, d.grid_size_x, d.workgroup_size_x "
     "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x order by s.kernel_name, d.grid_size_x" % (kd, ks))
for r in cur.execute(q):
    if pat in r[0]:
        print("%-70s n=%5d avg %7.2f us min %7.2f  wgs %6d x %d" % (r[0][:70], r[1], r[2], r[3], r[4] // r[5], r[5]))
