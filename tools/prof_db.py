"""kernel durations out of a rocprofv3 results database (rocpd sqlite): name, launches, avg / min / max in us"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s "
     "on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (disp, sym))
for r in cur.execute(q):
    print("%-100s n=%d avg=%.1f us min=%.1f max=%.1f" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3))
