"""the kernels of ONE sweep out of a rocprofv3 results database: start, duration, queue, stream (argv[2]: which occurrence of the transfer kernel, default -2)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from %s d join %s s on d.kernel_id=s.id order by d.start" % (disp, sym)))
idx = [i for i, r in enumerate(rows) if 'chain_walk_kernelILi0' in r[0]]
i0 = idx[int(sys.argv[2]) if len(sys.argv) > 2 else -2]
t0 = rows[i0][1]
for r in rows[max(0, i0 - 4):i0 + 14]:
    name = r[0].replace('_ZN4klat12_GLOBAL__N_1', '').replace('_ZN4klat', '')[:44]
    print("%-46s start %8.1f us  end %8.1f  dur %7.1f us  queue %s stream %s" % (name, (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4]))
