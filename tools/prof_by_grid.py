#!/usr/bin/env python
"""Per-(kernel, grid) launch durations from a rocprofv3 results .db (rocpd sqlite): the same kernel launched on different
problem shapes (e.g. the attention kernels: self-attention vs the two cross-attentions) shows as separate lines.
usage: prof_by_grid.py p_results.db <name substring> [...]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cur = c.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
def col(*cands):
    for k in cands:
        if k in cols:
            return k
    raise SystemExit('no column of %s in %s' % (cands, cols))
name, start, end = col('name', 'kernel_name'), col('start', 'start_timestamp'), col('end', 'end_timestamp')
gx, gy = col('grid_x', 'grid_size_x'), col('grid_y', 'grid_size_y')
wx = col('workgroup_x', 'workgroup_size_x')
agg = collections.OrderedDict()
for n, s, e, x, y, w in c.execute("select %s,%s,%s,%s,%s,%s from kernels" % (name, start, end, gx, gy, wx)):
    if not any(k in n for k in sys.argv[2:]):
        continue
    key = (n[:70], x // max(w, 1), y)
    agg.setdefault(key, []).append((e - s) / 1e3)
print('%-72s %8s %6s %6s %10s %10s %10s' % ('kernel', 'blocks_x', 'y', 'calls', 'avg_us', 'min_us', 'total_us'))
for (n, x, y), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-72s %8d %6d %6d %10.1f %10.1f %10.1f' % (n, x, y, len(v), sum(v) / len(v), min(v), sum(v)))
