#!/usr/bin/env python
"""Print a per-kernel summary (calls, total us, avg us, %) from a rocprofv3 results .db (rocpd sqlite)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print('%-90s %8s %12s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
for n, k, t, a, p in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('%-90s %8d %12.1f %10.2f %6.2f' % (n[:90], k, t / 1 if t else 0, a, p))
