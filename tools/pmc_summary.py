#!/usr/bin/env python
"""Per-kernel mean of one rocprofv3 PMC counter (csv output of `rocprofv3 --kernel-trace --pmc X --output-format csv`).

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [top_n]

Prints kernel, dispatches, mean counter value per dispatch and the total, sorted by total.  FETCH_SIZE / WRITE_SIZE are
reported by rocprofv3 in KiB-like units of 1024 B? -- no: in BYTES / 1024 is NOT applied here; the raw value is printed
and the caller applies the unit and the gfx950 correction (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads -> double it)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    files = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        print('no *counter_collection.csv under', root)
        return
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name') or row.get('kernel_name')
                cn = row.get('Counter_Name') or row.get('counter_name')
                cv = float(row.get('Counter_Value') or row.get('counter_value') or 0)
                a = agg[cn][name]
                a[0] += 1
                a[1] += cv
    for cn, d in agg.items():
        print('== %s' % cn)
        print('%-90s %8s %16s %16s' % ('kernel', 'calls', 'mean/dispatch', 'total'))
        for name, (n, tot) in sorted(d.items(), key=lambda kv: -kv[1][1])[:top]:
            print('%-90s %8d %16.1f %16.1f' % (name[:90], n, tot / n, tot))


if __name__ == '__main__':
    main()
