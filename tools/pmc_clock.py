"""Summarise tools/pmc_clock.sh: per kernel mean duration (kernel trace) and mean of every collected counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
dur = defaultdict(lambda: [0, 0.0])
cnt = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        d = dur[row['Kernel_Name']]
        d[0] += 1
        d[1] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-3
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        c = cnt[row['Kernel_Name']][row['Counter_Name']]
        c[0] += 1
        c[1] += float(row['Counter_Value'])
names = sorted(dur, key=lambda k: -dur[k][1])[:14]
for k in names:
    n, t = dur[k]
    print('%s\n   calls %d  mean %.1f us (profiled)' % (k[:110], n, t / n))
    for c, (m, v) in sorted(cnt[k].items()):
        print('   %-32s %16.4g' % (c, v / m))
