"""Per-kernel table from rocprofv3 PMC passes (tools/pmc_clock.sh, tools/profile_round.sh): mean duration (kernel trace of the
same passes), effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, matrix-pipe duty = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
cycles), and the split of the waves' time (SQ_WAVE_CYCLES) into issuing (SQ_ACTIVE_INST_ANY), waiting on s_waitcnt / barriers
(SQ_WAIT_ANY) and waiting for an issue slot / the pipe (SQ_WAIT_INST_ANY); VALU instructions per MFMA.

    python tools/pmc_clock.py <dir> [top_n]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
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
names = sorted(dur, key=lambda k: -dur[k][1])[:top]
print('%-64s %9s %6s %7s | %6s %6s %6s | %9s' % ('kernel (mean per dispatch, profiled passes)', 'us', 'GHz', 'MFMA %', 'issue', 'waitc', 'pipe', 'VALU/MFMA'))
for k in names:
    n, t = dur[k]
    us = t / n
    c = {a: v / m for a, (m, v) in cnt[k].items()}
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    ghz = cyc / us * 1e-3 if us else 0.0
    duty = 100.0 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024.0 * cyc) if cyc else 0.0
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    pct = lambda x: 100.0 * c.get(x, 0.0) / wc if wc else 0.0
    vm = c.get('SQ_INSTS_VALU', 0.0) / c['SQ_INSTS_MFMA'] if c.get('SQ_INSTS_MFMA') else float('nan')
    print('%-64s %9.1f %6.2f %7.1f | %6.1f %6.1f %6.1f | %9.2f' % (k.replace('(anonymous namespace)::', '').replace('void ', '')[:64], us, ghz, duty,
                                                                  pct('SQ_ACTIVE_INST_ANY'), pct('SQ_WAIT_ANY'), pct('SQ_WAIT_INST_ANY'), vm))
if '-v' in sys.argv:
    for k in names:
        print(k[:120])
        for cn, (m, v) in sorted(cnt[k].items()):
            print('   %-32s %16.4g' % (cn, v / m))
