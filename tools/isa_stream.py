"""Instruction-stream view of one kernel of a hipcc -S listing: per basic block the counts by class and the issue order as a
string (M = MFMA, v = VALU, t = transcendental, d = LDS, g = global / LDS-DMA, s = SALU, w = s_waitcnt, b = barrier, n = s_nop).

    python tools/isa_stream.py file.s <kernel-name-substring> [min-instructions-per-block]
"""
import re
import sys


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 'M'
    if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op):
        return 't'
    if op.startswith('v_'):
        return 'v'
    if op.startswith('ds_'):
        return 'd'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'g'
    if op == 's_waitcnt':
        return 'w'
    if op == 's_barrier':
        return 'b'
    if op == 's_nop':
        return 'n'
    if op.startswith('s_'):
        return 's'
    return '?'


def main():
    path, name = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if re.match(r'^[A-Za-z_][^ ]*:', l) and name in l.split(':')[0]:
            start = i
            break
    if start is None:
        raise SystemExit('kernel not found')
    blocks, cur, label = [], [], lines[start]
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end') or s.startswith('s_endpgm'):
            blocks.append((label, cur))
            break
        if re.match(r'^\.LBB\d+_\d+:', s):
            blocks.append((label, cur))
            cur, label = [], s.split(';')[0].strip() + ('  ' + s.split(';', 1)[1].strip() if ';' in s else '')
            continue
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        cur.append(s.split()[0])
        if s.startswith('s_cbranch') or s.startswith('s_branch'):
            blocks.append((label, cur))
            cur, label = [], '(after %s)' % s.replace('\t', ' ')
    for label, ops in blocks:
        if len(ops) < minn:
            continue
        st = ''.join(classify(o) for o in ops)
        cnt = {c: st.count(c) for c in 'Mvtdgswbn?'}
        print('%s  n=%d  %s' % (label[:70], len(ops), ' '.join('%s=%d' % kv for kv in cnt.items() if kv[1])))
        for i in range(0, len(st), 120):
            print('   ' + st[i:i + 120])
        scr = [o for o in ops if o.startswith('scratch_')]
        if scr:
            print('   scratch ops: %d' % len(scr))


if __name__ == '__main__':
    main()
