import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.test_grad_noise_gpu import _measure, _kinks, FIXTURES, GOLDEN
for fx in FIXTURES:
    g = np.load(os.path.join(GOLDEN, fx + '.npz'), allow_pickle=False)
    for kernel in ('r3/f16', 'f16/f16'):
        eq, rows, loss = _measure(g, 'bf16x3', kernel.split('/')[0] if kernel.startswith('f16') else kernel, True, '%s/%s forced' % (fx[10:], kernel), kinks=_kinks(fx))
        top = max(q[3] for q in rows)
        rel = [r[2] / (r[3] + 1e-300) for r in rows if r[3] > 1e-6 * top]
        print('SUMMARY %-10s %-8s  Q err %.2e  median %.2e  worst %.2e  worst x gate %.2f  n>gate %d' % (fx[10:], kernel, eq, float(np.median(rel)), max(rel), rows[0][0], sum(1 for r in rows if r[0] > 1.0)), flush=True)
