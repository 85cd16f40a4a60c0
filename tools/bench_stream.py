#!/usr/bin/env python
"""Micro-benchmark of the grid-sized streaming kernels of the step at BASELINE configs[1] (B = 16, V = 100, 64 channels): the Cout = 1
translation head (forward / weight gradient), the fused input conv (forward).  Prints ms per call and the HBM rate on the algorithmic bytes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voxactb_amd import ops

dev = 'cuda:0'
B, V, C = 16, 100, 64
g = torch.Generator(device=dev); g.manual_seed(0)
u = torch.randn(B, V, V, V, C, device=dev, generator=g)
w = torch.randn(1, C, 3, 3, 3, device=dev, generator=g) * 0.05
bias = torch.zeros(1, device=dev)
vox = torch.randn(B, V, V, V, 10, device=dev, generator=g)
Wi = torch.randn(C, 10, device=dev, generator=g) * 0.3
bi = torch.zeros(C, device=dev)
dq = torch.randn(B, V, V, V, device=dev, generator=g) * 1e-3
dw, db = torch.zeros_like(w), torch.zeros(1, device=dev)


def t(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ops.PRECISION = 'bf16x3'
gb = B * V ** 3 * 4 / 1e9
out = {}
ms = t(lambda: ops.conv3_c1_fwd(u, w, bias, B, V)); out['c1_fwd'] = {'ms': ms, 'TBps': (gb * 65) / ms}
ms = t(lambda: ops.conv3_c1_wgrad(u, dq, dw, db, B, V)); out['c1_wgrad'] = {'ms': ms, 'TBps': (gb * 65) / ms}
ms = t(lambda: ops.pointwise_ss3d_fwd(vox, Wi, bi, B, V)); out['pw_fwd_ss'] = {'ms': ms, 'TBps': (gb * 74) / ms}
print(json.dumps(out))
