#!/usr/bin/env python
"""fixture f5gb8 (configs[1] geometry, B = 8): which arithmetic moves the conv bias gradients of the up-block away from the float64 sums of
the reference's dY?  Runs the default forward once and the backward under several switches; prints per-sample maxima of dq / du."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tests.test_c2_reference_gpu import DEV, T, _setup
from tests.conftest import GOLDEN
from tools.experiments.fwd_sensitivity_gpu import loss_grads
from voxactb_amd import ops

g = np.load(os.path.join(GOLDEN, (sys.argv[1] if len(sys.argv) > 1 else 'f5gb8_encoder_c2_b8_grads') + '.npz'), allow_pickle=False)
enc, rs, grid, arm, V, B = _setup(g)
eng = enc.engine()
eng.precision = 'bf16x3'
outs, cache = eng.forward(grid, rs['low_dim_state'].to(DEV), rs['lang_token_embs'].to(DEV), training=False, save=True)
dq, d_o, d_arm = loss_grads(rs, outs, cache, arm, V, B)
print('per-sample max |dq_trans|:', [float(x) for x in dq.abs().amax(1)])
P = dict(enc.named_parameters())
names = [str(n) for n in g['grad_names']]
ref_norm = dict(zip(names, [float(x) for x in T(g['grad_norms'])]))


def run(tag, **sw):
    old = {k: getattr(ops, k) for k in sw if hasattr(ops, k)}
    olde = {k: getattr(eng, k) for k in sw if hasattr(eng, k) and not hasattr(ops, k)}
    for k, v in sw.items():
        setattr(ops if hasattr(ops, k) else eng, k, v)
    for p in enc.parameters():
        p.grad = None
    eng.backward(dict(cache), dq, d_o, d_arm)
    out = []
    for n in ('up0.conv_up.0.conv3d.bias', 'up0.conv_up.2.conv3d.bias', 'final.conv3d.bias', 'input_preprocess.conv3d.bias'):
        ref = T(g['dysum64__' + n])
        out.append('%s %.2e' % (n.split('.conv3d')[0], float((P[n].grad.double().cpu() - ref).abs().max()) / float(ref.abs().max())))
    worst = max((abs(float(P[n].grad.norm()) - ref_norm[n]) / (ref_norm[n] + 1e-12), n) for n in names if ref_norm[n] > 1e-4 and ('dysum64__' + n) not in g.files)
    print('%-44s bias-gradient error / max: %s | worst norm error %.2e (%s)' % (tag, '  '.join(out), worst[0], worst[1]), flush=True)
    for k, v in old.items():
        setattr(ops, k, v)
    for k, v in olde.items():
        setattr(eng, k, v)


run('default')
run('DGRAD_PRECISION=bf16x3 (conv data gradients)', DGRAD_PRECISION='bf16x3')
run('LIN_DGRAD_X2 off (linear data gradients x3)', LIN_DGRAD_X2=False)
run('wgrad_precision=bf16x3', wgrad_precision='bf16x3')
run('attn_bwd_kernel off (round-3 x3 backward)', attn_bwd_kernel='')
run('all of the above', DGRAD_PRECISION='bf16x3', LIN_DGRAD_X2=False, wgrad_precision='bf16x3', attn_bwd_kernel='')
