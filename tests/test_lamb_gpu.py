"""GPU: the fused multi-tensor LAMB kernel (vxb_lamb_step_f32) replays the reference optimizer's known-answer fixture F7
(tests/golden/f7_lamb.npz: three tensors x three steps of peract/helpers/optim/lamb.py:60-124, incl. the all-zero tensor
whose first step takes the `weight_norm == 0 or adam_norm == 0 -> trust_ratio = 1` branch, :108-115)."""
import numpy as np
import pytest
import torch

from voxactb_amd.flat_params import FlatParams
from voxactb_amd.helpers.optim.lamb import Lamb

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_lamb_kat_f7(golden):
    g = golden('f7_lamb')
    names = ('w_rand', 'w_zero', 'w_big')
    mod = torch.nn.Module()
    for n in names:                     # one FlatParams arena holding all three tensors (one fused launch per step)
        mod.register_parameter(n, torch.nn.Parameter(T(g[n + '_w0']).clone()))
    arena = FlatParams(mod, DEV)
    opt = Lamb(mod.parameters(), lr=5e-4, weight_decay=1e-6, betas=(0.9, 0.999), adam=False)
    opt.attach(arena)
    P = dict(mod.named_parameters())
    worst = 0.0
    for step in range(3):
        arena.zero_grad()
        for n in names:
            P[n].grad.copy_(T(g['%s_g%d' % (n, step)]).to(DEV))
        opt.step()
        for n in names:
            ref = T(g[n + '_w'])[step]
            got = P[n].data.cpu()
            # the update is w - lr * r * u with r = ||w|| / ||u||: the two norms are block-tree sums on the device and
            # torch's pairwise sums on the CPU, so r may differ in its last bit -> the update by <= 1 ulp of |lr * r * u|
            err = float((got - ref).abs().max())
            tol = 4 * float(np.spacing(np.float32(ref.abs().max())))
            assert err <= tol, (n, step, err, tol)
            worst = max(worst, err)
        assert opt.state[P['w_rand']]['step'] == step + 1
    # the zero tensor's first step: gradient 0, weights 0 -> u = 0, trust ratio forced to 1, weights stay exactly 0
    assert float(T(g['w_zero_w'])[0].abs().max()) == 0.0
    print('worst abs error over 3 tensors x 3 steps: %.3e' % worst)
