"""GPU parity of the `one_policy_more_heads` baseline encoder (reference
PerceiverVoxelLang2RobotsEncoder, perceiver_lang_io.py:488-860) against fixtures captured from the reference
(tests/golden/make_golden.py, section f11): the six outputs within 1e-4, the summed two-arm loss within 1e-4, every parameter
gradient within 3e-3 -- in exact 'fp32' and in the default 'bf16x3' precision."""
import numpy as np
import pytest
import torch

from oracle import weights as ow
from voxactb_amd import ops, synthetic
from voxactb_amd.agents.peract_bc.perceiver_lang_io import PerceiverVoxelLang2RobotsEncoder

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_Q = 1e-4


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def run(g, cams, precision):
    V, B = int(g['cfg_V']), int(g['cfg_B'])
    enc = PerceiverVoxelLang2RobotsEncoder(
        depth=int(g['cfg_depth']), iterations=1, voxel_size=V, initial_dim=10, low_dim_size=int(g['cfg_low_dim']),
        num_latents=int(g['cfg_latents']), voxel_patch_size=int(g['cfg_k']), voxel_patch_stride=int(g['cfg_s']),
        activation='lrelu', input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0)
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    enc = enc.to(DEV)
    rs = synthetic.make_replay_sample(B, cams, (int(g['cfg_H']), int(g['cfg_W'])), V, int(g['cfg_low_dim']), seed=1)
    rs = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
    grid = T(g['grid']).to(DEV)
    eng = enc.engine()
    eng.precision = precision
    outs, cache = eng.forward(grid, rs['low_dim_state'].float().to(DEV), rs['lang_token_embs'].float().to(DEV), training=False,
                              save=True, proprio_left=T(g['proprio_left']).to(DEV))
    assert len(outs) == 6
    errs = [maxerr(o, T(g[k])) for o, k in zip(outs, ('q_trans_right', 'rot_grip_right', 'collision_right', 'q_trans_left',
                                                      'rot_grip_left_out', 'collision_left'))]
    print('%s forward max-abs (right trans/rot/coll, left trans/rot/coll): %s' % (precision, ' '.join('%.2e' % e for e in errs)))
    assert max(errs) < TOL_Q, errs
    # the module's forward() is the reference's call signature (agent :944-952)
    o2 = enc(grid.permute(0, 4, 1, 2, 3), rs['low_dim_state'].float().to(DEV), T(g['proprio_left']).to(DEV), None,
             rs['lang_token_embs'].float().to(DEV), None, None, None)
    assert len(o2) == 6 and maxerr(o2[3], outs[3]) == 0.0

    # summed two-arm loss (agent :1283-1369) and its gradients
    def arm_loss(q, o, trans, rot_grip):
        at = trans.long()
        lab = ((at[:, 0] * V + at[:, 1]) * V + at[:, 2]).int().to(DEV)
        dq = torch.empty((B, V ** 3), device=DEV)
        l_t, _, _ = ops.ce_big(q.reshape(B, -1), lab, dq, 1.0 / B)
        labs = torch.cat([rot_grip.int(), rs['ignore_collisions'].int()[:, :1]], 1).to(DEV).contiguous()
        d_o = torch.empty_like(o)
        l_h, _ = ops.ce_rows(o, [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)], labs, d_o, 1.0 / B)
        return l_t + l_h.sum(1), dq, d_o

    lr_, dq_r, do_r = arm_loss(outs[0], cache['o'], rs['trans_action_indicies'], rs['rot_grip_action_indicies'])
    ll_, dq_l, do_l = arm_loss(outs[3], cache['left']['o'], T(g['trans_left']), T(g['rot_grip_left']))
    loss = float((lr_ + ll_).mean())
    assert abs(loss - float(g['loss'])) < 1e-4, (loss, float(g['loss']))
    for p in enc.parameters():
        p.grad = None
    eng.backward(cache, dq_r, do_r, dq_trans_left=dq_l, d_o_left=do_l)
    P = dict(enc.named_parameters())
    bad, worst = [], 0.0
    for n, rn in zip([str(n) for n in g['grad_names']], T(g['grad_norms'])):
        gn, rn = float(P[n].grad.norm()), float(rn)
        if abs(gn - rn) > 3e-3 * rn + 1e-5:
            bad.append((n, gn, rn))
        elif rn > 1e-4:
            worst = max(worst, abs(gn - rn) / rn)
        key = 'grad__' + n
        if key in g.files:
            ref = T(g[key])
            e = maxerr(P[n].grad, ref)
            if e > 3e-3 * float(ref.abs().max()) + 1e-5:
                bad.append((n, 'full', e, float(ref.abs().max())))
    print('%s loss %.6f (reference %.6f), worst grad-norm rel. error %.2e' % (precision, loss, float(g['loss']), worst))
    assert not bad, bad


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_2robots_tiny(golden, precision):
    run(golden('f11_encoder_2robots_tiny'), ['front', 'wrist'], precision)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_2robots_c1(golden, precision):
    run(golden('f11_encoder_2robots_c1'), ['front'], precision)
