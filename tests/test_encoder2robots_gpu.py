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


def run(g, cams, precision, norm_tol=3e-3, elem_tol=3e-3, flips=0, choices=None):
    V, B = int(g['cfg_V']), int(g['cfg_B'])
    enc = PerceiverVoxelLang2RobotsEncoder(
        depth=int(g['cfg_depth']), iterations=1, voxel_size=V, initial_dim=10, low_dim_size=int(g['cfg_low_dim']),
        num_latents=int(g['cfg_latents']), voxel_patch_size=int(g['cfg_k']), voxel_patch_stride=int(g['cfg_s']),
        activation='lrelu', input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0)
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    enc = enc.to(DEV)
    rs = synthetic.make_replay_sample(B, cams, (int(g['cfg_H']), int(g['cfg_W'])), V, int(g['cfg_low_dim']), seed=1)
    rs = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
    if 'grid' in g.files:
        grid = T(g['grid']).to(DEV)
    else:                                                    # headline-size fixture: voxelize the seeded clouds on the device
        from voxactb_amd.voxel.voxel_grid import VoxelGrid
        H, W = int(g['cfg_H']), int(g['cfg_W'])
        vg = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, H * W * len(cams))
        rgbs = [((rs['%s_rgb' % c].float() / 255.0) * 2.0 - 1.0).to(DEV) for c in cams]
        grid = vg.voxelize_cameras([rs['%s_point_cloud' % c].float().to(DEV) for c in cams], rgbs,
                                   torch.tensor([synthetic.SCENE_BOUNDS]).to(DEV))
    eng = enc.engine()
    eng.precision = precision
    outs, cache = eng.forward(grid, rs['low_dim_state'].float().to(DEV), rs['lang_token_embs'].float().to(DEV), training=False,
                              save=True, proprio_left=T(g['proprio_left']).to(DEV))
    assert len(outs) == 6
    if 'q_trans_right' in g.files:
        errs = [maxerr(o, T(g[k])) for o, k in zip(outs, ('q_trans_right', 'rot_grip_right', 'collision_right', 'q_trans_left',
                                                          'rot_grip_left_out', 'collision_left'))]
    else:
        errs = [maxerr(outs[1], T(g['rot_grip_right'])), maxerr(outs[2], T(g['collision_right'])),
                maxerr(outs[4], T(g['rot_grip_left_out'])), maxerr(outs[5], T(g['collision_left']))]
        sidx = T(g['q_trans_sample_idx']).long()
        for side, q in (('right', outs[0]), ('left', outs[3])):
            flat = q.reshape(B, -1).float().cpu()
            assert torch.equal(flat.argmax(1), T(g['q_trans_%s_argmax' % side]))
            errs.append(float((flat[:, sidx] - T(g['q_trans_%s_sample' % side])).abs().max()))
            errs.append(float((torch.gather(flat, 1, T(g['q_trans_%s_top_idx' % side]).long()) - T(g['q_trans_%s_top_vals' % side])).abs().max()))
            errs.append(float((torch.logsumexp(flat.double(), 1) - T(g['q_trans_%s_lse' % side])).abs().max()))
    print('%s forward max-abs (right trans/rot/coll, left trans/rot/coll): %s' % (precision, ' '.join('%.2e' % e for e in errs)))
    assert max(errs) < TOL_Q, errs
    # the module's forward() is the reference's call signature (agent :944-952)
    if V <= 32:
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
    if choices is not None and precision != 'fp32':
        # the backward at the reference run's LeakyReLU / max-pool choices (fixture supplement *_pools.npz, round 6), as the single-arm digests
        from tests.test_c2_reference_gpu import force_kinks, force_pools
        assert np.array_equal(choices['q_trans_lse'], g['q_trans_right_lse'])             # (the supplement of THIS forward)
        print('%s: LeakyReLU choices near zero that differ from the reference run\'s %d, max-pool ties %s' % (
            precision, force_kinks(cache, choices), force_pools(cache, choices, eng.T0)))
    eng.backward(cache, dq_r, do_r, dq_trans_left=dq_l, d_o_left=do_l)
    P = dict(enc.named_parameters())
    bad, worst = [], 0.0
    for n, rn in zip([str(n) for n in g['grad_names']], T(g['grad_norms'])):
        gn, rn = float(P[n].grad.norm()), float(rn)
        key64 = 'dysum64__' + n
        if key64 in g.files:
            # bias of a grid conv at V = 100: the yardstick is the reference's dY summed in float64 (its own fp32 sum is off by
            # up to 1.5e-2 here; tests/test_c2_reference_gpu.py has the full story)
            ref = T(g[key64])
            e = float((P[n].grad.double().cpu() - ref).abs().max())
            if e > max(2e-3, elem_tol) * float(ref.abs().max()) + 2e-5:
                bad.append((n, 'vs float64 dY sum', e, float(ref.abs().max())))
            continue
        if abs(gn - rn) > norm_tol * rn + 1e-5:
            bad.append((n, gn, rn))
        elif rn > 1e-4:
            worst = max(worst, abs(gn - rn) / rn)
        key = 'grad__' + n
        if key in g.files:
            ref = T(g[key])
            d = (P[n].grad.detach().float().cpu() - ref).abs()
            lim = elem_tol * float(ref.abs().max()) + 1e-5
            if int((d > lim).sum()) > flips:           # `flips`: elements allowed to sit on the other side of a LeakyReLU kink
                bad.append((n, 'full', float(d.max()), float(ref.abs().max())))
    print('%s loss %.6f (reference %.6f), worst grad-norm rel. error %.2e' % (precision, loss, float(g['loss']), worst))
    for b_ in bad:
        print('   BAD', b_)
    assert not bad, bad


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_2robots_tiny(golden, precision):
    run(golden('f11_encoder_2robots_tiny'), ['front', 'wrist'], precision)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_2robots_c1(golden, precision):
    run(golden('f11_encoder_2robots_c1'), ['front'], precision)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_2robots_headline_size_digest(golden, precision):
    """V=100, depth 6, 2048 latents (BASELINE.json configs[1] geometry) with the 192-wide context and both head sets:
    forward digests, summed two-arm loss, every parameter gradient."""
    # round 6: the fixture's supplement carries the reference run's LeakyReLU choices near zero and its max-pool arg-maxima, and with the
    # backward evaluated there the default precision holds the SAME gates as exact fp32 (3e-3 on norms and on the small tensors in full, no
    # element exempted).  Rounds 2 - 5 ran this digest at 5e-3 / 2e-2 with one exempted element and read the differences as SpatialSoftmax3D's
    # 1 / 0.01 temperature amplifying a 1e-5 forward difference; they were kinks (13 752 pre-activations of u0 lie within 3e-5 of zero) and
    # one pool whose two largest voxels are 6e-7 apart.  fp32: un-forced.
    run(golden('f11c2_encoder_2robots_c2_digest'), synthetic.CAMERAS4, precision, choices=golden('f11c2_encoder_2robots_c2_digest_pools'))
    torch.cuda.empty_cache()
