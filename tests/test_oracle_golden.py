"""CPU: the oracle must reproduce every fixture captured from the reference (tests/golden/make_golden.py)."""
import numpy as np
import torch

from oracle import agent as oagent, perceiver as operc, se3 as ose3, voxel_grid as ovox, weights as ow


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_voxel_kats(golden):
    g = golden('f1_voxel_kats')
    out = ovox.voxelize(T(g['kat_coords']), T(g['kat_feats']), T(g['kat_bounds']), int(g['kat_V']))
    assert torch.equal(out, T(g['kat_grid']))
    occ = ovox.occupied_cells(out)[:, 1:].tolist()
    # SURVEY.md section 4 known answers
    assert occ == [[0, 0, 0], [0, 2, 3], [1, 2, 3], [2, 2, 2], [3, 2, 1]]
    assert torch.allclose(out[0, 2, 2, 2], torch.tensor([.65, .65, .65, .45, .45, .45, .5, .5, .5, 1.]))
    assert out[0, 3, 3, 3].tolist() == [0, 0, 0, 0, 0, 0, 0.75, 0.75, 0.75, 0]
    for c in range(int(g['n_cases'])):
        o = ovox.voxelize(T(g['c%d_coords' % c]), T(g['c%d_feats' % c]), T(g['c%d_bounds' % c]), int(g['c%d_V' % c]))
        assert torch.equal(torch.nan_to_num(o), torch.nan_to_num(T(g['c%d_grid' % c]))), c


def test_res_fp32_quirks():
    # res at V=100 is fp32 0.00999999977..., and res + 1e-12 == res
    b = torch.tensor([[-0.3, -0.5, 0.6, 0.7, 0.5, 1.6]])
    res = (b[:, 3:] - b[:, :3]) / (torch.tensor([100.]) + 1e-12)
    assert torch.equal(res + 1e-12, res)
    assert abs(float(res[0, 0]) - 0.009999999776) < 1e-12


def _enc_inputs(g, cams):
    from voxactb_amd import synthetic
    B, H, W, V = int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), int(g['cfg_V'])
    rs = synthetic.make_replay_sample(B, cams, (H, W), V, int(g['cfg_low_dim']), seed=1, arm_pred_loss=bool(g['cfg_arm']))
    rs = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
    rs = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in rs.items()}
    return rs


def _check_encoder(g, cams, with_grads=True):
    rs = _enc_inputs(g, cams)
    V = int(g['cfg_V'])
    arm = bool(g['cfg_arm'])
    shapes = operc.param_shapes(int(g['cfg_depth']), V, int(g['cfg_low_dim']), num_latents=int(g['cfg_latents']),
                                voxel_patch_size=int(g['cfg_k']), voxel_patch_stride=int(g['cfg_s']), arm_pred_loss=arm)
    P = {k: v.requires_grad_(with_grads) for k, v in ow.hashed_state_dict(shapes, 0).items()}
    coords, feats = ovox.flatten_cameras([rs['%s_point_cloud' % c] for c in cams], [rs['%s_rgb' % c] for c in cams])
    from voxactb_amd import synthetic
    grid = ovox.voxelize(coords, feats, torch.tensor([synthetic.SCENE_BOUNDS]), V)
    assert torch.equal(grid, T(g['grid']))
    outs = operc.forward(P, grid.permute(0, 4, 1, 2, 3), rs['low_dim_state'], rs['lang_token_embs'],
                         depth=int(g['cfg_depth']), voxel_patch_stride=int(g['cfg_s']), arm_pred_loss=arm)
    assert float((outs[0].detach() - T(g['q_trans'])).abs().max()) < 2e-5
    assert float((outs[1].detach() - T(g['rot_grip'])).abs().max()) < 2e-5
    assert float((outs[2].detach() - T(g['collision'])).abs().max()) < 2e-5
    if arm:
        assert float((outs[3].detach() - T(g['arm_out'])).abs().max()) < 2e-5
    if with_grads:
        total, _ = oagent.losses(outs[0], outs[1], outs[2], rs['trans_action_indicies'], rs['rot_grip_action_indicies'],
                                 rs['ignore_collisions'], outs[3] if arm else None, rs.get('label'))
        assert abs(float(total) - float(g['loss'])) < 2e-5
        names = [str(n) for n in g['grad_names']]
        grads = torch.autograd.grad(total, [P[n] for n in names])
        ref_norms = T(g['grad_norms'])
        for n, gr, rn in zip(names, grads, ref_norms):
            assert abs(float(gr.norm()) - float(rn)) <= 2e-3 * float(rn) + 1e-6, n
            key = 'grad__' + n
            if key in g.files:
                ref = T(g[key])
                assert float((gr - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-6, n


def test_encoder_tiny(golden):
    _check_encoder(golden('f3_encoder_tiny'), ['front', 'wrist'])


def test_encoder_c1(golden):
    _check_encoder(golden('f3_encoder_c1'), ['front'])


def test_lamb(golden):
    g = golden('f7_lamb')
    for name in ('w_rand', 'w_zero', 'w_big'):
        w = T(g[name + '_w0'])
        m, v = torch.zeros_like(w), torch.zeros_like(w)
        for step in range(3):
            w, m, v, _ = oagent.lamb_step(w, T(g['%s_g%d' % (name, step)]), m, v)
            assert torch.equal(w, T(g[name + '_w'])[step]), (name, step)


def test_se3_fixture(golden):
    """F8: outputs of the REFERENCE's apply_se3_augmentation / _2Robots (augmentation.py:68-185, :187-348) for scripted draws
    (make_golden.py f8: shared and per-sample bounds, layer 0 / 1, a forced whole-batch retry, two arms)."""
    g = golden('f8_se3')
    for tag in [str(c) for c in g['cases']]:
        used, layer = int(g[tag + '_attempts']), int(g[tag + '_layer'])
        pcd = [T(g[tag + '_pcd0']), T(g[tag + '_pcd1'])]
        args = (pcd, T(g[tag + '_pose']), T(g[tag + '_rot_grip']), T(g[tag + '_bounds']))
        for a in range(used):
            ti, ri, pp, ok = ose3.augment(*args, T(g[tag + '_shift_unit'])[a], T(g[tag + '_rpy_steps'])[a], [0.125] * 3, 5, 100, 5, layer=layer)
            assert ok == (a == used - 1), (tag, a)
        assert torch.equal(ti.long(), T(g[tag + '_trans_idx']).long()) and torch.equal(ri.long(), T(g[tag + '_rot_grip_idx']).long())
        assert torch.equal(pp[0], T(g[tag + '_pcd0_out'])) and torch.equal(pp[1], T(g[tag + '_pcd1_out']))
        # rigid: pairwise distances preserved
        B = pcd[0].shape[0]
        a, b = pcd[0].reshape(B, 3, -1), pp[0].reshape(B, 3, -1)
        da = (a[:, :, :50, None] - a[:, :, None, :50]).norm(dim=1)
        db = (b[:, :, :50, None] - b[:, :, None, :50]).norm(dim=1)
        assert torch.allclose(da, db, atol=1e-5)
    # two arms, one perturbation; kept only when BOTH arms stay inside
    used = int(g['t_attempts'])
    pcd, bounds = [T(g['t_pcd0'])], T(g['t_bounds'])
    for a in range(used):
        r = ose3.augment(pcd, T(g['t_pose_right']), T(g['t_rot_grip_right']), bounds, T(g['t_shift_unit'])[a], T(g['t_rpy_steps'])[a], [0.125] * 3, 5, 100, 5)
        l = ose3.augment(pcd, T(g['t_pose_left']), T(g['t_rot_grip_left']), bounds, T(g['t_shift_unit'])[a], T(g['t_rpy_steps'])[a], [0.125] * 3, 5, 100, 5)
        assert (r[3] and l[3]) == (a == used - 1)
    assert torch.equal(r[0].long(), T(g['t_trans_idx_right']).long()) and torch.equal(r[1].long(), T(g['t_rot_grip_idx_right']).long())
    assert torch.equal(l[0].long(), T(g['t_trans_idx_left']).long()) and torch.equal(l[1].long(), T(g['t_rot_grip_idx_left']).long())
    assert torch.equal(r[2][0], T(g['t_pcd0_out']))


def test_depth_to_point_cloud_fixture(golden):
    """F10: clouds produced by PyRep's own pointcloud_from_depth_and_camera_params (compiled from the reference file by
    make_golden.py) for seeded depth buffers / cameras."""
    g = golden('f10_depth_clouds')
    for b in range(int(g['cfg_B'])):
        for c in range(int(g['cfg_ncam'])):
            tag = 'b%d_c%d_' % (b, c)
            mine, _ = ovox.depth_to_point_cloud(g[tag + 'depth01'], g[tag + 'ext'], g[tag + 'int'], float(g['near']), float(g['far']))
            assert np.array_equal(mine, g[tag + 'cloud'])
