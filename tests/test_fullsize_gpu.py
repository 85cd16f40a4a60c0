"""GPU, BASELINE.json config-2 geometry (100^3 voxels, 4 cameras 128x128, PerceiverIO depth 6 / 2048 latents) at B = 2:
size-independent properties of the whole path through the C ABI.  The reference cannot run here and the CPU oracle
needs ~6 s per sample at this size, so the checks are (a) two independent kernel families agreeing -- the default
'bf16x3' precision against the exact-fp32 matrix-core kernels -- inside the 1e-4 north-star bound on every Q head and
inside 4e-3 (max and L2, relative) on every parameter gradient with the backward evaluated at the exact-fp32 run's LeakyReLU / max-pool choices, see below, (b) run-to-run determinism, (c) occupancy bookkeeping
of the voxel grid that feeds it."""
import numpy as np
import pytest
import torch

from oracle import weights as ow
from voxactb_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
V, B, HW = 100, 2, 128
GRAD_GATE = 4e-3          # measured 2.0e-3 max / 1.2e-3 L2 (round 6, backward at the exact-fp32 run's choices); rounds 1 - 5: 1e-2 un-aligned


@pytest.fixture(scope='module')
def rig():
    from voxactb_amd import synthetic
    from voxactb_amd.agents.peract_bc import launch_utils as lu
    cfg = lu.default_cfg(method__voxel_sizes=[V], method__voxel_patch_size=5, method__voxel_patch_stride=5,
                         method__transformer_depth=6, method__num_latents=2048, replay__batch_size=B,
                         rlbench__camera_resolution=[HW, HW], ddp__num_devices=1)
    torch.manual_seed(77)
    agent = lu.create_agent(cfg)
    agent.build(training=True, device=0)
    qa = agent._pose_agent._qattention_agents[0]
    rs = {k: v.to(DEV) for k, v in synthetic.make_replay_sample(B, cfg.rlbench.cameras, (HW, HW), V, 4, seed=11).items()}
    pcd = [rs['%s_point_cloud' % c][:, 0] for c in cfg.rlbench.cameras]
    rgb = [(rs['%s_rgb' % c][:, 0].float() / 255.0) * 2.0 - 1.0 for c in cfg.rlbench.cameras]     # PreprocessAgent's normalisation
    grid = qa._q.voxelize([[r, p] for r, p in zip(rgb, pcd)], pcd, qa._coordinate_bounds.to(DEV))
    prop = rs['low_dim_state'][:, 0].float() if rs['low_dim_state'].dim() > 2 else rs['low_dim_state'].float()
    lang = rs['lang_token_embs'][:, 0].float() if rs['lang_token_embs'].dim() > 3 else rs['lang_token_embs'].float()
    return dict(agent=agent, qa=qa, eng=qa._q.encoder.engine(), grid=grid, prop=prop, lang=lang, pcd=pcd, rgb=rgb)


def test_voxel_grid_bookkeeping(rig):
    g = rig['grid']
    assert g.shape == (B, V, V, V, 10)
    occ = g[..., 9]
    assert set(torch.unique(occ).tolist()) <= {0.0, 1.0}
    n_occ = int(occ.sum())
    assert 0 < n_occ <= B * 4 * HW * HW
    # empty cells: zero features; every cell carries its own index / V in channels 6..8 (voxel_grid.py:192-198)
    assert float(g[..., :6][occ == 0].abs().max()) == 0.0
    idx = (torch.arange(V, dtype=torch.float32) / V).to(DEV)          # IEEE division on the host, as the reference's CPU path
    assert torch.equal(g[0, :, 0, 0, 6], idx) and torch.equal(g[1, 0, :, 0, 7], idx) and torch.equal(g[0, 0, 0, :, 8], idx)
    again = rig['qa']._q.voxelize([[r, p] for r, p in zip(rig['rgb'], rig['pcd'])], rig['pcd'], rig['qa']._coordinate_bounds.to(DEV))
    assert torch.equal(again, g)                                    # deterministic, bit for bit


def _run(eng, rig, mode, backward, choices_of=None, keep_cache=False):
    eng.precision = mode
    outs, cache = eng.forward(rig['grid'], rig['prop'], rig['lang'], training=False, save=backward)
    outs = [o.float().clone() for o in outs[:3]]
    grads = None
    if backward and choices_of is not None:
        from tests.test_c2_reference_gpu import align_choices
        print('%s backward at the %s run\'s choices: LeakyReLU flips %d, max-pool ties %s' % ((mode, choices_of[0]) + align_choices(cache, choices_of[1], eng.T0)))
    if backward:
        arena = rig['qa']._arena
        arena.zero_grad()
        torch.manual_seed(5)
        dq = (torch.randn(B, V ** 3, device=DEV) * 1e-3).contiguous()
        d_o = torch.randn_like(cache['o']) * 1e-1
        eng.backward(cache, dq, d_o, None)
        grads = {n: p.grad.clone() for n, p in rig['qa']._q.named_parameters()}
    if keep_cache:
        return outs, grads, cache
    return outs, grads


def test_q_values_and_gradients_agree_across_kernel_families(rig):
    eng = rig['eng']
    keep = eng.precision
    try:
        o1, g1, c1 = _run(eng, rig, 'fp32', True, keep_cache=True)
        o3, g3 = _run(eng, rig, 'bf16x3', True, choices_of=('fp32', c1))
        del c1
        torch.cuda.empty_cache()
        o3b, _ = _run(eng, rig, 'bf16x3', False)
    finally:
        eng.precision = keep
    for a, b in zip(o3, o3b):
        assert torch.equal(a, b)                                    # run-to-run deterministic
    for name, a, b in zip(('q_trans', 'rot_grip', 'collision'), o3, o1):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max())), name      # the north-star bound, at full size
    worst = 0.0
    report = []
    for n in g1:
        ref = g1[n]
        den = float(ref.abs().max())
        if den == 0.0:
            assert float(g3[n].abs().max()) == 0.0, n
            continue
        err = float((g3[n] - ref).abs().max()) / den
        l2 = float((g3[n] - ref).norm() / ref.norm())
        worst = max(worst, err)
        report.append((err, l2, n))
    report.sort(reverse=True)
    print('\n'.join('%.2e max-rel  %.2e l2-rel  %s' % r for r in report[:12]))
    # 1e-5 differences in the forward flip the LeakyReLU mask / max-pool winner of the ~1e-5 fraction of activations that sit that close
    # to the kink; each flip changes a gradient term by O(1).  Rounds 1 - 5 bounded that at 1e-2; since round 6 the default precision's
    # backward is evaluated at the exact-fp32 run's choices (align_choices: every replaced choice is asserted to be a tie within 1e-4),
    # which leaves the arithmetic alone
    assert report[0][0] < GRAD_GATE and max(r[1] for r in report) < GRAD_GATE, report[0]
    assert worst > 0.0                                              # (two different kernel families really ran)


def test_largest_grid_forward_agrees_across_kernel_families():
    """BASELINE.json configs[4] geometry (200^3 voxels, 40^3 patches, 8.0e6 voxels per sample) at B = 1, forward only:
    the default 'bf16x3' kernels (LDS-halo convs, tap-list polyphase, fused attention, matrix-core Cout = 1 conv) against
    the exact-fp32 generic kernels -- 32-bit index arithmetic and tile edge handling at the maximum size."""
    from voxactb_amd import synthetic
    from voxactb_amd.agents.peract_bc import launch_utils as lu
    V2, HW2 = 200, 128
    cfg = lu.default_cfg(method__voxel_sizes=[V2], method__voxel_patch_size=5, method__voxel_patch_stride=5,
                         method__transformer_depth=2, method__num_latents=512, replay__batch_size=1,
                         rlbench__camera_resolution=[HW2, HW2], ddp__num_devices=1)
    torch.manual_seed(78)
    agent = lu.create_agent(cfg)
    agent.build(training=False, device=0)
    qa = agent._pose_agent._qattention_agents[0]
    rs = {k: v.to(DEV) for k, v in synthetic.make_replay_sample(1, cfg.rlbench.cameras, (HW2, HW2), V2, 4, seed=12).items()}
    pcd = [rs['%s_point_cloud' % c][:, 0] for c in cfg.rlbench.cameras]
    rgb = [(rs['%s_rgb' % c][:, 0].float() / 255.0) * 2.0 - 1.0 for c in cfg.rlbench.cameras]
    grid = qa._q.voxelize([[r, p] for r, p in zip(rgb, pcd)], pcd, qa._coordinate_bounds.to(DEV))
    assert grid.shape == (1, V2, V2, V2, 10) and 0 < int(grid[..., 9].sum()) <= 4 * HW2 * HW2
    prop = rs['low_dim_state'][:, 0].float()
    lang = rs['lang_token_embs'][:, 0].float()
    eng = qa._q.encoder.engine()
    keep = eng.precision
    outs = {}
    try:
        for mode in ('bf16x3', 'fp32'):
            eng.precision = mode
            o, _ = eng.forward(grid, prop, lang, training=False, save=False)
            outs[mode] = [t.float().clone() for t in o[:3]]
    finally:
        eng.precision = keep
    for name, a, b in zip(('q_trans', 'rot_grip', 'collision'), outs['bf16x3'], outs['fp32']):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max())), name
    assert outs['fp32'][0].shape[-3:] == (V2, V2, V2)


def test_v200_forward_backward_agrees_across_kernel_families():
    """BASELINE.json configs[4] grid WITH the backward pass (the reference's CPU backward at 200^3 does not finish on the build
    container -- 75 minutes inside one ATen op -- so there is no reference digest for it; F5v200 pins the forward): V = 200, depth 6,
    2048 latents, B = 1, the reference digest's own seeded batch and name-hashed weights.  The default precision (bf16x3, fp16 leaf
    gradients, LDS-halo / tap-list / fused-attention kernels) against the exact-fp32 generic kernels: loss within 1e-4, every
    parameter-gradient norm within 3e-3 -- the same bounds the F5g digest of the reference is held to at V = 100."""
    import numpy as np
    from tests import test_c2_reference_gpu as R
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'f5v200_encoder_c5_digest.npz'), allow_pickle=False)
    enc, rs, grid, arm, V2, B2 = R._setup(g)
    R._check_grid(g, grid)
    eng = enc.engine()
    res = {}
    for mode in ('fp32', 'bf16x3'):
        eng.precision = mode
        outs, cache = eng.forward(grid, rs['low_dim_state'].to(DEV), rs['lang_token_embs'].to(DEV), training=False, save=True)
        R._check_forward(g, outs, arm, 'f5v200/' + mode)                          # both forwards against the REFERENCE digest
        at = rs['trans_action_indicies'].long()
        lab = ((at[:, 0] * V2 + at[:, 1]) * V2 + at[:, 2]).int().to(DEV)
        from voxactb_amd import ops
        dq = torch.empty((B2, V2 ** 3), device=DEV)
        l_t, _, _ = ops.ce_big(outs[0].view(B2, -1), lab, dq, 1.0 / B2)
        labs = torch.cat([rs['rot_grip_action_indicies'].int(), rs['ignore_collisions'].int()[:, :1]], 1).to(DEV).contiguous()
        d_o = torch.empty_like(cache['o'])
        l_h, _ = ops.ce_rows(cache['o'], [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)], labs, d_o, 1.0 / B2)
        loss = float((l_t + l_h.sum(1)).mean())
        for p in enc.parameters():
            p.grad = None
        eng.backward(cache, dq, d_o, None)
        res[mode] = (loss, {n: float(p.grad.norm()) for n, p in enc.named_parameters()})
        assert all(np.isfinite(v) for v in res[mode][1].values())
        del cache, outs
        torch.cuda.empty_cache()
    assert abs(res['fp32'][0] - res['bf16x3'][0]) < 1e-4, (res['fp32'][0], res['bf16x3'][0])
    # (+ 3e-5: trans_decoder's bias gradient is sum(softmax - onehot) = 0 mathematically, 1e-5 of rounding noise in either family)
    bad = [(n, a, res['bf16x3'][1][n]) for n, a in res['fp32'][1].items() if abs(a - res['bf16x3'][1][n]) > 3e-3 * a + 3e-5]
    worst = max(abs(a - res['bf16x3'][1][n]) / a for n, a in res['fp32'][1].items() if a > 1e-4)
    print('V=200 fwd+bwd: loss %.6f / %.6f, worst gradient-norm difference fp32 vs default %.2e' % (res['fp32'][0], res['bf16x3'][0], worst))
    assert not bad, bad[:5]


def test_v200_batch8_update_steps():
    """BASELINE.json configs[4] per-GPU workload as a TRAINING step: V = 200, B = 8, depth 6, 2048 latents, SE(3) augmentation and
    dropout on, LAMB -- three update() calls through the agent stack in the default precision (the third one reuses a persistent
    voxel grid incrementally and the delayed fp16 scales): finite, decreasing-or-stable losses, every parameter moved, no NaN."""
    from voxactb_amd import synthetic
    from voxactb_amd.agents.peract_bc import launch_utils as lu
    V2, B2, HW2 = 200, 8, 128
    cfg = lu.default_cfg(method__voxel_sizes=[V2], method__voxel_patch_size=5, method__voxel_patch_stride=5,
                         method__transformer_depth=6, method__num_latents=2048, replay__batch_size=B2,
                         rlbench__camera_resolution=[HW2, HW2], ddp__num_devices=1)
    torch.manual_seed(79)
    agent = lu.create_agent(cfg)
    agent.build(training=True, device=0)
    qa = agent._pose_agent._qattention_agents[0]
    assert qa._q.encoder.engine().precision == 'bf16x3' and qa._transform_augmentation
    w0 = qa._arena.flat_w.clone()
    rs = {k: v.to(DEV) for k, v in synthetic.make_replay_sample(B2, cfg.rlbench.cameras, (HW2, HW2), V2, 4, seed=21).items()}
    losses = [float(agent.update(i, dict(rs))['total_losses']) for i in range(3)]
    print('V=200 B=8 update(): losses', losses)
    assert all(l == l and abs(l) < 1e3 for l in losses) and losses[2] < losses[0] + 1.0
    w1 = qa._arena.flat_w
    assert bool(torch.isfinite(w1).all())
    moved = [(n, float((p.data - w0[o:o + k].view_as(p)).abs().max())) for (n, p), (o, k) in zip(qa._q.named_parameters(), qa._arena.segments)]
    assert all(m > 0 for _, m in moved), [n for n, m in moved if m == 0][:5]
    del agent, qa
    torch.cuda.empty_cache()


def _train_curve(precision, steps, spike_at=None, B=2):
    """`steps` LAMB update() calls at BASELINE.json configs[1] geometry (V = 100, 4 cameras 128 x 128, depth 6, 2048 latents), B = 2,
    dropout and augmentation off, four batches in turn; spike_at: that step's loss (hence every gradient) is multiplied by 100."""
    import os
    from voxactb_amd.agents.peract_bc import launch_utils as lu
    os.environ['VOXACTB_PRECISION'] = precision
    try:
        cfg = lu.default_cfg(method__voxel_sizes=[100], method__voxel_patch_size=5, method__voxel_patch_stride=5, method__transformer_depth=6,
                             method__num_latents=2048, replay__batch_size=B, method__input_dropout=0.0, method__attn_dropout=0.0,
                             rlbench__cameras=synthetic.CAMERAS4, rlbench__camera_resolution=[128, 128])
        cfg.method.transform_augmentation.apply_se3 = False
        agent = lu.create_agent(cfg)
        enc = agent._pose_agent._qattention_agents[0]._perceiver_encoder
        enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
        agent.build(training=True, device=0)
    finally:
        del os.environ['VOXACTB_PRECISION']
    qa = agent._pose_agent._qattention_agents[0]
    assert qa._q.encoder.engine().precision == precision
    batches = [{k: v.to(DEV) for k, v in synthetic.make_replay_sample(B, synthetic.CAMERAS4, (128, 128), 100, 4, seed=300 + j).items()}
               for j in range(4)]
    curve = []
    for step in range(steps):
        qa._loss_weights = (100.0,) * 5 if step == spike_at else None
        r = agent.update(step, dict(batches[step % 4]))
        curve.append(float(r['total_losses']) / (100.0 if step == spike_at else 1.0))
    qa._loss_weights = None
    del agent
    torch.cuda.empty_cache()
    return np.array(curve)


def test_sixty_steps_at_headline_geometry_default_precision_tracks_exact_fp32():
    """Training equivalence AT SIZE (round 3 had it at V = 16 only): 60 LAMB steps at configs[1] geometry in the exact-fp32 mode and in the
    default precision (bf16x3 forward; single / double fp16 products with exact and DELAYED operand scales in the backward, pipelined
    fp16 attention backward) on identical batches.  Both curves fall, and they stay together."""
    a = _train_curve('fp32', 60)
    b = _train_curve('bf16x3', 60)
    print('fp32   ', np.round(a[::6], 3))
    print('default', np.round(b[::6], 3))
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert a[-8:].mean() < a[:8].mean() - 1.0 and b[-8:].mean() < b[:8].mean() - 1.0
    assert np.abs(a - b).max() < 0.05 * np.abs(a).max(), np.abs(a - b).max()


def test_gradient_spike_at_size_is_absorbed_by_the_delayed_scales():
    """The same run with every gradient multiplied by 100 at step 30 (loss weights): beyond the 32-fold headroom of the delayed fp16
    operand scales, so that step's weight gradients of the generic kernel saturate once and the scales re-centre on the next step.  The
    default-precision curve must follow the exact-fp32 curve of the SAME spiked run afterwards (LAMB normalises the step, so the spike
    itself is one ordinary-sized update in both arithmetics)."""
    a = _train_curve('fp32', 45, spike_at=30)
    b = _train_curve('bf16x3', 45, spike_at=30)
    print('fp32   ', np.round(a[24:45:2], 3))
    print('default', np.round(b[24:45:2], 3))
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert np.abs(a[:31] - b[:31]).max() < 0.05 * np.abs(a).max()
    assert np.abs(a[31:] - b[31:]).max() < 0.08 * np.abs(a).max(), np.abs(a[31:] - b[31:]).max()
