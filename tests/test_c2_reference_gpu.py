"""GPU parity AT THE HEADLINE SIZES against digests the REFERENCE produced (tests/golden/make_golden.py, sections
f5 / f5g / f5c3 / f5v200): BASELINE.json configs[1] geometry (V=100, 4 cameras 128x128, depth 6, 2048 latents), the
configs[2] twin-agent shape (low_dim 7, arm-prediction head, per-sample crop bounds, B=2) and the configs[4] grid (V=200,
depth 6) -- each in the exact-'fp32' AND the default 'bf16x3' precision, i.e. both kernel families are pinned to the
reference itself, not to each other.

Checked per fixture: voxel occupancy list bit-exact + per-channel sums, q_trans argmax / top-16 / 4096 sampled logits /
log-sum-exp and every rot_grip / collision (/ arm) logit within 1e-4 (BASELINE.json north_star), and -- where the
reference's backward was captured -- the loss within 1e-4 and every parameter-gradient norm within 3e-3 relative."""
import numpy as np
import pytest
import torch

from oracle import weights as ow
from voxactb_amd import ops, synthetic
from voxactb_amd.agents.peract_bc.perceiver_lang_io import PerceiverVoxelLangEncoder
from voxactb_amd.voxel.voxel_grid import VoxelGrid

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_Q = 1e-4


def T(a):
    return torch.from_numpy(np.asarray(a))


def _setup(g):
    arm, crop = bool(g['cfg_arm']), bool(g['cfg_crop']) if 'cfg_crop' in g.files else False
    V, B, H, W = int(g['cfg_V']), int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W'])
    cams = [str(c) for c in g['cfg_cams']] if 'cfg_cams' in g.files else synthetic.CAMERAS4[:int(g['cfg_ncam'])]
    seed = int(g['cfg_seed']) if 'cfg_seed' in g.files else 1
    var = {k[len('cfg_var_'):]: (str(g[k]) if g[k].dtype.kind == 'U' else int(g[k])) for k in g.files if k.startswith('cfg_var_')}
    for k in [k for k in var if k != 'iterations' and not isinstance(var[k], str)]:      # iterations / ablation switches / fusion type of the fixture
        var[k] = bool(var[k])
    enc = PerceiverVoxelLangEncoder(
        depth=int(g['cfg_depth']), iterations=var.pop('iterations', 1), voxel_size=V, initial_dim=10, low_dim_size=int(g['cfg_low_dim']),
        num_latents=int(g['cfg_latents']), voxel_patch_size=int(g['cfg_k']), voxel_patch_stride=int(g['cfg_s']),
        activation='lrelu', input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0, arm_pred_loss=arm, **var)
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    enc = enc.to(DEV)
    rs = synthetic.make_replay_sample(B, cams, (H, W), V, int(g['cfg_low_dim']), seed=seed, arm_pred_loss=arm,
                                      crop_target_obj_voxel=crop)
    rs = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
    rs = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in rs.items()}
    bounds = rs['target_object_scene_bounds'] if crop else torch.tensor([synthetic.SCENE_BOUNDS])
    vg = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, H * W * len(cams))
    grid = vg.voxelize_cameras([rs['%s_point_cloud' % c].to(DEV) for c in cams], [rs['%s_rgb' % c].to(DEV) for c in cams],
                               bounds.to(DEV))
    return enc, rs, grid, arm, V, B


def _check_grid(g, grid):
    occ = grid[..., -1] > 0
    flat = torch.nonzero(occ.reshape(-1))[:, 0].int().cpu()
    ref = T(g['grid_occ_flat'])
    assert flat.numel() == int(g['grid_occ_count']) and torch.equal(flat, ref)          # occupancy indices: bit-exact
    sums = grid.double().sum(dim=(0, 1, 2, 3)).cpu()
    rs = T(g['grid_channel_sums'])
    assert float(((sums - rs).abs() / (rs.abs() + 1.0)).max()) < 1e-9, (sums, rs)         # same floats, summed in double


def _check_forward(g, outs, arm, tag):
    B = outs[0].shape[0]
    flat = outs[0].reshape(B, -1).float().cpu()
    assert torch.equal(flat.argmax(1), T(g['q_trans_argmax'])), tag
    sidx = T(g['q_trans_sample_idx']).long()
    e_s = float((flat[:, sidx] - T(g['q_trans_sample'])).abs().max())
    tv, ti = T(g['q_trans_top_vals']), T(g['q_trans_top_idx']).long()
    e_t = float((torch.gather(flat, 1, ti) - tv).abs().max())                            # reference's top-16 positions
    mine_top = flat.topk(16, dim=1)
    e_tv = float((mine_top.values - tv).abs().max())                                       # and our own top-16 values
    e_l = float((torch.logsumexp(flat.double(), 1) - T(g['q_trans_lse'])).abs().max())
    e_r = float((outs[1].float().cpu() - T(g['rot_grip'])).abs().max())
    e_c = float((outs[2].float().cpu() - T(g['collision'])).abs().max())
    print('%s: q_trans samples %.2e top16 %.2e/%.2e lse %.2e | rot_grip %.2e collision %.2e (|q|max %.2f)'
          % (tag, e_s, e_t, e_tv, e_l, e_r, e_c, float(flat.abs().max())))
    assert max(e_s, e_t, e_tv, e_l, e_r, e_c) < TOL_Q, tag
    if arm:
        assert float((outs[3].float().cpu() - T(g['arm_out'])).abs().max()) < TOL_Q, tag
    return dict(q_samples=e_s, q_top=e_t, lse=e_l, rot_grip=e_r, collision=e_c)


KINK_KEYS = {'input_preprocess': 'd0', 'patchify': 'patch', 'up0.conv_up.0': 'z1', 'up0.conv_up.2': 'u0', 'final': 'u'}


KINK_FLIP_FRACTION = 5e-5       # of a site's elements: measured <= 2.2e-5 (2 779 of u0's 1.28 x 10^8 at B = 2 behind the single-fp16 attention forward)
KINK_FLIP_MARGIN = 3e-5         # a flipped element of the PRODUCT lies within tau + this of zero (forward difference to the reference <= ~2e-5)


def force_kinks(cache, g):
    """The loss is piecewise smooth: LeakyReLU' jumps from 0.02 to 1 where a pre-activation crosses zero, and a forward arithmetic that is
    1e-5 away from the reference's moves ~100 of the 64 M elements of u0 across (tools/experiments/fwd_sensitivity_gpu.py; DESIGN.md 5r5:
    those elements alone carry the 3-8 % gradient differences of the 'forward-sensitive' batches -- parameter gradients are cancelling sums
    over 10^6 voxels).  Fixtures that carry the reference run's pre-activations within 3e-5 of zero (make_golden.py: capture_kinks) let the
    backward be evaluated at the SAME subgradient choices, exactly as the max-pool arg-maxima are: the saved activation is given the
    reference's sign there (magnitude 1e-30: these values are < 3e-5 anyway).  Returns the number of choices that differed.

    What the forcing may NOT hide (round-5 advisor): a real forward or mask error at near-zero elements.  Asserted per site: the elements
    whose sign differs are a vanishing fraction of the tensor (KINK_FLIP_FRACTION) and the product's OWN value there is within
    tau + KINK_FLIP_MARGIN of zero, i.e. the two forwards agree to ~1e-5 at every patched element; every listed element (flipped or not)
    is within 1e-3 of zero (same layout, same elements)."""
    flips = 0
    tau = float(g['kink_tau'])
    for site, key in KINK_KEYS.items():
        k = 'kink__%s__idx' % site
        if k not in g.files:
            continue
        idx = T(g[k]).long().to(DEV)
        pos = T(g['kink__%s__pos' % site]).to(DEV) > 0
        t = cache[key]
        assert t.is_contiguous()
        flat = t.view(-1)
        cur = flat[idx]
        assert idx.numel() == 0 or float(cur.abs().max()) < 1e-3, (site, float(cur.abs().max()))      # (same layout, same elements)
        diff = (cur > 0) != pos
        n = int(diff.sum())
        if n:
            assert n <= max(4, KINK_FLIP_FRACTION * flat.numel()), (site, n, flat.numel())
            # LeakyReLU(x) = 0.02 x below zero: the saved activation of a negative pre-activation x is 0.02 x -- bound |x|
            worst = float(torch.where(cur[diff] > 0, cur[diff], cur[diff] / 0.02).abs().max())
            assert worst <= tau + KINK_FLIP_MARGIN, (site, n, worst)
        flips += n
        flat[idx] = torch.where(pos, torch.full_like(cur, 1e-30), torch.full_like(cur, -1e-30))
    return flips


POOL_KEYS = (('ss0', 'd0'), ('ss1', 'z'), ('ss2', 'u'))
POOL_TIE = 5e-5                 # the product's value at the reference's arg-max voxel is within this of the product's own maximum
POOL_MAX_FLIPS = 4              # per pool, of B x C choices (measured: at most 2)


def force_pools(cache, gp, T0):
    """Global max pools (perceiver_lang_io.py:360, :451, :470): the backward at the reference run's arg-max voxels (fixture supplement
    *_pools.npz, make_golden.py: pool_choices), as tests/test_grad_noise_gpu.py does at the float64 run's.  A choice may only be replaced
    where it IS a tie: the reference's top-2 margin there is below POOL_TIE and the product's own value at the reference's voxel is within
    POOL_TIE of the product's maximum -- a forward error would fail these asserts instead of being patched over.  T0: language tokens in
    front of the grid tokens of z.  Returns the flips per pool."""
    out = []
    for i, (key, src) in enumerate(POOL_KEYS):
        ss, mx, st, am = cache[key]
        ref = T(gp['pool_argmax_%d' % i]).to(DEV).int().reshape(am.shape).contiguous()
        diff = am != ref
        n = int(diff.sum())
        if n:
            assert n <= POOL_MAX_FLIPS, (key, n)
            margin = T(gp['pool_margin_%d' % i]).to(DEV).reshape(am.shape)
            assert float(margin[diff].max()) < POOL_TIE, (key, float(margin[diff].max()))       # the reference's own top two tie there
            t = cache[src]
            B, C = am.shape
            x = t.reshape(B, -1, C)
            if src == 'z':
                x = x[:, T0:]
            b, c = torch.nonzero(diff, as_tuple=True)
            mine_at_ref = x[b, ref[b, c].long(), c]
            gap = float((mx[b, c] - mine_at_ref).abs().max())
            assert gap < POOL_TIE, (key, n, gap)
            cache[key] = (ss, mx, st, ref)
        out.append(n)
    return out


def align_choices(cache, ref_cache, T0, tie=1e-4):
    """Evaluate `cache`'s backward at the LeakyReLU / max-pool choices of ANOTHER run of the product on the same batch (`ref_cache`, e.g. the
    exact-fp32 kernels'): where the two saved activations have different signs and both are within `tie` of zero, `cache` takes the other
    run's sign; where the pool arg-maxima differ and the values tie within `tie`, the other run's voxel.  Same asserts as force_kinks /
    force_pools: a vanishing fraction of elements, every one a genuine near-tie.  Returns (LeakyReLU flips, pool flips)."""
    flips = 0
    for key in KINK_KEYS.values():
        a, b = cache[key].view(-1), ref_cache[key].view(-1)
        diff = (a > 0) != (b > 0)
        idx = torch.nonzero(diff)[:, 0]
        n = int(idx.numel())
        if n:
            assert n <= max(4, KINK_FLIP_FRACTION * a.numel()), (key, n, a.numel())
            va, vb = a[idx], b[idx]
            worst = max(float(torch.where(va > 0, va, va / 0.02).abs().max()), float(torch.where(vb > 0, vb, vb / 0.02).abs().max()))
            assert worst <= tie, (key, n, worst)
            a[idx] = torch.where(vb > 0, torch.full_like(va, 1e-30), torch.full_like(va, -1e-30))
        flips += n
    pools = []
    for key, src in POOL_KEYS:
        ss, mx, st, am = cache[key]
        ref = ref_cache[key][3]
        diff = am != ref
        n = int(diff.sum())
        if n:
            assert n <= POOL_MAX_FLIPS, (key, n)
            B, C = am.shape
            x = cache[src].reshape(B, -1, C)
            if src == 'z':
                x = x[:, T0:]
            b, c = torch.nonzero(diff, as_tuple=True)
            assert float((mx[b, c] - x[b, ref[b, c].long(), c]).abs().max()) < tie, key
            cache[key] = (ss, mx, st, ref)
        pools.append(n)
    return flips, pools


def _loss_and_backward(g, enc, eng, rs, outs, cache, arm, V, B, tag, gate=1.0, gp=None, forced=True, forced_pools=None):
    at = rs['trans_action_indicies'].long()
    lab = ((at[:, 0] * V + at[:, 1]) * V + at[:, 2]).int().to(DEV)
    dq = torch.empty((B, V ** 3), device=DEV)
    l_t, _, _ = ops.ce_big(outs[0].view(B, -1), lab, dq, 1.0 / B)
    labs = torch.cat([rs['rot_grip_action_indicies'].int(), rs['ignore_collisions'].int()[:, :1]], 1).to(DEV).contiguous()
    d_o = torch.empty_like(cache['o'])
    l_h, _ = ops.ce_rows(cache['o'], [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)], labs, d_o, 1.0 / B)
    total = l_t + l_h.sum(1)
    d_arm = None
    if arm:
        d_arm = torch.empty_like(outs[3])
        la, _ = ops.ce_rows(outs[3], [(0, 2)], rs['label'].int()[:, :1].to(DEV).contiguous(), d_arm, 1.0 / B)
        total = total + la[:, 0]
    loss = float(total.mean())
    assert abs(loss - float(g['loss'])) < 1e-4, (tag, loss, float(g['loss']))
    for p in enc.parameters():
        p.grad = None
    if forced_pools is None:
        forced_pools = forced
    if not forced:
        gp = gp if forced_pools else None
    elif eng.precision == 'fp32' and not eng.bwd_precision:
        # exact-fp32 kernels: no forcing at all (round-5 advisor) -- their forward is within ~1e-6 of the reference's and every digest
        # holds its gates at the product's OWN LeakyReLU / max-pool choices (20 fixtures x modes, profiles/r06_unforced_fp32.log)
        gp = None
    elif 'kink_tau' in g.files:
        print('%s: LeakyReLU choices near zero that differ from the reference run\'s (backward evaluated at the reference\'s): %d' % (tag, force_kinks(cache, g)))
    if gp is not None:
        print('%s: max-pool choices that differ from the reference run\'s (ties; backward evaluated at the reference\'s): %s' % (tag, force_pools(cache, gp, eng.T0)))
    eng.backward(cache, dq, d_o, d_arm)
    P = dict(enc.named_parameters())
    for prm in P.values():            # (a block an ablation leaves unused has no gradient: the reference's .grad is None there, the fixture holds zeros)
        if prm.grad is None:
            prm.grad = torch.zeros_like(prm)
    bad, worst, ratio = [], 0.0, 0.0
    for n, rn in zip([str(n) for n in g['grad_names']], T(g['grad_norms'])):
        gn, rn = float(P[n].grad.norm()), float(rn)
        key64 = 'dysum64__' + n
        if key64 in g.files:
            # bias of a grid conv = sum of dY over up to 10^6 voxels per channel.  The fixture also holds the reference's dY
            # added up in FLOAT64 (a backward hook in make_golden.py); the reference's own fp32 sum is off from that by
            # 5.8e-3 (final), 4.4e-3 (up0.conv_up.2), 1.2e-3 (input_preprocess) at this size, and its trans_decoder bias
            # gradient (mathematically sum(softmax - onehot) = 0) is 5.9e-5 of pure rounding noise -- so the float64 sums
            # are the yardstick for these six tensors, not the fp32 ones
            ref = T(g[key64])
            e = float((P[n].grad.double().cpu() - ref).abs().max())
            ref32 = float((T(g['grad__' + n]).double() - ref).abs().max()) if ('grad__' + n) in g.files else float('nan')
            print('   %-34s |ours - f64 sum| %.2e   |reference fp32 - f64 sum| %.2e   (max |grad| %.2e)' % (n, e, ref32, float(ref.abs().max())))
            # (gate set at 10^6 voxels.  A conv bias gradient is a sum over V^3 voxels of terms that largely cancel; the noise of such a sum
            # grows with the count -- taken as its square root: x 2.83 at V = 200.  Measured there: the reference's OWN fp32 sums are off
            # by up to 4.0e-3 (final; 5.8e-3 and 1.4e-2 on the V = 100 fixtures), ours by at most 2.0e-3 (input_preprocess -- the same
            # 2.0e-3 from the exact-fp32 kernels and from the fused default-precision ones, two unrelated kernel families; on the V = 100
            # fixtures that tensor is at 3.8e-4 / 1.3e-3 with the reference's fp32 at 1.2e-3 / 1.7e-3))
            size = max(1.0, (V / 100.0) ** 1.5) * gate
            ratio = max(ratio, e / (size * (2e-3 * float(ref.abs().max()) + 2e-5)) * gate)
            if forced and e > size * (2e-3 * float(ref.abs().max()) + 2e-5):
                bad.append((n, 'vs float64 dY sum', e, float(ref.abs().max())))
            continue
        rel = abs(gn - rn) / (rn + 1e-12)
        ratio = max(ratio, abs(gn - rn) / (3e-3 * rn + 1e-5))
        if abs(gn - rn) > gate * 3e-3 * rn + 1e-5:
            bad.append((n, gn, rn))
        elif rn > 1e-4:
            worst = max(worst, rel)
        key = 'grad__' + n
        if key in g.files:
            ref = T(g[key])
            e = float((P[n].grad.float().cpu() - ref).abs().max())
            ratio = max(ratio, e / (3e-3 * float(ref.abs().max()) + 1e-5))
            if forced and e > gate * 3e-3 * float(ref.abs().max()) + 1e-5:
                bad.append((n, 'full', e, float(ref.abs().max())))
    print('%s: loss %.6f (reference %.6f), worst grad-norm rel. error %.2e, worst tensor at %.2f x its gate%s' % (
        tag, loss, float(g['loss']), worst, ratio, '' if forced else ' (UN-FORCED: the product\'s own LeakyReLU / max-pool choices)'))
    assert not bad, (tag, bad)
    return ratio


def _pools_of(g):
    """the *_pools.npz supplement of a gradient digest (the reference run's max-pool choices), found by the forward it belongs to"""
    import glob
    import os
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', '*_pools.npz'))):
        gp = np.load(f, allow_pickle=False)
        if 'q_trans_lse' in g.files and gp['q_trans_lse'].shape == g['q_trans_lse'].shape and np.array_equal(gp['q_trans_lse'], g['q_trans_lse']):
            return gp
    return None


def _run(g, precision, tag, backward, bwd_precision='', wgrad_precision=None, gate=1.0, forced=True, forced_pools=None):
    enc, rs, grid, arm, V, B = _setup(g)
    _check_grid(g, grid)
    eng = enc.engine()
    eng.precision = precision
    eng.bwd_precision = bwd_precision
    if wgrad_precision is not None:
        eng.wgrad_precision = wgrad_precision
    tag = tag + ('|bwd ' + bwd_precision if bwd_precision else '') + ('|wgrad ' + wgrad_precision if wgrad_precision else '')
    outs, cache = eng.forward(grid, rs['low_dim_state'].to(DEV), rs['lang_token_embs'].to(DEV), training=False, save=backward,
                              lang_goal_emb=rs['lang_goal_emb'].to(DEV))
    errs = _check_forward(g, outs, arm, '%s/%s' % (tag, precision))
    if backward:
        errs['worst_x_gate'] = _loss_and_backward(g, enc, eng, rs, outs, cache, arm, V, B, '%s/%s' % (tag, precision), gate=gate,
                                                  gp=_pools_of(g), forced=forced, forced_pools=forced_pools)
    del cache, outs
    torch.cuda.empty_cache()
    return errs


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_c2_forward_digest_f5(golden, precision):
    """V=100, depth 6, 2048 latents, B=1 -- forward (reference perceiver_lang_io.py:345-485)."""
    _run(golden('f5_encoder_c2_digest'), precision, 'f5', backward=False)


WIDE = pytest.mark.parametrize('precision,wide_dispatch', [('fp32', False), ('bf16x3', False), ('bf16x3', True)],
                               indirect=['wide_dispatch'], ids=['fp32', 'bf16x3', 'bf16x3-wide'])


@WIDE
def test_c2_forward_backward_digest(golden, precision, wide_dispatch):
    """the same configuration with the reference's loss and per-parameter gradient norms (fwd + bwd of the reference).  'bf16x3-wide':
    the same fixture through the kernels the B = 16 headline dispatches (conftest.wide_dispatch)."""
    _run(golden('f5g_encoder_c2_grads'), precision, 'f5g' + ('|wide' if wide_dispatch else ''), backward=True)


@WIDE
def test_c3_twin_agent_shape_digest(golden, precision, wide_dispatch):
    """BASELINE.json configs[2] shape of one twin agent: low_dim 7, arm head, per-sample crop bounds, B=2, fwd + bwd."""
    _run(golden('f5c3_encoder_c3_digest'), precision, 'f5c3' + ('|wide' if wide_dispatch else ''), backward=True)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_c2_b8_forward_backward_digest_at_the_headline_dispatch(golden, precision):
    """configs[1] geometry at B = 8 (fixture f5gb8, round 5: forward + backward of the reference, 16 384 rows per linear layer): the row
    count from which ops.WIDE_MIN_M sends the linear layers to gemm_wide<0> / gemm_wide<1> (fp16x2 data gradient) / wgrad_wide_f16 and the
    fused GEGLU epilogue WITHOUT any test switch -- the dispatch of bench.py's B = 16 headline, pinned to the reference
    (qattention_peract_bc_agent.py:418-641, perceiver_lang_io.py:74-132)."""
    g = golden('f5gb8_encoder_c2_b8_grads')
    assert int(g['cfg_B']) * int(g['cfg_latents']) >= ops.WIDE_MIN_M
    _run(g, precision, 'f5gb8', backward=True)


UNFORCED_NORM_GATE = 0.10 / 3e-3        # x the 3e-3 norm gate = 10 %: the un-forced bound of rounds 4 - 5 (test_grad_noise_gpu.py: wu < 0.10)


@pytest.mark.parametrize('fixture', ['f5g_encoder_c2_grads', 'f5c3_encoder_c3_digest', 'f5gb8_encoder_c2_b8_grads'])
def test_unforced_choices_on_the_headline_fixtures(golden, fixture):
    """On record (round-5 advisor: "keep one unforced reference-gate run per headline fixture"): the default precision's backward at the
    product's OWN LeakyReLU choices against the reference's gradients -- loss within 1e-4, every gradient NORM within 10 %; element-wise
    differences printed, not gated.  The loss is piecewise smooth and a forward that is ~2e-5 from the reference's lands a few thousand of
    10^8 near-zero pre-activations on the other side (3 165 on F5c3); parameter gradients are cancelling sums over 10^6 voxels of
    heavy-tailed terms (SpatialSoftmax3D's 1 / 0.01) in which single elements weigh percents.  The max-pool choices stay at the
    reference's where they tie (force_pools asserts the ties): on F5c3 two voxels of one pool are 1.1e-6 apart in the reference's OWN run,
    which way that coin falls moves `cross_attend_blocks.0.norm_context.weight`'s gradient norm by 38 % in ANY arithmetic that is not
    bit-identical to the reference's (printed below as the fully un-forced line).  How often and how far over 32
    batches: tools/experiments/kink_statistics_gpu.py, profiles/r06_kink_statistics.log."""
    g = golden(fixture)
    # F5c3 is the batch round 3 called ill-conditioned: at its own LeakyReLU choices (3 165 of 4 x 10^8 differ from the reference run's) the
    # norm of cross_attend_blocks.0.norm_context.weight's gradient is 38 % off and lang_preprocess.weight's 16 %, with the pools at the
    # reference's -- recorded, not gated; the other two headline fixtures hold 10 % (measured: 3.1e-3 on the B = 8 one)
    gate = 1e9 if fixture.startswith('f5c3') else UNFORCED_NORM_GATE
    r = _run(g, 'bf16x3', fixture[:5] + '|own-kinks', backward=True, gate=gate, forced=False, forced_pools=True)
    print('%s at its own LeakyReLU choices: worst tensor (norm, element or float64 bias sum) at %.1f x the regular gate' % (fixture, r['worst_x_gate']))
    try:
        r2 = _run(g, 'bf16x3', fixture[:5] + '|own-kinks-and-pools', backward=True, gate=1e9, forced=False, forced_pools=False)
        print('%s fully un-forced: worst tensor at %.1f x the regular gate' % (fixture, r2['worst_x_gate']))
    except AssertionError as e:           # (recorded, not gated)
        print('%s fully un-forced: %s' % (fixture, str(e)[:300]))


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_c5_v200_forward_digest(golden, precision):
    """BASELINE.json configs[4] grid: V=200 (40^3 patches, 64 077 context tokens), depth 6, 2048 latents, forward."""
    _run(golden('f5v200_encoder_c5_digest'), precision, 'f5v200', backward=False)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_c5_v200_forward_backward_digest(golden, precision):
    """The same grid with the reference's loss and backward (fixture f5v200g, round 4: the reference's stride-1 convs evaluated in slabs
    of eight output depths -- its backward at 200^3 did not finish in 75 minutes as one ATen op per layer and takes ~15 minutes this
    way): loss within 1e-4, every parameter-gradient norm within 3e-3, the small tensors element-wise, the conv bias gradients against
    the float64 sums of the reference's dY."""
    _run(golden('f5v200g_encoder_c5_grads'), precision, 'f5v200g', backward=True)


@pytest.mark.parametrize('fixture', ['f3v_encoder_tiny_iterations2', 'f3v_encoder_c1_iterations3', 'f3v_encoder_c1_no_language',
                                     'f3v_encoder_c1_no_skip_connection', 'f3v_encoder_c1_no_perceiver', 'f3v_encoder_c1_pos_encoding_grid_only',
                                     'f3v_encoder_c1_lang_concat', 'f3v_encoder_c1_weight_tie_layers'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_encoder_switches_reachable_from_the_configs(golden, fixture, precision):
    """`transformer_iterations` > 1 (the cross-attention block and the self-attention stack run again over the SAME weights, perceiver
    :429-437; gradients of the shared weights and of the context add up) and the ablations `no_language` (:374-376), `no_skip_connection`
    and `no_perceiver` (:457-460: `final` reads u0 / d0 alone; under no_perceiver the up-block gets no gradient) -- PERACT_BC.yaml /
    launch_utils.py:744-774 -- against the reference's forward + backward, in the exact-fp32 AND the default precision (round 5: the
    single-source `final` weight / data gradients, the summed context gradients, shared weights hit several times per step with their
    per-use delayed scales).  Both are held to this file's gates, the backward evaluated at the reference run's LeakyReLU choices
    (the fixtures carry them since round 5: without them lang_concat's small tensors are 2 % off in the default precision -- kinks, not
    arithmetic).  (Rounds 4 - 5 ran no_skip_connection at 1.5 x the element gate in the default precision: ONE element of the translation
    head's weight gradient, 64 x 27 values on single fp16 products over 32 768 voxels, sat at 1.04 x.  Since round 6 that weight gradient
    takes the fp16 kernel from 2^19 voxels on and the bf16x3 one below -- every fixture at the regular gate.)"""
    gate = 1.0
    _run(golden(fixture), precision, fixture[4:], backward=True, gate=gate)


@pytest.mark.parametrize('fixture', ['f5v50a_encoder_release_digest', 'f5v50b_encoder_release_digest'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_released_recipe_geometry_digest(golden, fixture, precision):
    """The recipe VoxAct-B releases (peract/scripts/train_open_jar_ours_vlm_10_demos_v2_11_acting.sh:8-36, launch_utils.py:738-743): V = 50
    (6 x 8 + 2: the part-tile paths of every halo kernel), cameras front | wrist | wrist2, replay batch 1 (M = 2048 rows: the 128^2
    linear kernels), proprioception 7 and 8, arm loss, crop bounds -- forward digest and the backward's norm / element gates of this file in
    both precisions (round 5: with the reference run's LeakyReLU choices in the fixtures the default precision holds them too -- worst
    gradient-norm error 2.8e-4; v50b was one of round 4's 'forward-sensitive' batches)."""
    _run(golden(fixture), precision, fixture[:6], backward=True)


@pytest.mark.parametrize('fixture', ['f5g_encoder_c2_grads', 'f5c3_encoder_c3_digest'])
@pytest.mark.parametrize('fwd,bwd,wgrad', [('fp32', 'bf16x3', 'bf16x3'), ('fp32', 'bf16x3', 'fp16'), ('bf16x3', '', 'bf16x3')])
def test_forward_and_backward_precisions_separately(golden, fixture, fwd, bwd, wgrad):
    """which pass the gradient differences of the default precision come from: an exact-fp32 FORWARD followed by the bf16x3
    BACKWARD (conv / linear products; the attention core of that cache is the unfused fp32 one), with the conv weight gradients
    as the bf16x3 triple or as single fp16 products, and the all-bf16x3 pair without the fp16 weight gradients -- every
    combination holds the same gates against the reference's gradients as the default (3e-3 on norms and on the small tensors
    in full).  Measured (tools/experiments/emu_precision.py): worst gradient-norm error 9e-5 (fp32 / fp32),
    9e-5 (fp32 / bf16x3), 1.2e-4 (bf16x3 / bf16x3) on F5g -- the backward products are not what moves the gradients; a
    forward difference of 1e-5 is (SpatialSoftmax3D's 1 / 0.01 temperature, max-pool ties)."""
    _run(golden(fixture), fwd, fixture[:4], backward=True, bwd_precision=bwd, wgrad_precision=wgrad)
