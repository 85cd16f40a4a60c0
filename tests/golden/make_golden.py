#!/usr/bin/env python
"""Generate tests/golden/*.npz by importing the REFERENCE's own Python modules.

Runs only in the build container (needs /root/reference, read-only).  Nothing from
the reference is copied: the fixtures hold inputs and the outputs the reference
computed for them.  Recipe = SURVEY.md Appendix B (path-load the numerical core;
MagicMock the simulator / viz imports for the agent-level oracle).

    python tests/golden/make_golden.py [--only f1,f3] [--skip-c2]

Each section also asserts that `oracle/` reproduces the reference on the same
inputs, i.e. generating the fixtures *is* the pinning of the oracle.
"""
import argparse
import importlib.util
import os
import sys
import time
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'
sys.path[:0] = [REF + '/peract', REF + '/YARR']

import numpy as np
import torch
import transformers  # noqa: F401  (must be imported BEFORE torchvision is stubbed)

from oracle import weights as ow, voxel_grid as ovox, perceiver as operc, agent as oagent, se3 as ose3
from voxactb_amd import synthetic


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, 'peract', rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


ref_vg = load('ref_voxel_grid', 'voxel/voxel_grid.py')
ref_pl = load('ref_perceiver', 'agents/peract_bc/perceiver_lang_io.py')
ref_lamb = load('ref_lamb', 'helpers/optim/lamb.py')


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


def ref_voxelize(coords, feats, bounds, V, B):
    vg = ref_vg.VoxelGrid(coord_bounds=bounds[0].tolist(), voxel_size=V, device='cpu', batch_size=B,
                          feature_size=feats.shape[-1], max_num_coords=coords.shape[1])
    return vg.coords_to_bounding_voxel_grid(coords, coord_features=feats, coord_bounds=bounds)


# ----------------------------------------------------------------------------- F1
def f1_voxel_kats():
    out = {}
    # KAT of SURVEY.md section 4: V=4, unit cube, 8 points, feature k/10
    pts = torch.tensor([[0.1, 0.1, 0.1], [0.1, 0.5, 0.75], [0.25, 0.5, 0.75], [0.2499999, 0.5, 0.75],
                        [0.6, 0.6, 0.6], [0.7, 0.7, 0.7], [1.0, 0.5, 0.5], [-0.01, 0.5, 0.5],
                        [0.99, 0.6, 0.3]]).unsqueeze(0)
    ft = (torch.arange(pts.shape[1]).float() / 10).view(1, -1, 1).repeat(1, 1, 3)
    bd = torch.tensor([[0., 0., 0., 1., 1., 1.]])
    g = ref_voxelize(pts, ft, bd, 4, 1)
    assert torch.equal(g, ovox.voxelize(pts, ft, bd, 4))
    out.update(kat_coords=pts, kat_feats=ft, kat_bounds=bd, kat_V=4, kat_grid=g)
    case = 0
    for V in (4, 8, 32):
        for B in (1, 3):
            rg = np.random.Generator(np.random.Philox(key=1000 + case))
            N = 257 if V < 32 else 3000
            lo = rg.uniform(-1, 0, (B, 3)).astype(np.float32)
            hi = lo + rg.uniform(0.5, 2.0, (B, 3)).astype(np.float32)
            per_sample = (case % 2 == 1)
            if not per_sample:
                lo[:], hi[:] = lo[0], hi[0]
            bd = torch.from_numpy(np.concatenate([lo, hi], 1))
            p = lo[:, None, :] + rg.uniform(-0.15, 1.15, (B, N, 3)).astype(np.float32) * (hi - lo)[:, None, :]
            res = (hi - lo) / np.float32(V)
            # duplicates, exact lattice points, exact upper/lower bounds, far away, NaN / inf
            p[:, 10:20] = p[:, 0:10]
            k = rg.integers(0, V + 1, (B, 30, 3)).astype(np.float32)
            p[:, 20:50] = lo[:, None, :] + k * res[:, None, :]
            p[:, 50] = hi
            p[:, 51] = lo
            p[:, 52] = 1e30
            p[:, 53] = -1e30
            p[:, 54, 0] = np.nan
            p[:, 55, 1] = np.inf
            p[:, 56:120] = lo[:, None, :] + (0.5 + 0.01 * rg.uniform(0, 1, (B, 64, 3)).astype(np.float32)) * (hi - lo)[:, None, :]
            f = rg.uniform(-1, 1, (B, N, 3)).astype(np.float32)
            p, f = torch.from_numpy(p), torch.from_numpy(f)
            bounds = bd if per_sample else bd[:1]
            g = ref_voxelize(p, f, bounds, V, B)
            o = ovox.voxelize(p, f, bounds, V)
            assert torch.equal(torch.nan_to_num(g), torch.nan_to_num(o)), (V, B)
            out['c%d_coords' % case], out['c%d_feats' % case] = p, f
            out['c%d_bounds' % case], out['c%d_V' % case], out['c%d_grid' % case] = bounds, V, g
            case += 1
    out['n_cases'] = case
    save('f1_voxel_kats', **out)


# ----------------------------------------------------------------------------- configs
CFG_TINY = dict(V=8, k=3, s=2, depth=1, latents=16, low_dim=4, B=2, cams=['front', 'wrist'], H=16, W=16)
CFG_C1 = dict(V=32, k=5, s=4, depth=1, latents=64, low_dim=4, B=1, cams=['front'], H=64, W=64)
CFG_UPD = dict(V=16, k=3, s=4, depth=2, latents=32, low_dim=7, B=3, cams=['front', 'wrist'], H=16, W=16)
CFG_C2 = dict(V=100, k=5, s=5, depth=6, latents=2048, low_dim=4, B=1, cams=synthetic.CAMERAS4, H=128, W=128)
# BASELINE.json configs[2] shape of ONE of the twin agents (acting / stabilizing): low_dim 7, arm-prediction loss,
# per-sample crop bounds (scripts/train_open_jar_ours_vlm_10_demos_v2_11_acting.sh:22-24); B = 2 so that the two samples
# really use different bounds rows
CFG_C3 = dict(V=100, k=5, s=5, depth=6, latents=2048, low_dim=7, B=2, cams=synthetic.CAMERAS4, H=128, W=128)
# BASELINE.json configs[4] shape: 200^3 voxels, depth 6 (forward digest only: the autograd graph of the reference at this
# size does not fit the build container's memory budget comfortably)
# the recipe VoxAct-B releases (scripts/train_open_jar_ours_vlm_10_demos_v2_11_acting.sh:8-36): V = 50, cameras front | wrist | wrist2,
# replay.batch_size = 1, twin-agent proprioception of 7 (8 with arm_id_to_proprio, launch_utils.py:738-743), arm loss, crop bounds
CFG_V50 = dict(V=50, k=5, s=5, depth=6, latents=2048, low_dim=7, B=1, cams=['front', 'wrist', 'wrist2'], H=128, W=128)
CFG_V50B = dict(CFG_V50, low_dim=8)
CFG_C5 = dict(V=200, k=5, s=5, depth=6, latents=2048, low_dim=4, B=1, cams=synthetic.CAMERAS4, H=128, W=128)


def make_ref_encoder(cfg, arm=False, seed=0):
    var = dict(cfg.get('variant', {}))          # encoder switches the configs can reach (launch_utils.py:744-774): transformer_iterations, ablations
    enc = ref_pl.PerceiverVoxelLangEncoder(
        depth=cfg['depth'], iterations=var.pop('iterations', 1), voxel_size=cfg['V'], initial_dim=10, low_dim_size=cfg['low_dim'],
        num_latents=cfg['latents'], voxel_patch_size=cfg['k'], voxel_patch_stride=cfg['s'],
        activation='lrelu', input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0, arm_pred_loss=arm, **var)
    shapes = {n: tuple(p.shape) for n, p in enc.named_parameters()}
    mine = operc.param_shapes(cfg['depth'], cfg['V'], cfg['low_dim'], num_latents=cfg['latents'],
                              voxel_patch_size=cfg['k'], voxel_patch_stride=cfg['s'], arm_pred_loss=arm)
    assert shapes == mine or cfg.get('variant'), set(shapes.items()) ^ set(mine.items())      # (ablations change `final`'s input width)
    sd = ow.hashed_state_dict(shapes, seed)
    enc.load_state_dict(sd, strict=False)
    return enc.eval(), sd


def batch_for(cfg, seed=0, arm=False, crop=False):
    rs = synthetic.make_replay_sample(cfg['B'], cfg['cams'], (cfg['H'], cfg['W']), cfg['V'], cfg['low_dim'],
                                      seed=seed, arm_pred_loss=arm, crop_target_obj_voxel=crop)
    # PreprocessAgent.update (preprocess_agent.py:23-32)
    rs = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
    rs = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in rs.items()}
    return rs


def enc_kw(cfg, arm=False):
    return dict(depth=cfg['depth'], voxel_patch_stride=cfg['s'], arm_pred_loss=arm)


KINK_SITES = ('input_preprocess', 'patchify', 'up0.conv_up.0', 'up0.conv_up.2', 'final')      # the grid-sized LeakyReLU blocks
KINK_TAU = 3e-5


def capture_kinks(enc, store, tau=KINK_TAU):
    """forward hooks on the conv of every grid-sized LeakyReLU block: flat channels-LAST indices ([B, D, H, W, C], the product's layout) of
    the pre-activations with |x| < tau and whether they are positive.  The loss is only piecewise smooth: LeakyReLU' jumps from 0.02 to 1
    at x = 0, so ANY forward arithmetic that moves such an x across zero (the default precision's forward is within ~1e-5 of fp32) evaluates
    another subgradient there -- and parameter gradients are heavily cancelling sums over 10^6 voxels in which one such element can weigh
    percents (tools/experiments/fwd_sensitivity_gpu.py: 95 of 64 M elements of u0 carry the whole 3-8 % of the 'forward-sensitive' batches).
    A test evaluates the product's backward at the SAME choices as the run the fixture's gradients come from."""
    hooks = []

    def make(site):
        def hook(mod, inp, out):
            x = out.detach()
            xl = x.permute(0, 2, 3, 4, 1).reshape(-1)
            idx = torch.nonzero(xl.abs() < tau)[:, 0]
            store['kink__%s__idx' % site] = idx.int()
            store['kink__%s__pos' % site] = (xl[idx] > 0).to(torch.uint8)
        return hook
    mods = dict(enc.named_modules())
    for site in KINK_SITES:
        hooks.append(mods[site + '.conv3d'].register_forward_hook(make(site)))
    return hooks


def encoder_fixture(name, cfg, arm=False, with_grads=True, digest=False, crop=False, check_oracle=True, f64_grads=False, kinks=False):
    enc, sd = make_ref_encoder(cfg, arm)
    rs = batch_for(cfg, seed=1, arm=arm, crop=crop)
    pcd = [rs['%s_point_cloud' % c] for c in cfg['cams']]
    rgb = [rs['%s_rgb' % c] for c in cfg['cams']]
    bounds = rs['target_object_scene_bounds'] if crop else torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    grid = ref_voxelize(coords, feats, bounds, cfg['V'], cfg['B'])
    if check_oracle:
        assert torch.equal(grid, ovox.voxelize(coords, feats, bounds, cfg['V']))
    ins = grid.permute(0, 4, 1, 2, 3).detach()
    t0 = time.time()
    for p in enc.parameters():
        p.requires_grad_(with_grads)
    dy_sums = {}
    hooks = []
    if f64_grads and with_grads:
        # bias gradients of the grid convs are sums of dY over up to 10^6 voxels per channel; the reference adds them up in
        # fp32.  The same dY values added up in float64 (a backward hook on every conv output) separate the reference's own
        # summation error from real differences.
        def tap(name):
            def fwd_hook(mod, inp, out):
                out.register_hook(lambda gr: dy_sums.__setitem__(name, gr.double().sum(dim=(0, 2, 3, 4))))
            return fwd_hook
        for n, m in enc.named_modules():
            if isinstance(m, torch.nn.Conv3d):
                hooks.append(m.register_forward_hook(tap(n)))
    kink_store = {}
    if kinks:
        hooks += capture_kinks(enc, kink_store)
    with torch.set_grad_enabled(with_grads):
        outs = enc(ins, rs['low_dim_state'], rs['lang_goal_emb'], rs['lang_token_embs'], None, bounds, None)
    print('%s: reference forward %.1fs' % (name, time.time() - t0))
    if kinks:
        print('%s: pre-activations within %.0e of zero: %s' % (name, KINK_TAU, {k.split('__')[1]: int(v.numel()) for k, v in kink_store.items() if k.endswith('idx')}))
    inter = None
    if check_oracle:
        with torch.no_grad():
            o_outs, inter = operc.forward({k: v for k, v in sd.items()}, ins, rs['low_dim_state'],
                                          rs['lang_token_embs'], return_intermediates=True, **enc_kw(cfg, arm))
        errs = [float((a.detach() - b).abs().max()) for a, b in zip(outs, o_outs)]
        print('%s: oracle vs reference max-abs %s' % (name, errs))
        assert max(errs) < 2e-5, errs
        if digest:
            inter = None
    arrs = dict(cfg_V=cfg['V'], cfg_k=cfg['k'], cfg_s=cfg['s'], cfg_depth=cfg['depth'], cfg_latents=cfg['latents'],
                cfg_low_dim=cfg['low_dim'], cfg_B=cfg['B'], cfg_H=cfg['H'], cfg_W=cfg['W'], cfg_ncam=len(cfg['cams']),
                cfg_cams=np.array(cfg['cams']), cfg_arm=int(arm), cfg_crop=int(crop), rot_grip=outs[1], collision=outs[2])
    for vk, vv in cfg.get('variant', {}).items():
        arrs['cfg_var_' + vk] = np.array(vv) if isinstance(vv, str) else int(vv)
    if arm:
        arrs['arm_out'] = outs[3]
    qt = outs[0].detach()
    if digest:
        flat = qt.reshape(cfg['B'], -1)
        top = flat.topk(16, dim=1)
        sidx = ow.hashed_int('digest', (4096,), 0, flat.shape[1])
        arrs.update(q_trans_argmax=flat.argmax(1), q_trans_top_vals=top.values, q_trans_top_idx=top.indices,
                    q_trans_sample_idx=sidx, q_trans_sample=flat[:, sidx], q_trans_sum=flat.double().sum(1),
                    q_trans_lse=torch.logsumexp(flat.double(), 1))
        occ = grid[..., -1] > 0
        arrs.update(grid_occ_count=occ.sum(), grid_channel_sums=grid.double().sum(dim=(0, 1, 2, 3)),
                    grid_occ_flat=torch.nonzero(occ.reshape(-1))[:, 0].int())
    else:
        arrs.update(q_trans=qt, grid=grid)
        for k in ('z', 'z1', 'latents_out', 'feats'):
            arrs['int_' + k] = inter[k]
        arrs['int_d0_sum'] = inter['d0'].double().sum()
        arrs['int_u0_sum'] = inter['u0'].double().sum()
        arrs['int_u_sum'] = inter['u'].double().sum()
    if with_grads:
        # same scalar the product tests differentiate: the A.4 loss with labels from the batch
        total, parts = oagent.losses(outs[0], outs[1], outs[2], rs['trans_action_indicies'],
                                     rs['rot_grip_action_indicies'], rs['ignore_collisions'],
                                     outs[3] if arm else None, rs.get('label'))
        total.backward()
        arrs['loss'] = total.detach()
        names = [n for n, _ in enc.named_parameters()]
        arrs['grad_names'] = np.array(names)
        for _, p in enc.named_parameters():          # (ablations leave unused blocks without a gradient: .grad is None there)
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        arrs['grad_norms'] = torch.stack([p.grad.norm() for _, p in enc.named_parameters()])
        for n, p in enc.named_parameters():
            if p.numel() <= 20000 and not n.startswith(('pos_encoding', 'latents')):
                arrs['grad__' + n] = p.grad
        for n, v in dy_sums.items():
            arrs['dysum64__' + n + '.bias'] = v
            print('%s: %s.bias  fp32-summed vs float64-summed dY: %.2e (max |grad| %.2e)' % (
                name, n, float((dict(enc.named_parameters())[n + '.bias'].grad.double() - v).abs().max()), float(v.abs().max())))
    for h in hooks:
        h.remove()
    arrs.update(kink_store)
    if kinks:
        arrs['kink_tau'] = KINK_TAU
    save(name, **arrs)



def _pool_hook(enc, pools):
    def hook(mod, inp, out):
        top = inp[0].detach().flatten(2).topk(2, dim=-1)
        pools.append((top.indices[..., 0].int(), (top.values[..., 0] - top.values[..., 1]).float()))
    return enc.global_maxp.register_forward_hook(hook)


def _save_pools(fixture, pools, lse_key, lse, extra=None):
    base = np.load(os.path.join(HERE, fixture + '.npz'), allow_pickle=False)
    assert np.array_equal(base[lse_key], lse.numpy()), 'not the forward the fixture holds'
    arrs = dict(q_trans_lse=lse)
    for i, (am, mg) in enumerate(pools):
        arrs['pool_argmax_%d' % i] = am
        arrs['pool_margin_%d' % i] = mg
    arrs.update(extra or {})
    print('%s: smallest top-2 margins of the global max pools: %s' % (fixture, [float(mg.min()) for _, mg in pools]), flush=True)
    save(fixture + '_pools', **arrs)


def pool_choices(fixture, cfg, arm=False, crop=False, seed=1, slabs=False):
    """Supplement of an encoder digest (same encoder, same batch, fp32, forward only; round 6): the arg-max voxel of every (sample,
    channel) of the three global max pools (perceiver_lang_io.py:360, :451, :470) and the margin between the largest and the second
    largest value.  The loss is only piecewise smooth: a pool whose two largest voxels are closer than a forward arithmetic's rounding
    (~1e-5) hands its whole gradient to the other voxel -- ONE such choice of 512 moves `decoder_cross_attn.fn.to_out.weight`'s gradient by
    4 % on the F5c3 batch (tools/experiments/attn_fwd_variants.py: every single-fp16 attention forward "failed" that digest by exactly
    this one flip, whatever the operand treatment; the two voxels are 1.1e-6 apart in the reference's run).  As for the LeakyReLU choices
    (capture_kinks) a test evaluates the product's backward at the SAME choices as the run the fixture's gradients come from, after
    checking that the product's own values tie there.  slabs: the stride-1 convs as _f5v200g evaluates them."""
    enc, sd = make_ref_encoder(cfg, arm)
    rs = batch_for(cfg, seed=seed, arm=arm, crop=crop)
    pcd = [rs['%s_point_cloud' % c] for c in cfg['cams']]
    rgb = [rs['%s_rgb' % c] for c in cfg['cams']]
    bounds = rs['target_object_scene_bounds'] if crop else torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    grid = ref_voxelize(coords, feats, bounds, cfg['V'], cfg['B'])
    ins = grid.permute(0, 4, 1, 2, 3).detach()
    pools = []
    h = _pool_hook(enc, pools)
    if slabs:
        torch.set_num_threads(8)
        torch.nn.functional.conv3d = _chunked_conv3d_f64
    try:
        with torch.no_grad():
            outs = enc(ins, rs['low_dim_state'], rs['lang_goal_emb'], rs['lang_token_embs'], None, bounds, None)
    finally:
        torch.nn.functional.conv3d = _ORIG_CONV3D
    h.remove()
    assert len(pools) == 3
    _save_pools(fixture, pools, 'q_trans_lse', torch.logsumexp(outs[0].reshape(cfg['B'], -1).double(), 1))


def pool_choices_2robots(fixture, cfg):
    """The same supplement for a 2Robots digest (encoder2_fixture's encoder and batch), plus the LeakyReLU choices near zero the F11
    fixtures never carried (capture_kinks): the round-5 review asked for them so that a default-precision difference on that digest can
    be told apart into kinks and arithmetic.  The left arm's ss_final pools the same tensor as the right arm's (perceiver :838-846)."""
    enc = ref_pl.PerceiverVoxelLang2RobotsEncoder(
        depth=cfg['depth'], iterations=1, voxel_size=cfg['V'], initial_dim=10, low_dim_size=cfg['low_dim'],
        num_latents=cfg['latents'], voxel_patch_size=cfg['k'], voxel_patch_stride=cfg['s'],
        activation='lrelu', input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0)
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    enc.eval()
    rs = batch_for(cfg, seed=1)
    B, V = cfg['B'], cfg['V']
    pcd = [rs['%s_point_cloud' % c] for c in cfg['cams']]
    rgb = [rs['%s_rgb' % c] for c in cfg['cams']]
    bounds = torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    grid = ref_voxelize(coords, feats, bounds, V, B)
    ins = grid.permute(0, 4, 1, 2, 3).detach()
    proprio_left = ow.hashed_uniform('f11.proprio_left', (B, cfg['low_dim']), 0.0, 1.0)
    pools, store = [], {}
    hooks = [_pool_hook(enc, pools)] + capture_kinks(enc, store)
    with torch.no_grad():
        outs = enc(ins, rs['low_dim_state'], proprio_left, rs['lang_goal_emb'], rs['lang_token_embs'], None, bounds, None)
    for h in hooks:
        h.remove()
    assert len(pools) in (3, 4), len(pools)
    if len(pools) == 4:
        assert torch.equal(pools[2][0], pools[3][0])          # ss_final and ss_final_left_arm pool the same tensor
    print('%s: pre-activations within %.0e of zero: %s' % (fixture, KINK_TAU, {k.split('__')[1]: int(v.numel()) for k, v in store.items() if k.endswith('idx')}))
    _save_pools(fixture, pools[:3], 'q_trans_right_lse', torch.logsumexp(outs[0].reshape(B, -1).double(), 1), dict(store, kink_tau=KINK_TAU))


def _f5v200g():
    """configs[4] geometry, forward + backward of the reference in fp32: stride-1 convs evaluated in slabs of eight output depths
    (_chunked_conv3d_f64 below: the module code is untouched), a stack dump every ten minutes so that a stuck ATen op is named."""
    import faulthandler
    faulthandler.dump_traceback_later(600, repeat=True)
    torch.set_num_threads(8)
    torch.nn.functional.conv3d = _chunked_conv3d_f64
    try:
        encoder_fixture('f5v200g_encoder_c5_grads', CFG_C5, with_grads=True, digest=True, check_oracle=False, f64_grads=True, kinks=True)
    finally:
        torch.nn.functional.conv3d = _ORIG_CONV3D
        faulthandler.cancel_dump_traceback_later()


# ----------------------------------------------------------------------------- F5n: the reference's own fp32-vs-fp64 noise floor
_ORIG_CONV3D = torch.nn.functional.conv3d


def _chunked_conv3d_f64(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """torch's float64 Conv3d on the CPU is the vol2col path: k^3 Cin columns per output voxel, 27.6 GB for the `final` conv at V = 100 (the
    float32 path is oneDNN and has no such buffer) -- the build container's 64 GB did not survive it.  Same arithmetic in slabs of eight
    output depths (the reference's module code is untouched: only the functional it calls is wrapped, for float64 inputs only)."""
    st = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
    # (float32 too from 150 voxels per side on: the reference's backward at 200^3 sat inside one ATen op for 75 minutes in round 3)
    big = input.dim() == 5 and (input.shape[2] >= 40 if input.dtype == torch.float64 else input.shape[2] >= 150)
    if st != (1, 1, 1) or not big or isinstance(padding, str):
        return _ORIG_CONV3D(input, weight, bias, stride, padding, dilation, groups)
    pd = (padding,) * 3 if isinstance(padding, int) else tuple(padding)
    k = weight.shape[2]
    if pd[0]:
        input = torch.nn.functional.pad(input, (0, 0, 0, 0, pd[0], pd[0]))
    d_out = input.shape[2] - k + 1
    return torch.cat([_ORIG_CONV3D(input[:, :, z0:min(d_out, z0 + 8) + k - 1], weight, bias, 1, (0, pd[1], pd[2]), dilation, groups)
                      for z0 in range(0, d_out, 8)], 2)


def grad_noise_fixture(name, cfg, seed, arm=False, crop=False, nproj=16):
    """The reference encoder run TWICE on one batch: in fp32 (what F5g / F5c3 hold) and in float64 (the truth both the reference's fp32
    arithmetic and the product approximate).  Stored per parameter tensor: ||g64||, ||g32 - g64|| (the reference's own rounding noise,
    exact), `nproj` random +-1 projections of g64 (oracle/weights.py: projection_signs -- the GPU test estimates ||g - g64|| of ITS
    gradient from the same projections without the 133 MB tensor) and, for small tensors, g64 in full; the Q-value digest of the fp64
    forward with |q32 - q64|.  A gradient gate of "k x the reference's own spread" replaces per-fixture hand-set tolerances where the
    reference itself is ill-conditioned (a global max-pool tie hands a whole gradient to another voxel under a 1e-6 perturbation)."""
    enc, sd = make_ref_encoder(cfg, arm)
    rs = batch_for(cfg, seed=seed, arm=arm, crop=crop)
    pcd = [rs['%s_point_cloud' % c] for c in cfg['cams']]
    rgb = [rs['%s_rgb' % c] for c in cfg['cams']]
    bounds = rs['target_object_scene_bounds'] if crop else torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    grid = ref_voxelize(coords, feats, bounds, cfg['V'], cfg['B'])
    ins = grid.permute(0, 4, 1, 2, 3).detach()
    res = {}
    pools = []          # arg-max voxel of every (sample, channel) of the three global max pools, in call order (perceiver :360, :451, :470)
    hook = enc.global_maxp.register_forward_hook(lambda mod, inp, out: pools.append(inp[0].detach().flatten(2).argmax(-1).int()))
    for dt in (torch.float32, torch.float64):
        t0 = time.time()
        torch.nn.functional.conv3d = _chunked_conv3d_f64 if dt == torch.float64 else _ORIG_CONV3D
        enc = enc.to(dt)
        for p in enc.parameters():
            p.requires_grad_(True)
            p.grad = None
        outs = enc(ins.to(dt), rs['low_dim_state'].to(dt), rs['lang_goal_emb'].to(dt), rs['lang_token_embs'].to(dt), None, bounds, None)
        total, _ = oagent.losses(outs[0], outs[1], outs[2], rs['trans_action_indicies'], rs['rot_grip_action_indicies'],
                                 rs['ignore_collisions'], outs[3] if arm else None, rs.get('label'))
        total.backward()
        res[dt] = dict(outs=[o.detach().double() for o in outs[:4 if arm else 3]], loss=total.detach().double(),
                       grads={n: p.grad.detach().double().clone() for n, p in enc.named_parameters()})
        print('%s: reference %s forward + backward %.0fs' % (name, dt, time.time() - t0), flush=True)
        del outs, total
    torch.nn.functional.conv3d = _ORIG_CONV3D
    hook.remove()
    assert len(pools) == 6
    r32, r64 = res[torch.float32], res[torch.float64]
    arrs = dict(cfg_V=cfg['V'], cfg_k=cfg['k'], cfg_s=cfg['s'], cfg_depth=cfg['depth'], cfg_latents=cfg['latents'],
                cfg_low_dim=cfg['low_dim'], cfg_B=cfg['B'], cfg_H=cfg['H'], cfg_W=cfg['W'], cfg_ncam=len(cfg['cams']),
                cfg_cams=np.array(cfg['cams']), cfg_arm=int(arm), cfg_crop=int(crop), cfg_seed=seed, nproj=nproj)
    flat64, flat32 = r64['outs'][0].reshape(cfg['B'], -1), r32['outs'][0].reshape(cfg['B'], -1)
    sidx = ow.hashed_int('digest', (4096,), 0, flat64.shape[1])
    top = flat64.topk(16, dim=1)
    arrs.update(q_trans_argmax=flat64.argmax(1), q_trans_top_vals=top.values, q_trans_top_idx=top.indices, q_trans_sample_idx=sidx,
                q_trans_sample=flat64[:, sidx], q_trans_lse=torch.logsumexp(flat64, 1), rot_grip=r64['outs'][1], collision=r64['outs'][2],
                q_spread32=torch.stack([(a - b).abs().max() for a, b in zip(r32['outs'], r64['outs'])]),
                loss=r64['loss'], loss32=r32['loss'])
    if arm:
        arrs['arm_out'] = r64['outs'][3]
    # the loss is only piecewise smooth: a global max pool whose two largest voxels are closer than the arithmetic's rounding hands its
    # whole gradient to another voxel.  The float64 run's choices are stored so that a test can evaluate the product's backward at the SAME
    # subgradient (and count how many of the product's own choices differ); reference fp32 vs float64 differ in `pool_flips32` of them
    for i in range(3):
        arrs['pool_argmax64_%d' % i] = pools[3 + i]
    arrs['pool_flips32'] = np.array([int((pools[i] != pools[3 + i]).sum()) for i in range(3)])
    print('%s: max-pool choices that differ between the reference in fp32 and in float64: %s' % (name, arrs['pool_flips32'].tolist()))
    names = list(r64['grads'])
    arrs['grad_names'] = np.array(names)
    arrs['grad_norm64'] = torch.stack([r64['grads'][n].norm() for n in names])
    arrs['grad_err32'] = torch.stack([(r32['grads'][n] - r64['grads'][n]).norm() for n in names])
    arrs['grad_max64'] = torch.stack([r64['grads'][n].abs().max() for n in names])
    arrs['grad_proj64'] = torch.stack([ow.project(r64['grads'][n], n, nproj) for n in names])
    # self-check of the estimator on the reference's own fp32 gradient: estimated vs exact ||g32 - g64||
    est = torch.stack([ow.projection_error(ow.project(r32['grads'][n], n, nproj), arrs['grad_proj64'][i])
                       for i, n in enumerate(names)])
    rel = (est / (arrs['grad_err32'] + 1e-300))
    print('%s: projection estimate / exact ||g32 - g64||: median %.2f, range %.2f .. %.2f' % (name, float(rel.median()), float(rel.min()), float(rel.max())))
    worst = sorted(((float(arrs['grad_err32'][i] / (arrs['grad_norm64'][i] + 1e-300)), n) for i, n in enumerate(names)), reverse=True)[:8]
    print('%s: largest reference fp32 noise ||g32 - g64|| / ||g64||: %s' % (name, ', '.join('%s %.1e' % (n, e) for e, n in worst)))
    print('%s: |q32 - q64| per output %s, loss32 - loss64 %.2e' % (name, [float(x) for x in arrs['q_spread32']], float(r32['loss'] - r64['loss'])))
    for n in names:
        if r64['grads'][n].numel() <= 20000 and not n.startswith(('pos_encoding', 'latents')):
            arrs['grad64__' + n] = r64['grads'][n]
            arrs['grad32__' + n] = r32['grads'][n].float()
    save(name, **arrs)


def kink_statistics(name, cfg, seeds, nproj=16):
    """Round 6 (round-5 review, item 7: "quantify the kink effect instead of only explaining it"): the reference encoder in fp32 AND in
    float64 on MANY seeded batches of one geometry, slim per batch -- per parameter tensor ||g64||, ||g32 - g64|| (exact: the reference's
    own un-forced fp32 error against float64) and `nproj` +-1 projections of g64, from which tools/experiments/kink_statistics_gpu.py
    estimates ||g - g64|| of the PRODUCT's un-forced gradient in each arithmetic.  Resumable: one temporary file per seed."""
    part_dir = os.path.join(HERE, '_' + name)
    os.makedirs(part_dir, exist_ok=True)
    for seed in seeds:
        part = os.path.join(part_dir, 's%d.npz' % seed)
        if os.path.exists(part):
            continue
        enc, sd = make_ref_encoder(cfg)
        rs = batch_for(cfg, seed=seed)
        pcd = [rs['%s_point_cloud' % c] for c in cfg['cams']]
        rgb = [rs['%s_rgb' % c] for c in cfg['cams']]
        bounds = torch.tensor([synthetic.SCENE_BOUNDS])
        coords, feats = ovox.flatten_cameras(pcd, rgb)
        grid = ref_voxelize(coords, feats, bounds, cfg['V'], cfg['B'])
        ins = grid.permute(0, 4, 1, 2, 3).detach()
        res = {}
        t0 = time.time()
        for dt in (torch.float32, torch.float64):
            torch.nn.functional.conv3d = _chunked_conv3d_f64 if dt == torch.float64 else _ORIG_CONV3D
            try:
                enc = enc.to(dt)
                for p in enc.parameters():
                    p.requires_grad_(True)
                    p.grad = None
                outs = enc(ins.to(dt), rs['low_dim_state'].to(dt), rs['lang_goal_emb'].to(dt), rs['lang_token_embs'].to(dt), None, bounds, None)
                total, _ = oagent.losses(outs[0], outs[1], outs[2], rs['trans_action_indicies'], rs['rot_grip_action_indicies'], rs['ignore_collisions'])
                total.backward()
            finally:
                torch.nn.functional.conv3d = _ORIG_CONV3D
            res[dt] = dict(loss=total.detach().double(), grads={n: p.grad.detach().double().clone() for n, p in enc.named_parameters()})
            del outs, total
        names = list(res[torch.float64]['grads'])
        g64, g32 = res[torch.float64]['grads'], res[torch.float32]['grads']
        np.savez_compressed(part, seed=seed, loss64=res[torch.float64]['loss'].numpy(), loss32=res[torch.float32]['loss'].numpy(),
                            grad_names=np.array(names),
                            grad_norm64=torch.stack([g64[n].norm() for n in names]).numpy(),
                            grad_err32=torch.stack([(g32[n] - g64[n]).norm() for n in names]).numpy(),
                            grad_proj64=torch.stack([ow.project(g64[n], n, nproj) for n in names]).numpy())
        rel = sorted(((float((g32[n] - g64[n]).norm() / (g64[n].norm() + 1e-300)), n) for n in names if float(g64[n].norm()) > 1e-6), reverse=True)
        print('%s seed %d: %.0fs; reference fp32 vs float64, worst tensors: %s' % (name, seed, time.time() - t0, ', '.join('%s %.1e' % (n, e) for e, n in rel[:3])), flush=True)
    parts = [np.load(os.path.join(part_dir, 's%d.npz' % sd_), allow_pickle=False) for sd_ in seeds]
    save(name, cfg_V=cfg['V'], cfg_k=cfg['k'], cfg_s=cfg['s'], cfg_depth=cfg['depth'], cfg_latents=cfg['latents'], cfg_low_dim=cfg['low_dim'],
         cfg_B=cfg['B'], cfg_H=cfg['H'], cfg_W=cfg['W'], cfg_ncam=len(cfg['cams']), nproj=nproj, seeds=np.array(list(seeds)),
         grad_names=parts[0]['grad_names'], loss64=np.stack([p['loss64'] for p in parts]), loss32=np.stack([p['loss32'] for p in parts]),
         grad_norm64=np.stack([p['grad_norm64'] for p in parts]), grad_err32=np.stack([p['grad_err32'] for p in parts]),
         grad_proj64=np.stack([p['grad_proj64'] for p in parts]))


def grad_noise_kinks(name, cfg, seed, arm=False, crop=False):
    """Supplement of grad_noise_fixture (same encoder, same batch): the FLOAT64 forward's pre-activations within KINK_TAU of zero at every
    grid-sized LeakyReLU (capture_kinks) -- the subgradient choices of the run whose gradients f5n_noise_* holds.  Forward only."""
    enc, sd = make_ref_encoder(cfg, arm)
    rs = batch_for(cfg, seed=seed, arm=arm, crop=crop)
    pcd = [rs['%s_point_cloud' % c] for c in cfg['cams']]
    rgb = [rs['%s_rgb' % c] for c in cfg['cams']]
    bounds = rs['target_object_scene_bounds'] if crop else torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    grid = ref_voxelize(coords, feats, bounds, cfg['V'], cfg['B'])
    ins = grid.permute(0, 4, 1, 2, 3).detach()
    dt = torch.float64
    torch.nn.functional.conv3d = _chunked_conv3d_f64
    store = {}
    t0 = time.time()
    try:
        enc = enc.to(dt)
        hooks = capture_kinks(enc, store)
        with torch.no_grad():
            outs = enc(ins.to(dt), rs['low_dim_state'].to(dt), rs['lang_goal_emb'].to(dt), rs['lang_token_embs'].to(dt), None, bounds, None)
        for h in hooks:
            h.remove()
    finally:
        torch.nn.functional.conv3d = _ORIG_CONV3D
    print('%s: float64 forward %.0fs; pre-activations within %.0e of zero: %s' % (
        name, time.time() - t0, KINK_TAU, {k.split('__')[1]: int(v.numel()) for k, v in store.items() if k.endswith('idx')}), flush=True)
    flat = outs[0].reshape(cfg['B'], -1)
    save(name, kink_tau=KINK_TAU, cfg_seed=seed, q_trans_lse=torch.logsumexp(flat, 1), **store)      # (lse: ties the file to its f5n fixture)


# ----------------------------------------------------------------------------- F11: the 2Robots (one_policy_more_heads) encoder
def encoder2_fixture(name, cfg, digest=False):
    """reference PerceiverVoxelLang2RobotsEncoder (perceiver_lang_io.py:488-860) forward + backward of the summed two-arm loss
    (agent :1283-1363) on hashed weights: six outputs, loss, every parameter gradient's norm (small ones in full)."""
    enc = ref_pl.PerceiverVoxelLang2RobotsEncoder(
        depth=cfg['depth'], iterations=1, voxel_size=cfg['V'], initial_dim=10, low_dim_size=cfg['low_dim'],
        num_latents=cfg['latents'], voxel_patch_size=cfg['k'], voxel_patch_stride=cfg['s'],
        activation='lrelu', input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0)
    shapes = {n: tuple(p.shape) for n, p in enc.named_parameters()}
    from voxactb_amd.agents.peract_bc.perceiver_lang_io import PerceiverVoxelLang2RobotsEncoder as Mine
    mine = Mine(depth=cfg['depth'], iterations=1, voxel_size=cfg['V'], initial_dim=10, low_dim_size=cfg['low_dim'],
                num_latents=cfg['latents'], voxel_patch_size=cfg['k'], voxel_patch_stride=cfg['s'], activation='lrelu',
                input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0)
    assert [(n, tuple(p.shape)) for n, p in mine.named_parameters()] == list(shapes.items())      # names, shapes AND order
    enc.load_state_dict(ow.hashed_state_dict(shapes, 0), strict=False)
    enc.eval()
    rs = batch_for(cfg, seed=1)
    B, V = cfg['B'], cfg['V']
    pcd = [rs['%s_point_cloud' % c] for c in cfg['cams']]
    rgb = [rs['%s_rgb' % c] for c in cfg['cams']]
    bounds = torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    grid = ref_voxelize(coords, feats, bounds, V, B)
    ins = grid.permute(0, 4, 1, 2, 3).detach()
    proprio_left = ow.hashed_uniform('f11.proprio_left', (B, cfg['low_dim']), 0.0, 1.0)
    trans_left = ow.hashed_int('f11.trans_left', (B, 3), 0, V)
    rot_left = torch.cat((ow.hashed_int('f11.rot_left', (B, 3), 0, 72), ow.hashed_int('f11.grip_left', (B, 1), 0, 2)), 1)
    dy_sums, hooks = {}, []
    if digest:       # conv-bias gradients sum dY over 10^6 voxels: keep the float64 sums next to the reference's fp32 ones (see F5g)
        def tap(nm):
            def fwd_hook(mod, inp, out):
                out.register_hook(lambda gr: dy_sums.__setitem__(nm, gr.double().sum(dim=(0, 2, 3, 4))))
            return fwd_hook
        for n, m_ in enc.named_modules():
            if isinstance(m_, torch.nn.Conv3d):
                hooks.append(m_.register_forward_hook(tap(n)))
    outs = enc(ins, rs['low_dim_state'], proprio_left, rs['lang_goal_emb'], rs['lang_token_embs'], None, bounds, None)
    tr, _ = oagent.losses(outs[0], outs[1], outs[2], rs['trans_action_indicies'], rs['rot_grip_action_indicies'], rs['ignore_collisions'])
    tl, _ = oagent.losses(outs[3], outs[4], outs[5], trans_left, rot_left, rs['ignore_collisions'])
    total = tr + tl              # mean over B of the sum of both arms' heads (agent :1365-1369)
    total.backward()
    arrs = dict(cfg_V=V, cfg_k=cfg['k'], cfg_s=cfg['s'], cfg_depth=cfg['depth'], cfg_latents=cfg['latents'],
                cfg_low_dim=cfg['low_dim'], cfg_B=B, cfg_H=cfg['H'], cfg_W=cfg['W'], cfg_ncam=len(cfg['cams']),
                proprio_left=proprio_left, trans_left=trans_left, rot_grip_left=rot_left,
                rot_grip_right=outs[1], collision_right=outs[2], rot_grip_left_out=outs[4], collision_left=outs[5],
                loss=total.detach())
    if digest:           # headline-size grid: keep digests of the two 10^6-logit maps instead of the tensors (as F5 does)
        sidx = ow.hashed_int('digest', (4096,), 0, V ** 3)
        for side, q in (('right', outs[0]), ('left', outs[3])):
            flat = q.detach().reshape(B, -1)
            top = flat.topk(16, dim=1)
            arrs.update({'q_trans_%s_argmax' % side: flat.argmax(1), 'q_trans_%s_top_vals' % side: top.values,
                         'q_trans_%s_top_idx' % side: top.indices, 'q_trans_%s_sample' % side: flat[:, sidx],
                         'q_trans_%s_lse' % side: torch.logsumexp(flat.double(), 1)})
        arrs['q_trans_sample_idx'] = sidx
    else:
        arrs.update(grid=grid, q_trans_right=outs[0].detach(), q_trans_left=outs[3].detach())
    arrs['grad_names'] = np.array([n for n, _ in enc.named_parameters()])
    arrs['grad_norms'] = torch.stack([p.grad.norm() for _, p in enc.named_parameters()])
    for n, p in enc.named_parameters():
        if p.numel() <= 20000 and not n.startswith(('pos_encoding', 'latents')):
            arrs['grad__' + n] = p.grad
    for n, v in dy_sums.items():
        arrs['dysum64__' + n + '.bias'] = v
    for h in hooks:
        h.remove()
    print('%s: loss %.6f' % (name, float(total)))
    save(name, **arrs)



# ----------------------------------------------------------------------------- F12: CLIP text encoder (act() path)
F12_SENTENCES = ['open the jar', 'open the drawer', 'put the cube in the drawer with the left arm',
                 "Don't spill: hold the jar's lid, then twist 2 times!"]


def f12_clip_text():
    """reference CLIP class (helpers/clip/core/clip.py:289-440) with name-hashed TEXT weights, fp32, on sentences tokenized by
    the reference tokenizer (simple_tokenizer.py + its BPE vocabulary): token ids, sentence features, token embeddings."""
    stub_modules()
    sys.modules['ftfy'].fix_text = lambda t: t            # ftfy is not installed here; identity on these ASCII sentences
    sys.path.insert(0, os.path.join(REF, 'peract'))
    # clip.py uses a relative import of its tokenizer: load it as the submodule of a stand-in package
    import types
    pkg = types.ModuleType('ref_clip_pkg')
    pkg.__path__ = [os.path.join(REF, 'peract', 'helpers', 'clip', 'core')]
    sys.modules['ref_clip_pkg'] = pkg
    clip = importlib.import_module('ref_clip_pkg.clip')
    model = clip.CLIP(1024, 224, (3, 4, 6, 3), 64, None, 77, 49408, 512, 8, 12).float().eval()
    sd = synthetic.hashed_clip_text_state_dict()
    missing = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if not k.startswith('visual.') and k != 'logit_scale'], missing.missing_keys
    assert not missing.unexpected_keys
    tokens = clip.tokenize(F12_SENTENCES)
    with torch.no_grad():
        feat, emb = model.encode_text_with_embeddings(tokens)
    save('f12_clip_text', tokens=tokens, feat=feat, emb=emb, sentences=np.array(F12_SENTENCES))
    print('f12: |feat| max %.3f, |emb| max %.3f, %d tokens in the longest sentence' % (
        float(feat.abs().max()), float(emb.abs().max()), int((tokens != 0).sum(1).max())))


# ----------------------------------------------------------------------------- F6 / F9: agent-level
def stub_modules():
    for name in ['torchvision', 'torchvision.transforms', 'pytorch3d', 'pytorch3d.transforms', 'pyrender',
                 'pyrender.trackball', 'trimesh', 'rlbench', 'rlbench.backend', 'rlbench.backend.const',
                 'rlbench.backend.observation_two_robots', 'rlbench.observation_config_two_robots',
                 'rlbench.utils', 'rlbench.demo', 'pyrep', 'pyrep.const', 'ftfy', 'wandb', 'matplotlib',
                 'matplotlib.pyplot', 'clip', 'cv2', 'PIL', 'PIL.Image', 'natsort']:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    sys.modules['rlbench.backend.const'].DEPTH_SCALE = 2 ** 24 - 1


def f6_update_traces(name='f6_update_traces', cfg=None, tags=(('a', False, False), ('b', True, True)), steps=3):
    stub_modules()
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    if not dist.is_initialized():
        dist.init_process_group('gloo', rank=0, world_size=1)
    ref_agent = sys.modules.get('ref_agent') or load('ref_agent', 'agents/peract_bc/qattention_peract_bc_agent.py')
    cfg = cfg or CFG_UPD
    out = {}
    for tag, arm, crop in tags:
        c = dict(cfg, low_dim=4 if tag == 'a' else 7)
        enc, sd = make_ref_encoder(c, arm)
        enc.train()
        agent = ref_agent.QAttentionPerActBCAgent(
            layer=0, coordinate_bounds=synthetic.SCENE_BOUNDS, perceiver_encoder=enc, camera_names=c['cams'],
            batch_size=c['B'], voxel_size=c['V'], bounds_offset=None, voxel_feature_size=3, image_crop_size=64,
            num_rotation_classes=72, rotation_resolution=5, lr=5e-4, include_low_dim_state=True,
            image_resolution=[c['H'], c['W']], lambda_weight_l2=1e-6, transform_augmentation=False,
            optimizer_type='lamb', crop_target_obj_voxel=crop, arm_pred_loss=arm)
        agent.build(training=True, device='cpu')
        losses = []
        batches = []
        for step in range(steps):
            rs = batch_for(c, seed=10 + step, arm=arm, crop=crop)
            batches.append(rs)
            r = agent.update(step, dict(rs))
            s = agent._summaries
            losses.append([float(r['total_loss']), float(s['losses/trans_loss']), float(s['losses/rot_loss']),
                           float(s['losses/grip_loss']), float(s['losses/collision_loss']),
                           float(s.get('losses/arm_loss', 0.0))])
        out[tag + '_losses'] = np.array(losses)
        names = [n.replace('_qnet.module.', '') for n, _ in agent._q.named_parameters()]
        out[tag + '_param_names'] = np.array(names)
        out[tag + '_param_sums'] = torch.stack([p.detach().double().sum() for _, p in agent._q.named_parameters()])
        out[tag + '_param_abs_sums'] = torch.stack([p.detach().double().abs().sum() for _, p in agent._q.named_parameters()])
        # oracle replay of the same three steps
        P = {k: v.clone() for k, v in sd.items()}
        ob = []
        for rs in batches:
            ob.append(dict(pcd=[rs['%s_point_cloud' % cam] for cam in c['cams']],
                           rgb=[rs['%s_rgb' % cam] for cam in c['cams']],
                           proprio=rs['low_dim_state'], lang_token_embs=rs['lang_token_embs'],
                           bounds=rs['target_object_scene_bounds'] if crop else torch.tensor([synthetic.SCENE_BOUNDS]),
                           trans=rs['trans_action_indicies'], rot_grip=rs['rot_grip_action_indicies'],
                           ignore_collisions=rs['ignore_collisions'], label=rs.get('label')))
        tr = oagent.train_steps(P, ob, c['V'], steps, **enc_kw(c, arm))
        ol = np.array([t['total'] for t in tr])
        print('update trace %s: reference %s oracle %s' % (tag, np.array(losses)[:, 0], ol))
        # Step 0 is a pure forward comparison.  Later steps go through LAMB, whose first-step update is
        # ~lr*trust*3.16*sign(g) per element (v starts at 0, lamb.py:99-107): elements whose true gradient is
        # zero (e.g. trans_decoder bias: sum(softmax - onehot) == 0) get a full-size step whose SIGN is fp32
        # rounding noise, so two correct fp32 implementations drift apart by ~1e-4..1e-3 in the loss.
        dl = np.abs(ol - np.array(losses)[:, 0])
        assert dl[0] < 5e-5 and dl.max() < 5e-3, dl
        out[tag + '_oracle_losses'] = ol
    out.update(cfg_V=cfg['V'], cfg_k=cfg['k'], cfg_s=cfg['s'], cfg_depth=cfg['depth'], cfg_latents=cfg['latents'],
               cfg_B=cfg['B'], cfg_H=cfg['H'], cfg_W=cfg['W'], cfg_ncam=len(cfg['cams']))
    save(name, **out)



# ----------------------------------------------------------------------------- F13: update() of the reference 2Robots agent
def two_arm_batch(cfg, seed):
    """the single-arm synthetic sample re-keyed for two arms (tests/test_agent2robots_gpu.py builds the same batch):
    right = its labels / proprio / pose, left = name-hashed labels and proprio."""
    rs = batch_for(cfg, seed=seed)
    B, V = cfg['B'], cfg['V']
    out = {k: v for k, v in rs.items() if k.endswith(('_rgb', '_point_cloud')) or k in ('lang_goal_emb', 'lang_token_embs', 'ignore_collisions')}
    out['low_dim_state_right_arm'] = rs['low_dim_state']
    out['low_dim_state_left_arm'] = ow.hashed_uniform('f13.proprio_left', (B, cfg['low_dim']), 0.0, 1.0, seed)
    out['trans_action_indicies_right'] = rs['trans_action_indicies']
    out['rot_grip_action_indicies_right'] = rs['rot_grip_action_indicies']
    out['gripper_pose_right'] = rs['gripper_pose']
    out['trans_action_indicies_left'] = ow.hashed_int('f13.trans_left', (B, 3), 0, V, seed).float()
    out['rot_grip_action_indicies_left'] = torch.cat((ow.hashed_int('f13.rot_left', (B, 3), 0, 72, seed),
                                                      ow.hashed_int('f13.grip_left', (B, 1), 0, 2, seed)), 1).float()
    out['gripper_pose_left'] = rs['gripper_pose'].clone()
    return out


def f13_update_traces_2robots():
    """three update() steps (LAMB, no augmentation, no dropout) of the REFERENCE QAttentionPerActBCAgent2Robots
    (qattention_peract_bc_agent.py:966-1455) on name-hashed weights: losses per step and the parameters afterwards."""
    stub_modules()
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    if not dist.is_initialized():
        dist.init_process_group('gloo', rank=0, world_size=1)
    ref_agent = sys.modules.get('ref_agent') or load('ref_agent', 'agents/peract_bc/qattention_peract_bc_agent.py')
    # QFunction2Robots wraps with DDP(device_ids=[device]) unconditionally (agent :897-898; the single-arm QFunction has a CPU
    # branch, :50-54): hand it a DDP that drops the argument on the CPU -- the reference source is untouched
    real_ddp = ref_agent.DDP
    ref_agent.DDP = lambda module, device_ids=None: real_ddp(module)
    c = dict(CFG_UPD, low_dim=4)
    enc = ref_pl.PerceiverVoxelLang2RobotsEncoder(
        depth=c['depth'], iterations=1, voxel_size=c['V'], initial_dim=10, low_dim_size=c['low_dim'], num_latents=c['latents'],
        voxel_patch_size=c['k'], voxel_patch_stride=c['s'], activation='lrelu', input_dropout=0.0, attn_dropout=0.0,
        decoder_dropout=0.0)
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    enc.train()
    agent = ref_agent.QAttentionPerActBCAgent2Robots(
        layer=0, coordinate_bounds=synthetic.SCENE_BOUNDS, perceiver_encoder=enc, camera_names=c['cams'], batch_size=c['B'],
        voxel_size=c['V'], bounds_offset=None, voxel_feature_size=3, image_crop_size=64, num_rotation_classes=72,
        rotation_resolution=5, lr=5e-4, include_low_dim_state=True, image_resolution=[c['H'], c['W']], lambda_weight_l2=1e-6,
        transform_augmentation=False, optimizer_type='lamb')
    agent.build(training=True, device='cpu')
    losses = []
    for step in range(3):
        r = agent.update(step, two_arm_batch(c, 10 + step))
        sm = agent._summaries
        losses.append([float(r['total_loss']), float(sm['losses/trans_loss']), float(sm['losses/rot_loss']),
                       float(sm['losses/grip_loss']), float(sm['losses/collision_loss'])])
    ref_agent.DDP = real_ddp
    print('2Robots update trace: %s' % np.array(losses)[:, 0])
    names = [n.replace('_qnet.module.', '') for n, _ in agent._q.named_parameters()]
    save('f13_update_traces_2robots', losses=np.array(losses), param_names=np.array(names),
         param_abs_sums=torch.stack([p.detach().double().abs().sum() for _, p in agent._q.named_parameters()]),
         cfg_V=c['V'], cfg_k=c['k'], cfg_s=c['s'], cfg_depth=c['depth'], cfg_latents=c['latents'], cfg_B=c['B'], cfg_H=c['H'],
         cfg_W=c['W'], cfg_low_dim=c['low_dim'])


# ----------------------------------------------------------------------------- F7
def f7_lamb():
    out = {}
    tensors = {'w_rand': ow.hashed_normal('lamb_w', (37, 5)), 'w_zero': torch.zeros(11), 'w_big': 100 * ow.hashed_normal('lamb_b', (64,))}
    for name, w0 in tensors.items():
        p = torch.nn.Parameter(w0.clone())
        opt = ref_lamb.Lamb([p], lr=5e-4, weight_decay=1e-6, betas=(0.9, 0.999), adam=False)
        ws = []
        wo, m, v = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0)
        for step in range(3):
            g = ow.hashed_normal('lamb_g_%s_%d' % (name, step), w0.shape)
            if name == 'w_zero' and step == 0:
                g = torch.zeros_like(g)
            p.grad = g.clone()
            opt.step()
            ws.append(p.detach().clone())
            wo, m, v, _ = oagent.lamb_step(wo, g, m, v)
            assert torch.equal(wo, p.detach()), (name, step, float((wo - p.detach()).abs().max()))
            out['%s_g%d' % (name, step)] = g
        out[name + '_w0'] = w0
        out[name + '_w'] = torch.stack(ws)
    save('f7_lamb', **out)


# ----------------------------------------------------------------------------- F8: SE(3) augmentation, run by the REFERENCE
def _load_ref_augmentation():
    """path-load the reference's voxel/augmentation.py.  Its `pytorch3d.transforms` (pytorch3d==0.3.0, not vendored, not
    installed) is a module holding the three restated helpers of oracle/se3.py -- the only part that is not reference code --
    and `helpers.utils.rand_dist / rand_discrete` (utils.py:501-508, torch.rand / torch.randint) are scripted so that the
    draws of every attempt are explicit inputs of the fixture."""
    import types
    stub_modules()
    t3d = types.ModuleType('pytorch3d.transforms')
    t3d.quaternion_to_matrix = ose3.quaternion_to_matrix
    t3d.euler_angles_to_matrix = ose3.euler_angles_to_matrix
    t3d.matrix_to_quaternion = ose3.matrix_to_quaternion
    sys.modules['pytorch3d'].transforms = t3d
    sys.modules['pytorch3d.transforms'] = t3d
    ref_aug = load('ref_augmentation', 'voxel/augmentation.py')
    return ref_aug, sys.modules['helpers.utils']


class _ScriptedDraws:
    """stands in for utils.rand_dist / utils.rand_discrete: attempt a returns unit[a] and, for the a-th roll / pitch / yaw
    call, steps[a][:, i]; counts the attempts the reference consumed."""

    def __init__(self, unit, steps):
        self.unit, self.steps, self.n_dist, self.n_disc = unit, steps, 0, 0

    def rand_dist(self, size, min=-1.0, max=1.0):
        u = self.unit[self.n_dist]
        self.n_dist += 1
        assert tuple(u.shape) == tuple(size)
        return u.clone()

    def rand_discrete(self, size, min=0, max=1):
        a, i = divmod(self.n_disc, 3)
        self.n_disc += 1
        v = self.steps[a][:, i:i + 1].clone()
        assert int(v.min()) >= min and int(v.max()) <= max, (min, max, v)
        return v


def f8_se3():
    ref_aug, ref_utils = _load_ref_augmentation()
    V = 100
    out = {}
    real = ref_utils.rand_dist, ref_utils.rand_discrete
    cases = [('a', 8, False, 0, False), ('b', 6, True, 1, False), ('c', 5, True, 0, False), ('r', 4, False, 0, True)]
    try:
        for tag, B, per_sample, layer, retry in cases:
            rs = synthetic.make_replay_sample(B, ['front', 'wrist'], (16, 16), V, 4, seed=5 + len(out), crop_target_obj_voxel=per_sample,
                                              crop_radius=0.45)
            pose = rs['gripper_pose'][:, 0].clone()
            rg = rs['rot_grip_action_indicies'][:, 0]
            tr = rs['trans_action_indicies'][:, 0]
            pcd = [rs['front_point_cloud'][:, 0], rs['wrist_point_cloud'][:, 0]]
            bounds = rs['target_object_scene_bounds'][:, 0] if per_sample else torch.tensor([synthetic.SCENE_BOUNDS])
            A = 3
            unit = ow.hashed_uniform('se3_shift_' + tag, (A, B, 3), -1, 1)
            steps = torch.cat([torch.zeros(A, B, 2, dtype=torch.int64), ow.hashed_int('se3_yaw_' + tag, (A, B, 1), -9, 10)], 2)
            if retry:
                # sample 2 sits in a corner of the scene: attempt 0 and 1 push it out of the lower bound (index < 0), so the
                # reference re-draws the WHOLE batch (augmentation.py:116); attempt 2 keeps everyone inside
                pose[2, :3] = torch.tensor(synthetic.SCENE_BOUNDS[:3]) + 0.01
                unit[0, 2], unit[1, 2], unit[2] = -1.0, -0.9, 0.05
            draws = _ScriptedDraws(unit, steps)
            ref_utils.rand_dist, ref_utils.rand_discrete = draws.rand_dist, draws.rand_discrete
            ti, ri, pp = ref_aug.apply_se3_augmentation(
                [p.clone() for p in pcd], pose, tr, rg, bounds, layer, torch.from_numpy(np.array([0.125] * 3)), [0.0, 0.0, 45.0], 5, V, 5, 'cpu')
            used = draws.n_dist
            assert (used == 3) if retry else (used == 1), used
            # the oracle's one-attempt restatement on the accepted draw
            oti, ori, opp, ok = ose3.augment(pcd, pose, rg, bounds, unit[used - 1], steps[used - 1], [0.125] * 3, 5, V, 5, layer=layer)
            assert ok and torch.equal(oti.long(), ti.long()) and torch.equal(ori.long(), ri.long())
            assert max(float((a - b).abs().max()) for a, b in zip(opp, pp)) == 0.0
            for a in range(used - 1):
                assert not ose3.augment(pcd, pose, rg, bounds, unit[a], steps[a], [0.125] * 3, 5, V, 5, layer=layer)[3]
            out.update({tag + '_pose': pose, tag + '_rot_grip': rg, tag + '_pcd0': pcd[0], tag + '_pcd1': pcd[1], tag + '_bounds': bounds,
                        tag + '_shift_unit': unit, tag + '_rpy_steps': steps, tag + '_layer': layer, tag + '_attempts': used,
                        tag + '_trans_idx': ti, tag + '_rot_grip_idx': ri, tag + '_pcd0_out': pp[0], tag + '_pcd1_out': pp[1]})
        # two arms under one perturbation (augmentation.py:187-348): right = the synthetic sample, left = a second one;
        # attempt 0 pushes only the LEFT arm of sample 1 out -> both arms are re-drawn
        B, A = 6, 3
        rs_r = synthetic.make_replay_sample(B, ['front'], (16, 16), V, 4, seed=70)
        rs_l = synthetic.make_replay_sample(B, ['front'], (16, 16), V, 4, seed=170)
        pose_r, rg_r, tr_r = rs_r['gripper_pose'][:, 0].clone(), rs_r['rot_grip_action_indicies'][:, 0], rs_r['trans_action_indicies'][:, 0]
        pose_l, rg_l, tr_l = rs_l['gripper_pose'][:, 0].clone(), rs_l['rot_grip_action_indicies'][:, 0], rs_l['trans_action_indicies'][:, 0]
        pcd = [rs_r['front_point_cloud'][:, 0]]
        bounds = torch.tensor([synthetic.SCENE_BOUNDS])
        unit = ow.hashed_uniform('se3_shift_2', (A, B, 3), -1, 1)
        steps = torch.cat([torch.zeros(A, B, 2, dtype=torch.int64), ow.hashed_int('se3_yaw_2', (A, B, 1), -9, 10)], 2)
        pose_l[1, :3] = torch.tensor(synthetic.SCENE_BOUNDS[:3]) + 0.01
        unit[0, 1], unit[1] = -1.0, 0.1
        draws = _ScriptedDraws(unit, steps)
        ref_utils.rand_dist, ref_utils.rand_discrete = draws.rand_dist, draws.rand_discrete
        tir, rir, til, ril, pp = ref_aug.apply_se3_augmentation_2Robots(
            [p.clone() for p in pcd], pose_r, tr_r, rg_r, pose_l, tr_l, rg_l, bounds, 0, torch.from_numpy(np.array([0.125] * 3)),
            [0.0, 0.0, 45.0], 5, V, 5, 'cpu')
        used = draws.n_dist
        assert used == 2, used
        o_r = ose3.augment(pcd, pose_r, rg_r, bounds, unit[1], steps[1], [0.125] * 3, 5, V, 5)
        o_l = ose3.augment(pcd, pose_l, rg_l, bounds, unit[1], steps[1], [0.125] * 3, 5, V, 5)
        assert o_r[3] and o_l[3] and torch.equal(o_r[0].long(), tir.long()) and torch.equal(o_r[1].long(), rir.long())
        assert torch.equal(o_l[0].long(), til.long()) and torch.equal(o_l[1].long(), ril.long())
        assert float((o_r[2][0] - pp[0]).abs().max()) == 0.0          # the clouds turn about the RIGHT arm (:343-346)
        assert ose3.augment(pcd, pose_r, rg_r, bounds, unit[0], steps[0], [0.125] * 3, 5, V, 5)[3]          # right alone would pass
        assert not ose3.augment(pcd, pose_l, rg_l, bounds, unit[0], steps[0], [0.125] * 3, 5, V, 5)[3]
        out.update(t_pose_right=pose_r, t_rot_grip_right=rg_r, t_pose_left=pose_l, t_rot_grip_left=rg_l, t_pcd0=pcd[0], t_bounds=bounds,
                   t_shift_unit=unit, t_rpy_steps=steps, t_attempts=used, t_trans_idx_right=tir, t_rot_grip_idx_right=rir,
                   t_trans_idx_left=til, t_rot_grip_idx_left=ril, t_pcd0_out=pp[0])
    finally:
        ref_utils.rand_dist, ref_utils.rand_discrete = real
    # invariants of the three restated pytorch3d helpers (the one piece the reference tree does not hold)
    q = torch.cat([pose_r[:, 6:7], pose_r[:, 3:6]], 1)
    R = ose3.quaternion_to_matrix(q)
    assert float((R @ R.transpose(1, 2) - torch.eye(3)).abs().max()) < 1e-5
    from scipy.spatial.transform import Rotation
    assert np.abs(R.numpy() - Rotation.from_quat(pose_r[:, 3:].numpy()).as_matrix()).max() < 1e-5
    q2 = ose3.matrix_to_quaternion(R)
    assert float(torch.minimum((q2 - q).abs().amax(1), (q2 + q).abs().amax(1)).max()) < 1e-5
    out['cases'] = np.array([c[0] for c in cases])
    save('f8_se3', **out)


# ----------------------------------------------------------------------------- F9: act() through the whole agent stack
class _StubTextEncoder:
    """Stands in for CLIP RN50 (weights `data/clip_rn50.pth` are in neither tree): a deterministic
    function of the tokens with CLIP's output shapes (helpers/clip/core/clip.py:426-440)."""

    def __init__(self):
        self.table = ow.hashed_normal('clip_stub_table', (997, 512))
        self.proj = ow.hashed_normal('clip_stub_proj', (512, 1024)) * 0.05

    def float(self):
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def encode_text_with_embeddings(self, tokens):
        emb = self.table[tokens.long() % 997]                       # [1,77,512]
        return emb.mean(1) @ self.proj, emb


def _act_observation(cfg, seed):
    """the observation dict RolloutGenerator hands to PreprocessAgent.act (rollout_generator.py:146-152): every
    element [1, 1, ...]; rgb still 0..255."""
    rs = synthetic.make_replay_sample(1, cfg['cams'], (cfg['H'], cfg['W']), cfg['V'], cfg['low_dim'], seed=seed)
    obs = {}
    g = np.random.Generator(np.random.Philox(key=4242 + seed))
    for c in cfg['cams']:
        obs['%s_rgb' % c] = rs['%s_rgb' % c].clone()
        obs['%s_point_cloud' % c] = rs['%s_point_cloud' % c].clone()
        # a plausible pinhole camera looking at the scene centre (values only need to be shared by both sides)
        from scipy.spatial.transform import Rotation
        ext = np.eye(4)
        ext[:3, :3] = Rotation.from_euler('xyz', g.uniform(-0.6, 0.6, 3)).as_matrix()
        ext[:3, 3] = np.array([0.2, 0.0, 1.1]) + g.uniform(-1.5, 1.5, 3)
        f = -cfg['W'] / (2 * np.tan(np.deg2rad(30)))
        K = np.array([[f, 0, cfg['W'] / 2], [0, f, cfg['H'] / 2], [0, 0, 1.0]])
        obs['%s_camera_extrinsics' % c] = torch.from_numpy(ext).float().view(1, 1, 4, 4)
        obs['%s_camera_intrinsics' % c] = torch.from_numpy(K).float().view(1, 1, 3, 3)
    obs['low_dim_state'] = rs['low_dim_state'].clone()
    obs['lang_goal_tokens'] = torch.from_numpy(g.integers(0, 49408, (1, 1, 77))).long()
    return obs


def f9_act():
    stub_modules()
    import types
    ref_agent = load('ref_agent', 'agents/peract_bc/qattention_peract_bc_agent.py')
    ref_agent.load_clip = lambda *a, **k: (MagicMock(), None)
    ref_agent.build_model = lambda sd: _StubTextEncoder()
    # the stack agent does `from helpers import utils` / `from agents.peract_bc.qattention_peract_bc_agent import ...`;
    # importing the `agents.peract_bc` package would pull launch_utils -> rlbench, so the package
    # levels are empty namespaces and the two modules are the path-loaded ones
    for pkg in ('agents', 'agents.peract_bc'):
        if pkg not in sys.modules:
            sys.modules[pkg] = types.ModuleType(pkg)
            sys.modules[pkg].__path__ = []
    sys.modules['agents.peract_bc.qattention_peract_bc_agent'] = ref_agent
    ref_stack = load('ref_stack', 'agents/peract_bc/qattention_stack_agent.py')
    ref_prep = load('ref_prep', 'helpers/preprocess_agent.py')
    out = {}
    for tag, cfg in (('s', dict(CFG_UPD, low_dim=4, B=1)), ('c2', CFG_C2)):
        enc, sd = make_ref_encoder(cfg, False)
        qa = ref_agent.QAttentionPerActBCAgent(
            layer=0, coordinate_bounds=synthetic.SCENE_BOUNDS, perceiver_encoder=enc, camera_names=cfg['cams'],
            batch_size=1, voxel_size=cfg['V'], bounds_offset=None, voxel_feature_size=3, image_crop_size=64,
            num_rotation_classes=72, rotation_resolution=5, lr=5e-4, include_low_dim_state=True,
            image_resolution=[cfg['H'], cfg['W']], lambda_weight_l2=1e-6, transform_augmentation=False,
            optimizer_type='lamb')
        agent = ref_prep.PreprocessAgent(ref_stack.QAttentionStackAgent([qa], 5, cfg['cams']))
        agent.build(training=False, device=torch.device('cpu'))
        obs = _act_observation(cfg, seed=21)
        with torch.no_grad():
            emb, tok = _StubTextEncoder().encode_text_with_embeddings(obs['lang_goal_tokens'][0])
        t0 = time.time()
        with torch.no_grad():
            res = agent.act(0, {k: v.clone() for k, v in obs.items()}, deterministic=True)
        print('f9 %s: reference act() %.1fs' % (tag, time.time() - t0))
        qt = res.info['q_depth0'].reshape(1, -1)
        pre = 'f9%s_' % tag
        out.update({pre + 'cfg_' + k: cfg[k] for k in ('V', 'k', 's', 'depth', 'latents', 'low_dim', 'H', 'W')})
        out[pre + 'cfg_ncam'] = len(cfg['cams'])
        out[pre + 'lang_goal_tokens'] = obs['lang_goal_tokens']
        out[pre + 'lang_goal_emb'], out[pre + 'lang_token_embs'] = emb, tok
        for c in cfg['cams']:
            out[pre + c + '_ext'] = obs['%s_camera_extrinsics' % c]
            out[pre + c + '_int'] = obs['%s_camera_intrinsics' % c]
            out[pre + c + '_pixel_coord'] = np.array(res.observation_elements['%s_pixel_coord' % c], dtype=np.float64)
        out[pre + 'continuous_action'] = np.asarray(res.action, dtype=np.float64)
        out[pre + 'attention_coordinate'] = res.observation_elements['attention_coordinate_layer_0']
        out[pre + 'trans_action_indicies'] = res.observation_elements['trans_action_indicies']
        out[pre + 'rot_grip_action_indicies'] = res.observation_elements['rot_grip_action_indicies']
        out[pre + 'coords'] = res.info['voxel_idx_depth0']
        top = qt.topk(16, dim=1)
        sidx = ow.hashed_int('digest', (4096,), 0, qt.shape[1])
        out[pre + 'q_top_vals'], out[pre + 'q_top_idx'] = top.values, top.indices
        out[pre + 'q_sample_idx'], out[pre + 'q_sample'] = sidx, qt[:, sidx]
        out[pre + 'q_sum'] = qt.double().sum()
        # the softmaxed rotation / grip / collision heads are not returned by act(); recompute them the way act() does
        # (agent :394-416) from the reference Q-function on the same preprocessed observation
        o2 = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in obs.items()}
        ob = [[o2['%s_rgb' % c][0], o2['%s_point_cloud' % c][0]] for c in cfg['cams']]
        pc = [o2['%s_point_cloud' % c][0] for c in cfg['cams']]
        with torch.no_grad():
            q_t, q_rg, q_c, vox = qa._q(ob, o2['low_dim_state'][0], pc, emb, tok, qa._coordinate_bounds, None, None)
            out[pre + 'q_rot_grip_softmax'] = qa._softmax_q_rot_grip(q_rg)
            out[pre + 'q_collision_softmax'] = qa._softmax_ignore_collision(q_c)
            # oracle agreement (pins oracle.agent.softmax_heads / attention_coordinate / choose_highest_action)
            P = {k: v for k, v in sd.items()}
            oo = oagent.qfunction_forward(P, pc, [o[0] for o in ob], o2['low_dim_state'][0], tok,
                                          torch.tensor([synthetic.SCENE_BOUNDS]), cfg['V'], **enc_kw(cfg))
            sqt, sqr, sqc = oagent.softmax_heads(oo[0], oo[1], oo[2])
            co, rg, ic = oagent.choose_highest_action(sqt, sqr, sqc)
            assert torch.equal(co.int(), res.info['voxel_idx_depth0'].int()), (co, res.info['voxel_idx_depth0'])
            assert float((sqr - out[pre + 'q_rot_grip_softmax']).abs().max()) < 1e-5
            ac = oagent.attention_coordinate(torch.tensor([synthetic.SCENE_BOUNDS]), co, cfg['V'])
            assert float((ac[0] - torch.from_numpy(np.asarray(out[pre + 'attention_coordinate']))).abs().max()) < 1e-6
    save('f9_act', **out)



# ----------------------------------------------------------------------------- F14: act() of the reference 2Robots stack
def f14_act_2robots():
    """QAttentionStackAgent2Robots.act (qattention_stack_agent.py:127-246) over QAttentionPerActBCAgent2Robots.act (agent
    :1457-1582) with a stub text encoder, for which_arm = right and left.  (The reference PreprocessAgent passes the single-arm
    extras positionally, preprocess_agent.py:46, which the 2Robots stack agent does not accept: the stack agent is called with
    the observation PreprocessAgent would have produced.)"""
    stub_modules()
    import types
    ref_agent = sys.modules.get('ref_agent') or load('ref_agent', 'agents/peract_bc/qattention_peract_bc_agent.py')
    ref_agent.load_clip = lambda *a, **k: (MagicMock(), None)
    ref_agent.build_model = lambda sd: _StubTextEncoder()
    for pkg in ('agents', 'agents.peract_bc'):
        if pkg not in sys.modules:
            sys.modules[pkg] = types.ModuleType(pkg)
            sys.modules[pkg].__path__ = []
    sys.modules['agents.peract_bc.qattention_peract_bc_agent'] = ref_agent
    ref_stack = sys.modules.get('ref_stack') or load('ref_stack', 'agents/peract_bc/qattention_stack_agent.py')
    cfg = dict(CFG_UPD, low_dim=4, B=1)
    enc = ref_pl.PerceiverVoxelLang2RobotsEncoder(
        depth=cfg['depth'], iterations=1, voxel_size=cfg['V'], initial_dim=10, low_dim_size=cfg['low_dim'],
        num_latents=cfg['latents'], voxel_patch_size=cfg['k'], voxel_patch_stride=cfg['s'], activation='lrelu',
        input_dropout=0.0, attn_dropout=0.0, decoder_dropout=0.0)
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    qa = ref_agent.QAttentionPerActBCAgent2Robots(
        layer=0, coordinate_bounds=synthetic.SCENE_BOUNDS, perceiver_encoder=enc, camera_names=cfg['cams'], batch_size=1,
        voxel_size=cfg['V'], bounds_offset=None, voxel_feature_size=3, image_crop_size=64, num_rotation_classes=72,
        rotation_resolution=5, lr=5e-4, include_low_dim_state=True, image_resolution=[cfg['H'], cfg['W']],
        lambda_weight_l2=1e-6, transform_augmentation=False, optimizer_type='lamb')
    agent = ref_stack.QAttentionStackAgent2Robots([qa], 5, cfg['cams'])
    agent.build(training=False, device=torch.device('cpu'))
    # (the agent keeps its bounds as a list until update() / a caller turns them into a tensor; act() indexes a tensor, :1467)
    qa._coordinate_bounds = torch.tensor(synthetic.SCENE_BOUNDS).unsqueeze(0)
    obs = _act_observation(cfg, seed=23)
    obs['low_dim_state_right_arm'] = obs.pop('low_dim_state')
    obs['low_dim_state_left_arm'] = ow.hashed_uniform('f14.proprio_left', (1, 1, cfg['low_dim']), 0.0, 1.0)
    with torch.no_grad():
        emb, tok = _StubTextEncoder().encode_text_with_embeddings(obs['lang_goal_tokens'][0])
    out = {'cfg_' + k: cfg[k] for k in ('V', 'k', 's', 'depth', 'latents', 'low_dim', 'H', 'W', 'B')}
    out.update(lang_goal_tokens=obs['lang_goal_tokens'], lang_goal_emb=emb, lang_token_embs=tok,
               low_dim_state_left_arm=obs['low_dim_state_left_arm'])
    for c in cfg['cams']:
        out[c + '_ext'], out[c + '_int'] = obs['%s_camera_extrinsics' % c], obs['%s_camera_intrinsics' % c]
    prep = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in obs.items()}
    for arm in ('right', 'left'):
        with torch.no_grad():
            res = agent.act(0, {k: v.clone() for k, v in prep.items()}, True, arm)
        out[arm + '_continuous_action'] = np.asarray(res.action, dtype=np.float64)
        out[arm + '_attention_coordinate'] = res.observation_elements['attention_coordinate_layer_0']
        out[arm + '_trans_action_indicies'] = res.observation_elements['trans_action_indicies']
        out[arm + '_rot_grip_action_indicies'] = res.observation_elements['rot_grip_action_indicies']
        for c in cfg['cams']:
            out[arm + '_' + c + '_pixel_coord'] = np.array(res.observation_elements['%s_pixel_coord' % c], dtype=np.float64)
        qt = res.info['q_depth_%s0' % arm].reshape(1, -1)
        top = qt.topk(8, dim=1)
        out[arm + '_q_top_vals'], out[arm + '_q_top_idx'] = top.values, top.indices
        print('f14 %s: action %s' % (arm, np.round(out[arm + '_continuous_action'], 4)))
    save('f14_act_2robots', **out)


# ----------------------------------------------------------------------------- F10: RGB-D -> point cloud (PyRep)
def f10_depth():
    """PyRep's VisionSensor module cannot be imported (it loads the CoppeliaSim backend), but the point-cloud code is
    pure numpy: the three module-level helpers and the static method are compiled FROM the reference file where it lies
    (ast-selected definitions, executed in a namespace that only holds numpy) and run here."""
    import ast
    path = REF + '/PyRep/pyrep/objects/vision_sensor.py'
    tree = ast.parse(open(path).read())
    want = {'_create_uniform_pixel_coords_image', '_transform', '_pixel_to_world_coords'}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'VisionSensor'][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'pointcloud_from_depth_and_camera_params'][0]
    fn.decorator_list = []
    ns = {'np': np}
    exec(compile(ast.Module(body=body + [fn], type_ignores=[]), path, 'exec'), ns)
    ref_fn = ns['pointcloud_from_depth_and_camera_params']
    from scipy.spatial.transform import Rotation
    g = np.random.Generator(np.random.Philox(key=77))
    B, H, W, cams = 2, 24, 20, 2
    out = dict(cfg_B=B, cfg_H=H, cfg_W=W, cfg_ncam=cams)
    near, far = 0.01, 4.5
    for b in range(B):
        for c in range(cams):
            ext = np.eye(4)
            ext[:3, :3] = Rotation.from_euler('xyz', g.uniform(-2.5, 2.5, 3)).as_matrix()
            ext[:3, 3] = np.array([0.2, 0.0, 1.1]) + g.uniform(-1.0, 1.0, 3)
            f = -W / (2 * np.tan(np.deg2rad(g.uniform(25, 40))))
            K = np.array([[f, 0, W / 2], [0, f * g.uniform(0.9, 1.1), H / 2], [0, 0, 1.0]])
            d01 = g.uniform(0.05, 0.6, (H, W)).astype(np.float32)                      # a 0..1 depth buffer
            depth_m = (np.float32(near) + d01 * np.float32(far - near)).astype(np.float32)
            cloud = ref_fn(depth_m, ext, K).astype(np.float32)                         # stored as float32 observations
            mine, inv = ovox.depth_to_point_cloud(d01, ext, K, near, far)
            assert np.array_equal(mine, cloud), (b, c, np.abs(mine - cloud).max())
            tag = 'b%d_c%d_' % (b, c)
            out[tag + 'depth01'], out[tag + 'ext'], out[tag + 'int'], out[tag + 'cloud'] = d01, ext, K, cloud
    out['near'], out['far'] = near, far
    save('f10_depth_clouds', **out)


# ----------------------------------------------------------------------------- F15: launch_utils labels + replay fill (reference-run)
from tests.f15_common import *      # noqa: E402,F401,F403  (stub demo / stubs / case lists shared with tests/test_replay_cpu.py)
from tests import f15_common as f15c      # noqa: E402


def f15_launch_utils():
    """`_get_action` for every which_arm branch and `_add_keypoints_to_replay` on a stub demo, run by the REFERENCE's own
    launch_utils.py (:167-298, :301-486; path-loaded with the simulator / CLIP / hydra imports stubbed)."""
    import copy
    stub_modules()
    for n in ('omegaconf', 'hydra'):
        sys.modules.setdefault(n, MagicMock())
    ref_lu = sys.modules.get('ref_launch_utils') or load('ref_launch_utils', 'agents/peract_bc/launch_utils.py')
    ref_utils = sys.modules['helpers.utils']
    d = f15_stub_demo()
    out = {'demo_' + k: v for k, v in d.items()}
    for ci, (arm, label, dom) in enumerate(F15_ACTION_CASES):
        for di, (vs, off, crop_aug, seed) in enumerate(F15_DEPTH_CASES):
            res = []
            np.random.seed(seed)
            demo = f15_observations(d)
            for i in range(len(demo)):
                a = (demo[i], demo[max(0, i - 1)], list(F15_BOUNDS), vs, off, 5, crop_aug, arm, label)
                res.append(ref_lu._get_action(*a, dom))
            tag = 'act%d_%d' % (ci, di)
            names = (('trans', 'rot_grip', 'ignore', 'action', 'attention') if arm != 'both' else
                     ('trans_right', 'rot_grip_right', 'ignore', 'action_right', 'attention_right', 'trans_left', 'rot_grip_left',
                      'action_left', 'attention_left'))
            for j, nm in enumerate(names):
                out['%s_%s' % (tag, nm)] = np.array([np.asarray(r[j]) for r in res])
    real = ref_utils.extract_obs, ref_lu.tokenize
    ref_utils.extract_obs, ref_lu.tokenize = f15_extract_obs, f15_tokenize
    try:
        for tag, kw, labels, dom, bounds in F15_FILL_CASES:
            demo = f15_observations(d)
            rec = F15Recorder()
            try:
                ref_lu._add_keypoints_to_replay(f15_cfg(**kw), 'open_jar', 1, rec, demo[0], demo, F15_KEYPOINTS, F15_CAMS,
                                                copy.deepcopy(bounds), [F15_V], [0.15], 5, False, description=F15_DESCRIPTION,
                                                clip_model=F15Clip(), device='cpu', labels=labels, dominant_assistive_arm=dom)
            except TypeError as e:
                # upstream's which_arm == 'both' branch calls _get_action with 9 of its 10 positional arguments
                # (launch_utils.py:356-359 vs :167-177): the reference cannot fill a two-arm replay as published.  The build
                # gives `dominant_assistive_arm` a default; its two-arm fill is checked against the per-arm labels above.
                assert kw['which_arm'] == 'both' and 'dominant_assistive_arm' in str(e), e
                out['fill_both_reference_raises'] = np.array(str(e))
                continue
            f15_flatten_calls('fill_' + tag, rec.calls, out)
    finally:
        ref_utils.extract_obs, ref_lu.tokenize = real
    save('f15_launch_utils', **out)


# ----------------------------------------------------------------------------- F16: YARR's replay store (reference-run)
def f16_replay():
    """identical add / add_final sequences into the reference's TaskUniformReplayBuffer (YARR/yarr/replay_buffer/
    task_uniform_replay_buffer.py + uniform_replay_buffer.py:259-301, :639-756; pickle-per-transition on disk, as
    launch_utils.create_replay configures it): `sample_transition_batch(indices=...)` of every valid row, the rows it refuses,
    the per-task index lists and, for a two-rank world, the set of rows each rank's sampler ever draws."""
    import tempfile
    stub_modules()
    if not hasattr(np, 'bool'):
        np.bool = bool                       # uniform_replay_buffer.py:706 was written for numpy < 1.24 (environment shim)
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29536')
    if not dist.is_initialized():
        dist.init_process_group('gloo', rank=0, world_size=1)
    from yarr.replay_buffer.task_uniform_replay_buffer import TaskUniformReplayBuffer
    from yarr.replay_buffer.replay_buffer import ReplayElement
    from yarr.utils.observation_type import ObservationElement
    from tests import f16_common as fc
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        obs = [ObservationElement(n, sh, t) if is_obs else ReplayElement(n, sh, t) for n, sh, t, is_obs in fc.F16_OBS]
        extra = [ReplayElement(n, sh, t) for n, sh, t in fc.F16_EXTRA]
        buf = TaskUniformReplayBuffer(save_dir=os.path.join(tmp, 'replay'), batch_size=4, timesteps=1, replay_capacity=1000,
                                      action_shape=(8,), action_dtype=np.float32, reward_shape=(), reward_dtype=np.float32,
                                      update_horizon=1, observation_elements=obs, extra_replay_elements=extra)
        n = fc.f16_fill(buf)
        assert int(buf.add_count) == n
        valid, invalid = [], []
        for i in range(n):
            try:
                b = buf.sample_transition_batch(1, indices=[i])
                valid.append(i)
                for k, v in b.items():
                    a = np.asarray(v)[0]
                    out.setdefault('row__' + k, []).append(a.astype(str) if a.dtype == object else a)      # (no pickles in fixtures)
            except ValueError:
                invalid.append(i)
        for k in list(out):
            out[k] = np.stack(out[k])
        b = buf.sample_transition_batch(len(valid), indices=valid)
        out['batch_keys'] = np.array(sorted(b))
        out['batch_dtypes'] = np.array([str(np.asarray(b[k]).dtype) for k in sorted(b)])
        out['batch_shapes'] = np.array([str(tuple(np.asarray(b[k]).shape[1:])) for k in sorted(b)])
        out['valid'], out['invalid'] = np.array(valid), np.array(invalid)
        for t, rows in buf._task_idxs.items():
            out['task_rows__' + t] = np.array(rows)
        np.random.seed(0)
        counts = {}
        for _ in range(60):
            for i in buf.sample_transition_batch(32)['indices'][:, 0]:
                counts[int(i)] = counts.get(int(i), 0) + 1
        out['world1_drawn'] = np.array(sorted(counts))
        for r in range(2):                  # DDP stride (task_uniform_replay_buffer.py:103-108): what rank r of 2 can ever draw
            buf._num_replicas, buf._rank = 2, r
            np.random.seed(r)
            seen = set()
            for _ in range(60):
                seen |= set(int(i) for i in buf.sample_transition_batch(32)['indices'][:, 0])
            out['world2_rank%d_drawn' % r] = np.array(sorted(seen))
        buf._num_replicas, buf._rank = 1, 0
        buf.shutdown()
    save('f16_replay', **out)


SECTIONS = {
    'f1': f1_voxel_kats,
    'f3tiny': lambda: encoder_fixture('f3_encoder_tiny', CFG_TINY, arm=True),
    'f3c1': lambda: encoder_fixture('f3_encoder_c1', CFG_C1),
    # encoder switches reachable from the configs (PERACT_BC.yaml: transformer_iterations, no_language): tiny and configs[0] size, fwd + bwd
    'f3v_it2': lambda: encoder_fixture('f3v_encoder_tiny_iterations2', dict(CFG_TINY, variant=dict(iterations=2)), arm=True, digest=True, check_oracle=False, kinks=True),
    'f3v_it3c1': lambda: encoder_fixture('f3v_encoder_c1_iterations3', dict(CFG_C1, latents=48, depth=2, variant=dict(iterations=3)), digest=True, check_oracle=False, kinks=True),
    'f3v_noskip': lambda: encoder_fixture('f3v_encoder_c1_no_skip_connection', dict(CFG_C1, variant=dict(no_skip_connection=True)), digest=True, check_oracle=False, kinks=True),
    'f3v_noperc': lambda: encoder_fixture('f3v_encoder_c1_no_perceiver', dict(CFG_C1, variant=dict(no_perceiver=True)), digest=True, check_oracle=False, kinks=True),
    'f3v_posgrid': lambda: encoder_fixture('f3v_encoder_c1_pos_encoding_grid_only', dict(CFG_C1, variant=dict(pos_encoding_with_lang=False)), digest=True, check_oracle=False, kinks=True),
    'f3v_concat': lambda: encoder_fixture('f3v_encoder_c1_lang_concat', dict(CFG_C1, variant=dict(lang_fusion_type='concat', pos_encoding_with_lang=False)), digest=True, check_oracle=False, kinks=True),
    'f3v_tie': lambda: encoder_fixture('f3v_encoder_c1_weight_tie_layers', dict(CFG_C1, depth=3, variant=dict(weight_tie_layers=True)), digest=True, check_oracle=False, kinks=True),
    'f3v_nolang': lambda: encoder_fixture('f3v_encoder_c1_no_language', dict(CFG_C1, variant=dict(no_language=True)), digest=True, check_oracle=False, kinks=True),
    'f5': lambda: encoder_fixture('f5_encoder_c2_digest', CFG_C2, with_grads=False, digest=True),
    # (kinks=True, round 5: the fp32 run's LeakyReLU pre-activations within 3e-5 of zero, so that a test can evaluate the product's backward
    # at the reference's subgradient choices -- capture_kinks)
    'f5g': lambda: encoder_fixture('f5g_encoder_c2_grads', CFG_C2, with_grads=True, digest=True, f64_grads=True, kinks=True),
    'f5c3': lambda: encoder_fixture('f5c3_encoder_c3_digest', CFG_C3, arm=True, with_grads=True, digest=True, crop=True, f64_grads=True, kinks=True),
    'f5v200': lambda: encoder_fixture('f5v200_encoder_c5_digest', CFG_C5, with_grads=False, digest=True),
    'f5v50a': lambda: encoder_fixture('f5v50a_encoder_release_digest', CFG_V50, arm=True, with_grads=True, digest=True, crop=True, f64_grads=True, kinks=True),
    'f5v50b': lambda: encoder_fixture('f5v50b_encoder_release_digest', CFG_V50B, arm=True, with_grads=True, digest=True, crop=True, f64_grads=True, kinks=True),
    # the same grid with the reference's loss and backward (committed since round 4: ~15 minutes on the 8-core build container with the
    # stride-1 convs evaluated in slabs, _f5v200g above; as one ATen op per layer the backward had not finished in 75 minutes)
    'f5v200g': lambda: _f5v200g(),
    # configs[1] geometry at B = 8: 16 384 rows per linear layer = the row count from which the product dispatches the 128 x 512-tile
    # ("wide") GEMM / fp16 weight-gradient / fp16x2 data-gradient kernels and the fused GEGLU epilogue -- the headline's own dispatch
    # (B = 16 takes the same kernels) pinned on the reference's forward + backward (round 5; ~25 GB resident, a few minutes)
    'f5gb8': lambda: encoder_fixture('f5gb8_encoder_c2_b8_grads', dict(CFG_C2, B=8), with_grads=True, digest=True, check_oracle=False, f64_grads=True, kinks=True),
    # the reference in fp32 AND float64 on three batches per headline shape (the fp32-vs-fp64 spread is the yardstick of the gradient gates)
    **{'f5n_c2_s%d' % sd: (lambda sd=sd: grad_noise_fixture('f5n_noise_c2_s%d' % sd, CFG_C2, sd)) for sd in (1, 2, 3)},
    **{'f5n_c3_s%d' % sd: (lambda sd=sd: grad_noise_fixture('f5n_noise_c3_s%d' % sd, CFG_C3, sd, arm=True, crop=True)) for sd in (1, 2, 3)},
    'f5n_tiny': lambda: grad_noise_fixture('f5n_noise_tiny_s1', CFG_TINY, 1, arm=True),
    # the float64 run's LeakyReLU choices near zero for the same batches (round 5: the root cause of the 'forward-sensitive' batches)
    **{'f5k_c2_s%d' % sd: (lambda sd=sd: grad_noise_kinks('f5n_kinks_c2_s%d' % sd, CFG_C2, sd)) for sd in (1, 2, 3)},
    **{'f5k_c3_s%d' % sd: (lambda sd=sd: grad_noise_kinks('f5n_kinks_c3_s%d' % sd, CFG_C3, sd, arm=True, crop=True)) for sd in (1, 2, 3)},
    'f5k_v50a': lambda: grad_noise_kinks('f5n_kinks_v50a_s1', CFG_V50, 1, arm=True, crop=True),
    'f5k_v50b': lambda: grad_noise_kinks('f5n_kinks_v50b_s1', CFG_V50B, 1, arm=True, crop=True),
    'f5n_v50a': lambda: grad_noise_fixture('f5n_noise_v50a_s1', CFG_V50, 1, arm=True, crop=True),
    'f5n_v50b': lambda: grad_noise_fixture('f5n_noise_v50b_s1', CFG_V50B, 1, arm=True, crop=True),
    # the fp32 reference run's max-pool choices for the gradient digests (round 6, forward only)
    'f5p_f5g': lambda: pool_choices('f5g_encoder_c2_grads', CFG_C2),
    'f5p_f5c3': lambda: pool_choices('f5c3_encoder_c3_digest', CFG_C3, arm=True, crop=True),
    'f5p_f5gb8': lambda: pool_choices('f5gb8_encoder_c2_b8_grads', dict(CFG_C2, B=8)),
    'f5p_f5v50a': lambda: pool_choices('f5v50a_encoder_release_digest', CFG_V50, arm=True, crop=True),
    'f5p_f5v50b': lambda: pool_choices('f5v50b_encoder_release_digest', CFG_V50B, arm=True, crop=True),
    'f5p_f5v200g': lambda: pool_choices('f5v200g_encoder_c5_grads', CFG_C5, slabs=True),
    'f5p_f11c2': lambda: pool_choices_2robots('f11c2_encoder_2robots_c2_digest', CFG_C2),
    # 32 seeded configs[1] batches, reference fp32 and float64, slim (round 6; ~5 minutes of the build container per batch)
    'f5s_c2': lambda: kink_statistics('f5s_kink_statistics_c2', CFG_C2, range(1, 33)),
    'f11tiny': lambda: encoder2_fixture('f11_encoder_2robots_tiny', CFG_TINY),
    'f11c1': lambda: encoder2_fixture('f11_encoder_2robots_c1', CFG_C1),
    'f11c2': lambda: encoder2_fixture('f11c2_encoder_2robots_c2_digest', CFG_C2, digest=True),
    'f12': f12_clip_text,
    'f13': f13_update_traces_2robots,
    'f14': f14_act_2robots,
    'f6': f6_update_traces,
    # three LAMB update() steps of the REAL reference agent at BASELINE.json configs[1] geometry (V=100, depth 6, 2048 latents, B=1)
    'f6c2': lambda: f6_update_traces('f6c2_update_traces_c2', CFG_C2, tags=(('a', False, False),)),
    'f9': f9_act,
    'f10': f10_depth,
    'f7': f7_lamb,
    'f8': f8_se3,
    'f15': f15_launch_utils,
    'f16': f16_replay,
}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    ap.add_argument('--skip-c2', action='store_true')
    a = ap.parse_args()
    todo = [s for s in a.only.split(',') if s] or list(SECTIONS)
    torch.manual_seed(0)
    for s in todo:
        if (s in ('f5', 'f5g', 'f5gb8', 'f5c3', 'f5v200', 'f5v200g') or s.startswith(('f5n_c', 'f5k_c'))) and a.skip_c2:
            continue
        print('==', s)
        SECTIONS[s]()
