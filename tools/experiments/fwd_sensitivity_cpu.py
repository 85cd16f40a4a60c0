#!/usr/bin/env python
"""Is the 3-8 % gradient spread of the 'forward-sensitive' batches (tests/test_grad_noise_gpu.py: c2_s3, v50b_s1) a property of the
REFERENCE on those batches, and where does it enter?   (build container only: imports the reference through tests/golden/make_golden.py)

The reference encoder is run in fp32 on one seeded batch; then again with a perturbation of `eps` (absolute, uniform in +-eps -- the size
of the default precision's forward difference, 1.4e-5) added to the OUTPUT of one module at a time (a forward hook; the module code is
untouched), and every parameter gradient is compared with the unperturbed run's: relative L2 change per tensor, grouped by block.

    python tools/experiments/fwd_sensitivity_cpu.py --cfg c2 --seed 3 [--sites d0,z,u0,u] [--eps 1.4e-5]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import make_golden as mg
from oracle import agent as oagent, voxel_grid as ovox
from voxactb_amd import synthetic

SITES = {
    'd0': 'input_preprocess', 'patch': 'patchify', 'x_cross': 'cross_attend_blocks.1', 'x_l2': 'layers.2.1', 'x_l5': 'layers.5.1',
    'z': 'decoder_cross_attn', 'z1': 'up0.conv_up.0', 'u0': 'up0', 'u': 'final',
    # the PRE-activation of a LeakyReLU block (the conv's output, before the block's activation sees it): a perturbation here can move
    # an element across zero, i.e. change LeakyReLU' from 0.02 to 1 -- the mechanism tools/experiments/fwd_sensitivity_gpu.py found
    'u0pre': 'up0.conv_up.2.conv3d', 'upre': 'final.conv3d', 'z1pre': 'up0.conv_up.0.conv3d', 'd0pre': 'input_preprocess.conv3d',
}
# perturb what ONE consumer of z / u reads (perceiver :451, :454, :470): SpatialSoftmax3D, the global max pool (2nd / 3rd call), up0
PRE_SITES = {'ss1_in': ('ss1', 1), 'maxp1_in': ('global_maxp', 2), 'up0_in': ('up0', 1), 'ss2_in': ('ss_final', 1), 'maxp2_in': ('global_maxp', 3),
             'transdec_in': ('trans_decoder', 1), 'ss0_in': ('ss0', 1), 'maxp0_in': ('global_maxp', 1), 'patchify_in': ('patchify', 1)}
GROUPS = [('heads', ('dense', 'rot_grip', 'arm_ff', 'trans_decoder')), ('final', ('final.',)), ('up0.2', ('up0.conv_up.2',)),
          ('up0.0', ('up0.conv_up.0',)), ('dec_xattn', ('decoder_cross_attn',)), ('layers', ('layers.',)),
          ('cross', ('cross_attend_blocks', 'latents')), ('ctx', ('pos_encoding', 'lang_preprocess', 'proprio_preprocess', 'patchify')),
          ('input', ('input_preprocess',))]


def run(enc, ins, rs, bounds, arm, site=None, eps=0.0, seed=0, dtype=torch.float32, mode='uniform'):
    mods = dict(enc.named_modules())
    hook = None
    if site is not None:
        gen = torch.Generator().manual_seed(1234 + seed)

        def noisy(out):
            if mode == 'uniform':
                return out + eps * (2 * torch.rand(out.shape, generator=gen, dtype=out.dtype) - 1)
            return out * (1 + eps * (2 * torch.rand(out.shape, generator=gen, dtype=out.dtype) - 1))       # relative

        def fwd_hook(mod, inp, out):
            return noisy(out)
        if site in PRE_SITES:
            # the INPUT of one consumer only (a forward pre-hook; `nth`: which call of a module used several times)
            name, nth = PRE_SITES[site]
            calls = [0]

            def pre_hook(mod, inp):
                calls[0] += 1
                if calls[0] == nth:
                    return (noisy(inp[0]),) + tuple(inp[1:])
                return None
            hook = mods[name].register_forward_pre_hook(pre_hook)
        else:
            hook = mods[SITES[site]].register_forward_hook(fwd_hook)
    for p in enc.parameters():
        p.requires_grad_(True)
        p.grad = None
    outs = enc(ins.to(dtype), rs['low_dim_state'].to(dtype), rs['lang_goal_emb'].to(dtype), rs['lang_token_embs'].to(dtype), None, bounds, None)
    total, _ = oagent.losses(outs[0], outs[1], outs[2], rs['trans_action_indicies'], rs['rot_grip_action_indicies'],
                             rs['ignore_collisions'], outs[3] if arm else None, rs.get('label'))
    total.backward()
    if hook is not None:
        hook.remove()
    return float(total), {n: p.grad.detach().clone() for n, p in enc.named_parameters()}, [o.detach() for o in outs]


def summarise(tag, g0, g1):
    gmax = max(float(v.norm()) for v in g0.values())
    rows = []
    for n in g0:
        nr = float(g0[n].norm())
        if nr < 1e-6 * gmax:
            continue
        rows.append((float((g1[n] - g0[n]).norm()) / nr, n))
    out = []
    for gname, pre in GROUPS:
        es = [e for e, n in rows if n.startswith(pre)]
        if es:
            out.append('%s %.1e/%.1e' % (gname, float(np.median(es)), max(es)))
    rows.sort(reverse=True)
    print('%-10s median/worst rel. L2 change by block:  %s' % (tag, '  '.join(out)), flush=True)
    print('           worst: %s' % ', '.join('%s %.2e' % (n, e) for e, n in rows[:4]), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='c2')
    ap.add_argument('--seed', type=int, default=3)
    ap.add_argument('--sites', default='d0,patch,x_cross,x_l5,z,z1,u0,u')
    ap.add_argument('--eps', type=float, default=1.4e-5)
    ap.add_argument('--mode', default='uniform')
    ap.add_argument('--f64', action='store_true')
    ap.add_argument('--draws', type=int, default=1, help='independent noise draws per site')
    a = ap.parse_args()
    torch.set_num_threads(8)
    cfg, arm, crop = {'c2': (mg.CFG_C2, False, False), 'c3': (mg.CFG_C3, True, True), 'v50a': (mg.CFG_V50, True, True),
                      'v50b': (mg.CFG_V50B, True, True), 'c1': (mg.CFG_C1, False, False)}[a.cfg]
    enc, sd = mg.make_ref_encoder(cfg, arm)
    rs = mg.batch_for(cfg, seed=a.seed, arm=arm, crop=crop)
    pcd = [rs['%s_point_cloud' % c] for c in cfg['cams']]
    rgb = [rs['%s_rgb' % c] for c in cfg['cams']]
    bounds = rs['target_object_scene_bounds'] if crop else torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    grid = mg.ref_voxelize(coords, feats, bounds, cfg['V'], cfg['B'])
    ins = grid.permute(0, 4, 1, 2, 3).detach()
    dt = torch.float64 if a.f64 else torch.float32
    if a.f64:
        torch.nn.functional.conv3d = mg._chunked_conv3d_f64
        enc = enc.to(dt)
    t0 = time.time()
    l0, g0, o0 = run(enc, ins, rs, bounds, arm, dtype=dt)
    print('%s seed %d: baseline loss %.6f (%.0f s)' % (a.cfg, a.seed, l0, time.time() - t0), flush=True)
    for site in [s for s in a.sites.split(',') if s]:
        for d in range(a.draws):
            l1, g1, o1 = run(enc, ins, rs, bounds, arm, site=site, eps=a.eps, seed=a.seed + 1000 * d, dtype=dt, mode=a.mode)
            dq = max(float((x - y).abs().max()) for x, y in zip(o0, o1))
            print('site %-8s draw %d loss %.6f  max |dQ| %.2e' % (site, d, l1, dq), flush=True)
            summarise(site, g0, g1)
