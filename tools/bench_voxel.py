#!/usr/bin/env python
"""Micro-benchmark of the voxelizer alone at BASELINE config 2 (for rocprofv3 runs)."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voxactb_amd import synthetic
from voxactb_amd.voxel.voxel_grid import VoxelGrid

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=16); ap.add_argument('--V', type=int, default=100)
ap.add_argument('--hw', type=int, default=128); ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--persistent', type=int, default=0)
ap.add_argument('--chain', type=int, default=0, help='vxb_voxelize_select_chain: 0 default, 3 = the separate-launch chain of rounds 2-4')
a = ap.parse_args()
dev = 'cuda:0'
if a.chain:
    from voxactb_amd import _lib
    assert _lib.lib().vxb_voxelize_select_chain(a.chain) == 0
rs = synthetic.make_replay_sample(a.B, synthetic.CAMERAS4, (a.hw, a.hw), a.V, 4, seed=0)
pcd = [rs['%s_point_cloud' % c][:, 0].to(dev) for c in synthetic.CAMERAS4]
rgb = [((rs['%s_rgb' % c][:, 0] / 255.0) * 2.0 - 1.0).to(dev) for c in synthetic.CAMERAS4]
vg = VoxelGrid(synthetic.SCENE_BOUNDS, a.V, dev, a.B, 3, 4 * a.hw * a.hw, persistent=a.persistent)
for _ in range(5):
    out = vg.voxelize_cameras(pcd, rgb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    out = vg.voxelize_cameras(pcd, rgb)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
alg = a.B * (4 * a.hw * a.hw * 6 * 4 + a.V ** 3 * 10 * 4)
print(json.dumps({'persistent': a.persistent, 'chain': a.chain, 'voxelize_ms': ms, 'algorithmic_bytes': alg, 'GBps': alg / ms / 1e6, 'frac_of_8TBps': alg / ms / 1e6 / 8000,
                  'occupied': int((out[..., -1] > 0).sum())}))
