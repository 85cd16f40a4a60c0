"""Why is the training path's voxelizer call 150 - 157 us inside bench.py's step and 122 us in tools/bench_voxel.py?  (GPU, round 6.)

The chain is six dependent launches.  Timed with one HIP event pair around the call in four situations:
  (a) back to back (the queue always holds the next call: what bench_voxel.py measures)
  (b) after a device synchronisation (empty queue: every launch of the chain waits for the host to enqueue it)
  (c) after a synchronisation, behind ~0.4 ms of unrelated device work that touches little memory (the host runs ahead; caches stay warm)
  (d) after a synchronisation, behind a 4 GB fill (the host runs ahead; L2 / MALL hold none of the voxelizer's workspace)
"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from voxactb_amd import synthetic
from voxactb_amd.voxel.voxel_grid import VoxelGrid

dev = 'cuda:0'
B, V, hw = 16, 100, 128
rs = synthetic.make_replay_sample(B, synthetic.CAMERAS4, (hw, hw), V, 4, seed=0)
pcd = [rs['%s_point_cloud' % c][:, 0].to(dev) for c in synthetic.CAMERAS4]
rgb = [((rs['%s_rgb' % c][:, 0] / 255.0) * 2.0 - 1.0).to(dev) for c in synthetic.CAMERAS4]
vg = VoxelGrid(synthetic.SCENE_BOUNDS, V, dev, B, 3, 4 * hw * hw, persistent=2)
small = torch.zeros(1 << 16, device=dev)
big = torch.empty(1 << 30, dtype=torch.float32, device=dev)
for _ in range(5):
    vg.voxelize_cameras(pcd, rgb)
torch.cuda.synchronize()


def timed(pre, n=30):
    tot = 0.0
    for _ in range(n):
        if pre != 'a':
            torch.cuda.synchronize()
        if pre == 'c':
            for _ in range(40):
                small.add_(1.0)
        if pre == 'd':
            big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vg.voxelize_cameras(pcd, rgb)
        e1.record()
        if pre != 'a':
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
    if pre == 'a':
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            vg.voxelize_cameras(pcd, rgb)
        e1.record()
        torch.cuda.synchronize()
        tot = e0.elapsed_time(e1)
    return tot / n * 1e3


out = {k: timed(k) for k in 'abcd'}
print(json.dumps({'us_per_call': out, 'what': {'a': 'back to back', 'b': 'empty queue', 'c': 'behind 40 small launches', 'd': 'behind a 4 GB fill'}}))
