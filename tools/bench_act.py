"""act() latency breakdown at config 2 geometry (B = 1): wall time per call, device time per call, top kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voxactb_amd import _lib, synthetic
from voxactb_amd.agents.peract_bc import launch_utils as lu

V, HW = 100, 128
cfg = lu.default_cfg(method__voxel_sizes=[V], method__voxel_patch_size=5, method__voxel_patch_stride=5,
                     method__transformer_depth=6, method__num_latents=2048, replay__batch_size=1,
                     rlbench__camera_resolution=[HW, HW], ddp__num_devices=1)
torch.manual_seed(1)
ev = lu.create_agent(cfg)
ev.build(training=False, device=0)
dev = 'cuda:0'
rs = synthetic.make_replay_sample(1, cfg.rlbench.cameras, (HW, HW), V, 4, seed=3)
obs = {k: v.to(dev) for k, v in rs.items()
       if k.endswith(('_rgb', '_point_cloud', '_camera_extrinsics', '_camera_intrinsics')) or k == 'low_dim_state'}
for cam in cfg.rlbench.cameras:
    obs['%s_camera_extrinsics' % cam][0, 0, 2, 3] = -1.0             # (a camera 1 m off the scene origin, as in bench.py)
obs['lang_goal_emb'] = rs['lang_goal_emb'][0].to(dev)
obs['lang_token_embs'] = rs['lang_token_embs'][0].to(dev)
for i in range(3):
    ev.act(i, dict(obs), deterministic=True)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    ev.act(i, dict(obs), deterministic=True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
timer = _lib.KernelTimer()
_lib.TIMER = timer
for i in range(5):
    ev.act(i, dict(obs), deterministic=True)
_lib.TIMER = None
agg = timer.summary()
tot = sum(d['ms'] for d in agg.values()) / 5
calls = sum(d['calls'] for d in agg.values()) / 5
print('act(): %.2f ms wall per call; %.2f ms in %d own kernels (event-timed, includes launch gaps inside a label)' % (wall, tot, calls))
for label, d in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])[:24]:
    print('  %-46s calls %4d  %7.3f ms' % (label, d['calls'] // 5, d['ms'] / 5))
