"""Host-side (Python) profile of update(): cProfile over 5 steps at config 2 -- shows that the interpreter spends
~16 ms per step (the GPU ~205 ms), i.e. the step is device-bound; run on a GPU box: python tools/profile_host.py"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import synthetic
from voxactb_amd.agents.peract_bc import launch_utils as lu
cfg = lu.default_cfg(method__voxel_sizes=[100], method__voxel_patch_size=5, method__voxel_patch_stride=5, replay__batch_size=16, rlbench__camera_resolution=[128,128])
agent = lu.create_agent(cfg); agent.build(training=True, device=0)
dev = torch.device('cuda', 0)
batch = {k: v.to(dev) for k, v in synthetic.make_replay_sample(16, cfg.rlbench.cameras, (128,128), 100, 4, seed=1).items()}
for i in range(3): agent.update(i, batch)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(5): float(agent.update(3+i, batch)['total_losses'])
torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/5
pr.disable()
print('step %.1f ms' % (dt*1e3))
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(22)
