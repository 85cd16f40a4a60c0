"""Experiment: can the evaluation forward (voxelize + encoder) of act() be captured in a HIP graph, and what does a replay cost?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voxactb_amd import synthetic
from voxactb_amd.agents.peract_bc import launch_utils as lu

V, HW = 100, 128
cfg = lu.default_cfg(method__voxel_sizes=[V], method__voxel_patch_size=5, method__voxel_patch_stride=5, method__transformer_depth=6,
                     method__num_latents=2048, replay__batch_size=1, rlbench__camera_resolution=[HW, HW], ddp__num_devices=1)
torch.manual_seed(1)
ev = lu.create_agent(cfg)
ev.build(training=False, device=0)
qa = ev._pose_agent._qattention_agents[0]
dev = 'cuda:0'
rs = synthetic.make_replay_sample(1, cfg.rlbench.cameras, (HW, HW), V, 4, seed=3)
rs = {k: (v[:, 0] if v.dim() > 2 else v) for k, v in rs.items()}
rs = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()).to(dev) for k, v in rs.items()}
cams = cfg.rlbench.cameras
obs = [[rs['%s_rgb' % c], rs['%s_point_cloud' % c]] for c in cams]
pcd = [rs['%s_point_cloud' % c] for c in cams]
bounds = qa._coordinate_bounds
args = (obs, rs['low_dim_state'], pcd, rs['lang_goal_emb'], rs['lang_token_embs'], bounds, None, None)


def run():
    return qa._q(*args)


for _ in range(3):
    out = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = run()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 20 * 1e3
ref = [o.clone() for o in out[:3]]
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        run()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    gout = run()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 20 * 1e3
same = all(torch.equal(a, b) for a, b in zip(ref, gout[:3]))
print('forward of act(): eager %.2f ms, graph replay %.2f ms, outputs identical: %s' % (eager, graph, same))
