"""Which ATen kernels does one update() launch, from where?  torch.profiler over 2 steps at configs[1] with stacks:
python tools/profile_aten.py  (GPU box).  Prints the device kernels that are not this library's, grouped by op and input shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import synthetic
from voxactb_amd.agents.peract_bc import launch_utils as lu
from torch.profiler import profile, ProfilerActivity
REL = '--release' in sys.argv          # the released recipe: V = 50, front | wrist | wrist2, replay batch 1, dominant arm
if REL:
    cfg = lu.default_cfg(method__which_arm='dominant', method__arm_pred_loss=True, method__crop_target_obj_voxel=True, method__voxel_sizes=[50],
                         method__voxel_patch_size=5, method__voxel_patch_stride=5, replay__batch_size=1, rlbench__camera_resolution=[128, 128],
                         rlbench__cameras=['front', 'wrist', 'wrist2'])
    cfg.method.transform_augmentation.aug_rpy = [0.0, 0.0, 45.0]
else:
    cfg = lu.default_cfg(method__voxel_sizes=[100], method__voxel_patch_size=5, method__voxel_patch_stride=5, replay__batch_size=16, rlbench__camera_resolution=[128, 128])
agent = lu.create_agent(cfg); agent.build(training=True, device=0)
dev = torch.device('cuda', 0)
if REL:
    batch = {k: v.to(dev) for k, v in synthetic.make_replay_sample(1, cfg.rlbench.cameras, (128, 128), 50, 7, seed=1, arm_pred_loss=True, crop_target_obj_voxel=True,
                                                                    crop_radius=0.3, keyframes_near_target=True).items()}
else:
    batch = {k: v.to(dev) for k, v in synthetic.make_replay_sample(16, cfg.rlbench.cameras, (128, 128), 100, 4, seed=1).items()}
for i in range(3):
    agent.update(i, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for i in range(2):
        agent.update(3 + i, batch)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='cuda_time_total', row_limit=30, max_name_column_width=50, max_shapes_column_width=60))
agg = {}
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::add_', 'aten::mul_', 'aten::cat', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::mul', 'aten::index', 'aten::gather',
                  'aten::sum', 'aten::div', 'aten::sub', 'aten::stack', 'aten::repeat', 'aten::where', 'aten::_to_copy', 'aten::index_select', 'aten::flip', 'aten::clone', 'aten::mean',
                  'aten::pow', 'aten::sqrt', 'aten::clamp', 'aten::exp', 'aten::neg', 'aten::abs', 'aten::max', 'aten::min', 'aten::argmax', 'aten::floor', 'aten::round', 'aten::remainder', 'aten::bmm', 'aten::mm', 'aten::matmul') and e.device_time_total > (2 if REL else 8):
        src = [f for f in (e.stack or []) if 'voxactb_amd' in f or 'bench.py' in f]
        key = (e.name, str(e.input_shapes)[:70], src[0].strip()[-90:] if src else '?')
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e.device_time_total
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%7.1f us  x%3d  %-12s %-70s %s' % (v[1] / 2, v[0] // 2 if v[0] > 1 else v[0], k[0], k[1], k[2]))
