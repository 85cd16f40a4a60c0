"""GPU: DeviceBatchStream -- batches sampled ahead on a thread, staged in pinned memory, copied on a side stream -- hands the
training loop device tensors with exactly the contents the store would return on the host, and drives `update()`."""
import numpy as np
import pytest
import torch

from voxactb_amd import replay as R, synthetic
from voxactb_amd.agents.peract_bc import launch_utils as lu

pytestmark = pytest.mark.gpu


def test_stream_feeds_update_with_store_contents():
    cams, V, HW, Bt = ['front', 'wrist'], 16, 16, 3
    cfg = lu.default_cfg(method__voxel_sizes=[V], method__voxel_patch_size=3, method__voxel_patch_stride=4, method__transformer_depth=1,
                         method__num_latents=16, rlbench__cameras=cams, rlbench__camera_resolution=[HW, HW], replay__batch_size=Bt)
    buf = lu.create_replay(Bt, 1, False, True, None, cams, [V], [HW, HW])
    src = synthetic.make_replay_sample(12, cams, (HW, HW), V, 4, seed=2)
    names = [e.name for e in buf._observation_elements]
    for i in range(12):
        row = {}
        for n in names:
            if n == 'task':
                row[n] = 'open_jar' if i % 2 else 'open_drawer'
            elif n == 'lang_goal':
                row[n] = np.array(['open it'], dtype=object)
            else:
                row[n] = src[n][i, 0].numpy()
        if i % 4 == 3:
            buf.add_final(**row)
        else:
            buf.add(np.zeros(8, np.float32), 0.0, i % 4 == 2, False, demo=True, **row)
    buf.seed(0)
    stream = R.DeviceBatchStream(buf, device=0, depth=2)
    agent = lu.create_agent(cfg)
    agent.build(training=True, device=0)
    losses = []
    for step, batch in zip(range(6), stream):
        assert all(v.is_cuda for v in batch.values()) and 'lang_goal' not in batch and 'task' not in batch
        idx = batch['indices'][:, 0].cpu().numpy()
        want = buf.sample_transition_batch(Bt, indices=idx.tolist())
        for k in ('front_rgb', 'wrist_point_cloud_tp1', 'trans_action_indicies', 'gripper_pose', 'terminal'):
            assert np.array_equal(batch[k].cpu().numpy(), want[k]), k
        losses.append(float(agent.update(step, batch)['total_losses']))
    stream.close()
    assert all(np.isfinite(losses))
