"""The replay schema and the add / add_final sequence of fixture F16, shared by the generator (tests/golden/make_golden.py feeds
it to the REFERENCE's TaskUniformReplayBuffer) and tests/test_replay_cpu.py (which feeds it to voxactb_amd.replay).  No
reference code, no reference imports."""
import numpy as np

# (name, shape, dtype, is ObservationElement) in the order launch_utils.create_replay lists them (launch_utils.py:56-145): camera
# and proprio observations are ObservationElements, labels / pose / language / task are plain ReplayElements inside the same list
F16_OBS = [('low_dim_state', (4,), np.float32, True), ('front_rgb', (3, 4, 4), np.float32, True),
           ('front_point_cloud', (3, 4, 4), np.float32, True), ('front_camera_extrinsics', (4, 4), np.float32, True),
           ('trans_action_indicies', (3,), np.int32, False), ('rot_grip_action_indicies', (4,), np.int32, False),
           ('ignore_collisions', (1,), np.int32, False), ('gripper_pose', (7,), np.float32, False),
           ('lang_goal_emb', (8,), np.float32, False), ('lang_token_embs', (3, 4), np.float32, False),
           ('task', (), str, False), ('lang_goal', (1,), object, False)]
F16_EXTRA = [('demo', (), bool)]
F16_TASKS = ('open_jar', 'open_drawer', 'hand_over_item')
F16_EPISODES = (3, 2, 4)         # episodes per task, interleaved below so that a task's rows are not contiguous
F16_STEPS = (4, 3, 5)


def _obs(val, task):
    g = np.random.Generator(np.random.Philox(key=1600 + int(val)))
    return dict(low_dim_state=np.full(4, val, np.float32), front_rgb=g.uniform(0, 255, (3, 4, 4)).astype(np.float32),
                front_point_cloud=g.uniform(-1, 1, (3, 4, 4)).astype(np.float32), front_camera_extrinsics=np.eye(4, dtype=np.float32) * val,
                trans_action_indicies=np.array([val, val + 1, val + 2], np.int32), rot_grip_action_indicies=np.array([val % 72, 1, 2, val % 2], np.int32),
                ignore_collisions=np.array([val % 2], np.int32), gripper_pose=g.uniform(-1, 1, 7).astype(np.float32),
                lang_goal_emb=np.full(8, 0.5 * val, np.float32), lang_token_embs=np.full((3, 4), 0.25 * val, np.float32), task=task,
                lang_goal=np.array(['do %s' % task], dtype=object))


def f16_fill(buf):
    """round-robin over the tasks, one episode at a time: `steps` transitions then the episode's final observation"""
    n = 0
    for ep in range(max(F16_EPISODES)):
        for ti, task in enumerate(F16_TASKS):
            if ep >= F16_EPISODES[ti]:
                continue
            steps = F16_STEPS[ti]
            for k in range(steps):
                val = 100 * ti + 10 * ep + k
                buf.add(np.full(8, val, np.float32), 100.0 if k == steps - 1 else 0.0, k == steps - 1, False, demo=True, **_obs(val, task))
                n += 1
            buf.add_final(**_obs(100 * ti + 10 * ep + steps, task))
            n += 1
    return n
