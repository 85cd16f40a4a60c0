"""Stub demo, stub observation extractor / tokenizer / text encoder and the case lists of fixture F15, shared by the generator
(tests/golden/make_golden.py, which runs the REFERENCE's launch_utils on them) and the test (tests/test_replay_cpu.py, which
runs voxactb_amd's launch_utils on them).  No reference code, no reference imports."""
import numpy as np
import torch

from voxactb_amd import synthetic

F15_CAMS, F15_HW, F15_V = ['front', 'wrist'], 4, 16
F15_BOUNDS = [float(v) for v in synthetic.SCENE_BOUNDS]


def f15_stub_demo(n=12, seed=0):
    """arrays of a stub two-arm demo: poses inside the scene, unit quaternions of both signs, open / closed grippers"""
    g = np.random.Generator(np.random.Philox(key=1500 + seed))
    lo, hi = np.array(F15_BOUNDS[:3]), np.array(F15_BOUNDS[3:])
    d = {}
    for side in ('right', 'left'):
        q = g.standard_normal((n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        d[side + '_pose'] = np.concatenate([lo + g.uniform(0.05, 0.95, (n, 3)) * (hi - lo), q], 1)
        d[side + '_open'] = (g.uniform(0, 1, n) > 0.4).astype(np.float64)
    d['ignore_collisions'] = (g.uniform(0, 1, n) > 0.5).astype(np.float64)
    d['target_object_pos'] = lo + g.uniform(0.3, 0.7, (n, 3)) * (hi - lo)
    d['auto_crop_radius'] = np.where(np.arange(n) % 4 == 0, 0.0, 0.25 + 0.01 * np.arange(n))
    return d


def f15_observations(d):
    from types import SimpleNamespace
    return [SimpleNamespace(gripper_right_pose=d['right_pose'][i].copy(), gripper_left_pose=d['left_pose'][i].copy(),
                            gripper_right_open=float(d['right_open'][i]), gripper_left_open=float(d['left_open'][i]),
                            ignore_collisions=float(d['ignore_collisions'][i]), target_object_pos=d['target_object_pos'][i].copy(),
                            auto_crop_radius=float(d['auto_crop_radius'][i]), index=i) for i in range(len(d['right_open']))]


def f15_extract_obs(obs, t, cameras, episode_length, which_arm, keypoint_label=None):
    """stands in for helpers/utils.py:516-635 (needs RLBench observations) on BOTH sides; what matters for the fill is
    which observation, which t / arm / label it is called with"""
    lab = -7.0 if keypoint_label is None else float(keypoint_label)
    if which_arm == 'both':
        d = {'low_dim_state_right_arm': np.array([obs.index, t, 1.0, lab], np.float32),
             'low_dim_state_left_arm': np.array([obs.index, t, 2.0, lab], np.float32)}
    else:
        d = {'low_dim_state': np.array([obs.index, t, {'right': 1.0, 'left': 2.0}.get(which_arm, 3.0), lab], np.float32)}
    d['ignore_collisions'] = np.array([obs.ignore_collisions], dtype=np.float32)
    d['wrist_world_to_cam'] = np.eye(4, dtype=np.float32)
    for c in cameras:
        d['%s_rgb' % c] = np.full((3, F15_HW, F15_HW), obs.index, np.float32)
        d['%s_point_cloud' % c] = np.full((3, F15_HW, F15_HW), 0.5 + t, np.float32)
    return d


def f15_tokenize(texts):
    return torch.tensor([[sum(ord(ch) for ch in texts[0]) % 4999] * 77], dtype=torch.int64)


class F15Clip:
    def encode_text_with_embeddings(self, tokens):
        v = tokens.float().mean() / 4999.0
        return v * torch.ones(1, 8), v * torch.ones(1, 3, 4)


class F15Recorder:
    def __init__(self):
        self.calls = []

    def add(self, action, reward, terminal, timeout, **kw):
        self.calls.append(dict(kw, __kind='add', __action=np.asarray(action), __reward=float(reward), __terminal=bool(terminal),
                               __timeout=bool(timeout)))

    def add_final(self, **kw):
        self.calls.append(dict(kw, __kind='add_final'))


def f15_cfg(which_arm, crop, crop_radius, arm_pred_loss, arm_id_to_proprio=False, arm_pred_input=False):
    from types import SimpleNamespace
    return SimpleNamespace(method=SimpleNamespace(which_arm=which_arm, crop_target_obj_voxel=crop, crop_radius=crop_radius,
                                                  arm_pred_loss=arm_pred_loss, arm_id_to_proprio=arm_id_to_proprio,
                                                  arm_pred_input=arm_pred_input),
                           rlbench=SimpleNamespace(episode_length=10))


F15_ACTION_CASES = [  # (which_arm, keypoint_label, dominant_assistive_arm)
    ('right', -1, ''), ('left', -1, ''), ('multiarm', 0, ''), ('multiarm', 1, ''), ('dominant', -1, 'right'), ('dominant', -1, 'left'),
    ('assistive', -1, 'right'), ('assistive', -1, 'left'), ('both', 0, '')]
F15_DEPTH_CASES = [([F15_V], [0.15], False, 0), ([F15_V, F15_V], [0.15], False, 0), ([F15_V, F15_V], [0.15], True, 123)]
F15_DESCRIPTION = 'hold the jar with the left hand and unscrew the lid with the right hand'
F15_FILL_CASES = [  # (tag, cfg kwargs, labels, dominant_assistive_arm, scene bounds)
    ('dom', dict(which_arm='dominant', crop=True, crop_radius=0.3, arm_pred_loss=True), [1, 0, 1], 'left', F15_BOUNDS),
    ('auto', dict(which_arm='assistive', crop=True, crop_radius='auto', arm_pred_loss=False, arm_id_to_proprio=True), [0, 0, 1], 'right', F15_BOUNDS),
    ('right', dict(which_arm='right', crop=False, crop_radius=0.0, arm_pred_loss=False), None, '', [F15_BOUNDS, [b + 0.1 for b in F15_BOUNDS]]),
    ('both', dict(which_arm='both', crop=False, crop_radius=0.0, arm_pred_loss=False), [0, 1, 0], '', F15_BOUNDS),
    ('multi', dict(which_arm='multiarm', crop=False, crop_radius=0.0, arm_pred_loss=False, arm_pred_input=True), [0, 1, 1], '', F15_BOUNDS)]
F15_KEYPOINTS = [3, 7, 11]


def f15_flatten_calls(tag, calls, out):
    out[tag + '_ncalls'] = len(calls)
    for i, c in enumerate(calls):
        out['%s_%d_keys' % (tag, i)] = np.array(sorted(c))
        for k, v in c.items():
            if k == 'lang_goal':
                v = np.array([str(x) for x in v])
            out['%s_%d__%s' % (tag, i, k)] = np.asarray(v)
