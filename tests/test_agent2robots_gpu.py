"""GPU: the `one_policy_more_heads` agent stack -- create_agent(variant='one_policy_more_heads') ->
PreprocessAgent -> QAttentionStackAgent2Robots -> QAttentionPerActBCAgent2Robots.  update() on the F11 batch reproduces the
loss the REFERENCE encoder + two-arm loss gave (tests/golden/f11_encoder_2robots_tiny.npz); act() returns both arms' actions."""
import numpy as np
import pytest
import torch

from oracle import weights as ow
from voxactb_amd import synthetic
from voxactb_amd.agents.peract_bc import launch_utils as lu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CAMS = ['front', 'wrist']


def T(a):
    return torch.from_numpy(np.asarray(a))


def make_agent(g, training=True, se3=False):
    cfg = lu.default_cfg(method__voxel_sizes=[int(g['cfg_V'])], method__voxel_patch_size=int(g['cfg_k']),
                         method__voxel_patch_stride=int(g['cfg_s']), method__transformer_depth=int(g['cfg_depth']),
                         method__num_latents=int(g['cfg_latents']), replay__batch_size=int(g['cfg_B']),
                         method__input_dropout=0.0, method__attn_dropout=0.0, method__which_arm='both',
                         method__variant='one_policy_more_heads', rlbench__cameras=CAMS,
                         rlbench__camera_resolution=[int(g['cfg_H']), int(g['cfg_W'])])
    cfg.method.transform_augmentation.apply_se3 = se3
    agent = lu.create_agent(cfg)
    enc = agent._pose_agent._qattention_agents[0]._perceiver_encoder
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    agent.build(training=training, device=0)
    return agent, cfg


def two_arm_batch(g, seed=1):
    """the single-arm synthetic sample re-keyed for two arms: right = its labels / proprio, left = the fixture's."""
    B, V = int(g['cfg_B']), int(g['cfg_V'])
    rs = synthetic.make_replay_sample(B, CAMS, (int(g['cfg_H']), int(g['cfg_W'])), V, int(g['cfg_low_dim']), seed=seed)
    out = {k: v for k, v in rs.items() if k.endswith(('_rgb', '_point_cloud', '_camera_extrinsics', '_camera_intrinsics'))
           or k in ('lang_goal_emb', 'lang_token_embs', 'ignore_collisions')}
    out['low_dim_state_right_arm'] = rs['low_dim_state']
    out['low_dim_state_left_arm'] = T(g['proprio_left'])[:, None]
    out['trans_action_indicies_right'] = rs['trans_action_indicies']
    out['rot_grip_action_indicies_right'] = rs['rot_grip_action_indicies']
    out['gripper_pose_right'] = rs['gripper_pose']
    out['trans_action_indicies_left'] = T(g['trans_left']).int()[:, None]
    out['rot_grip_action_indicies_left'] = T(g['rot_grip_left']).int()[:, None]
    out['gripper_pose_left'] = rs['gripper_pose'].clone()
    return {k: v.to(DEV) for k, v in out.items()}


def test_update_reproduces_the_reference_loss_and_trains(golden):
    g = golden('f11_encoder_2robots_tiny')
    agent, _ = make_agent(g)
    qa = agent._pose_agent._qattention_agents[0]
    assert type(qa).__name__ == 'QAttentionPerActBCAgent2Robots' and type(agent._pose_agent).__name__ == 'QAttentionStackAgent2Robots'
    first = float(agent.update(0, two_arm_batch(g))['total_losses'])
    assert abs(first - float(g['loss'])) < 1e-4, (first, float(g['loss']))
    names = [n.replace('_qnet.module.', '') for n, _ in qa._q.named_parameters()]
    assert names == [str(n) for n in g['grad_names']]                       # reference parameter names and order
    losses = [first] + [float(agent.update(s, two_arm_batch(g))['total_losses']) for s in range(1, 6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses    # LAMB steps on one batch


def test_update_with_se3_augmentation_of_both_arms(golden):
    g = golden('f11_encoder_2robots_tiny')
    agent, _ = make_agent(g, se3=True)
    b = two_arm_batch(g)
    # keep both grippers well inside the scene so that the shared perturbation is accepted
    centre = torch.tensor(synthetic.SCENE_BOUNDS[:3]) * 0.5 + torch.tensor(synthetic.SCENE_BOUNDS[3:]) * 0.5
    for k in ('gripper_pose_right', 'gripper_pose_left'):
        b[k][..., :3] = centre.to(DEV)
    for s in range(3):
        assert np.isfinite(float(agent.update(s, b)['total_losses']))
    agent._pose_agent._qattention_agents[0]._check_se3_status()           # raises if the retry budget was exhausted


def test_act_returns_both_arms(golden):
    g = golden('f11_encoder_2robots_tiny')
    agent, _ = make_agent(g, training=False)
    rs = synthetic.make_replay_sample(1, CAMS, (int(g['cfg_H']), int(g['cfg_W'])), int(g['cfg_V']), int(g['cfg_low_dim']), seed=3)
    obs = {k: v.to(DEV) for k, v in rs.items() if k.endswith(('_rgb', '_point_cloud', '_camera_extrinsics', '_camera_intrinsics'))
           or k in ('lang_goal_emb', 'lang_token_embs')}
    obs['low_dim_state_right_arm'] = rs['low_dim_state'].to(DEV)
    obs['low_dim_state_left_arm'] = (rs['low_dim_state'] * 0.5).to(DEV)
    qa = agent._pose_agent._qattention_agents[0]
    res = qa.act(0, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in agent_obs(obs).items()})
    assert len(res.action) == 6 and res.action[0].shape == (1, 3) and res.action[4].shape == (1, 4)
    assert 'attention_coordinate_right' in res.observation_elements and 'attention_coordinate_left' in res.observation_elements
    for arm in ('right', 'left'):
        out = agent.act(0, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in obs.items()}, which_arm=arm)
        assert out.action.shape == (9,) and np.isfinite(out.action).all()
        idx = res.action[0 if arm == 'right' else 3][0].cpu().numpy()
        assert np.array_equal(out.observation_elements['trans_action_indicies'], idx)


def agent_obs(obs):
    """what PreprocessAgent.act hands to the stack agent (preprocess_agent.py:34-48): rgb scaled to [-1, 1], floats."""
    out = {}
    for k, v in obs.items():
        v = v.float()
        out[k] = (v / 255.0) * 2.0 - 1.0 if 'rgb' in k else v
    return out


def test_update_traces_of_the_reference_agent(golden):
    """three update() steps (LAMB, no augmentation, no dropout) against the losses and parameters the REFERENCE
    QAttentionPerActBCAgent2Robots produced for the same batches (tests/golden/make_golden.py, section f13)."""
    g = golden('f13_update_traces_2robots')
    agent, _ = make_agent(g)
    qa = agent._pose_agent._qattention_agents[0]
    B, V, low = int(g['cfg_B']), int(g['cfg_V']), int(g['cfg_low_dim'])
    got = []
    for step in range(3):
        seed = 10 + step
        rs = synthetic.make_replay_sample(B, CAMS, (int(g['cfg_H']), int(g['cfg_W'])), V, low, seed=seed)
        b = {k: v for k, v in rs.items() if k.endswith(('_rgb', '_point_cloud')) or k in ('lang_goal_emb', 'lang_token_embs', 'ignore_collisions')}
        b['low_dim_state_right_arm'] = rs['low_dim_state']
        b['low_dim_state_left_arm'] = ow.hashed_uniform('f13.proprio_left', (B, low), 0.0, 1.0, seed)[:, None]
        b['trans_action_indicies_right'] = rs['trans_action_indicies']
        b['rot_grip_action_indicies_right'] = rs['rot_grip_action_indicies']
        b['gripper_pose_right'] = rs['gripper_pose']
        b['trans_action_indicies_left'] = ow.hashed_int('f13.trans_left', (B, 3), 0, V, seed).int()[:, None]
        b['rot_grip_action_indicies_left'] = torch.cat((ow.hashed_int('f13.rot_left', (B, 3), 0, 72, seed),
                                                        ow.hashed_int('f13.grip_left', (B, 1), 0, 2, seed)), 1).int()[:, None]
        b['gripper_pose_left'] = rs['gripper_pose'].clone()
        r = agent.update(step, {k: v.to(DEV) for k, v in b.items()})
        sm = qa._summaries
        got.append([float(r['total_losses']), float(sm['losses/trans_loss']), float(sm['losses/rot_loss']),
                    float(sm['losses/grip_loss']), float(sm['losses/collision_loss'])])
    got, ref = np.array(got), g['losses']
    print(got[:, 0], ref[:, 0])
    assert np.abs(got[0] - ref[0]).max() < 1e-4                     # step 0: forward + loss parity
    assert np.abs(got - ref).max() < 5e-3                           # later steps pass through LAMB (see test_agent_gpu.py)
    names = [n.replace('_qnet.module.', '') for n, _ in qa._q.named_parameters()]
    assert names == [str(n) for n in g['param_names']]
    asum = np.array([float(p.detach().double().abs().sum()) for _, p in qa._q.named_parameters()])
    assert np.abs(asum - g['param_abs_sums']).max() / np.abs(g['param_abs_sums']).max() < 1e-3


def test_act_against_the_reference_stack_agent(golden):
    """QAttentionStackAgent2Robots.act for both arms against what the REFERENCE stack agent returned (section f14): voxel
    indices, rotation / gripper indices, attention coordinate, pixel coordinates, the continuous 9-vector, top-8 of softmax(q)."""
    g = golden('f14_act_2robots')
    agent, _ = make_agent(g, training=False)
    rs = synthetic.make_replay_sample(1, CAMS, (int(g['cfg_H']), int(g['cfg_W'])), int(g['cfg_V']), int(g['cfg_low_dim']), seed=23)
    obs = {}
    for c in CAMS:
        obs['%s_rgb' % c], obs['%s_point_cloud' % c] = rs['%s_rgb' % c], rs['%s_point_cloud' % c]
        obs['%s_camera_extrinsics' % c], obs['%s_camera_intrinsics' % c] = T(g[c + '_ext']), T(g[c + '_int'])
    obs['low_dim_state_right_arm'] = rs['low_dim_state']
    obs['low_dim_state_left_arm'] = T(g['low_dim_state_left_arm'])
    obs['lang_goal_emb'], obs['lang_token_embs'] = T(g['lang_goal_emb']), T(g['lang_token_embs'])
    obs = {k: v.to(DEV) for k, v in obs.items()}
    for arm in ('right', 'left'):
        res = agent.act(0, {k: v.clone() for k, v in obs.items()}, deterministic=True, which_arm=arm)
        assert np.array_equal(res.observation_elements['trans_action_indicies'], g[arm + '_trans_action_indicies'])
        assert np.array_equal(res.observation_elements['rot_grip_action_indicies'], g[arm + '_rot_grip_action_indicies'])
        assert np.abs(res.observation_elements['attention_coordinate_layer_0'] - g[arm + '_attention_coordinate']).max() < 1e-6
        assert np.abs(res.action - g[arm + '_continuous_action']).max() < 1e-6
        for c in CAMS:
            assert np.array_equal(np.array(res.observation_elements['%s_pixel_coord' % c], dtype=np.float64), g[arm + '_' + c + '_pixel_coord'])
        q = res.info['q_depth_%s0' % arm].reshape(1, -1).float().cpu()
        assert float((torch.gather(q, 1, T(g[arm + '_q_top_idx']).long()) - T(g[arm + '_q_top_vals'])).abs().max()) < 1e-5
