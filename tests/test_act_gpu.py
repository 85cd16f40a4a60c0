"""GPU: `act()` through the whole agent stack (PreprocessAgent -> QAttentionStackAgent -> QAttentionPerActBCAgent) against
fixture F9 -- outputs of the REFERENCE stack's act() with a stub text encoder (tests/golden/make_golden.py:f9_act;
reference agent :643-787, stack agent :46-98, preprocess agent :34-48) at a small size and at the BASELINE.json configs[1]
geometry (V=100, depth 6, 2048 latents)."""
import numpy as np
import pytest
import torch

from oracle import weights as ow
from voxactb_amd import synthetic
from voxactb_amd.agents.peract_bc import launch_utils as lu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize('tag', ['s', 'c2'])
def test_act_matches_reference_fixture(golden, tag):
    g = golden('f9_act')
    pre = 'f9%s_' % tag
    c = {k: int(g[pre + 'cfg_' + k]) for k in ('V', 'k', 's', 'depth', 'latents', 'low_dim', 'H', 'W', 'ncam')}
    cams = (['front', 'wrist'] if c['ncam'] == 2 else synthetic.CAMERAS4[:c['ncam']])
    cfg = lu.default_cfg(method__voxel_sizes=[c['V']], method__voxel_patch_size=c['k'], method__voxel_patch_stride=c['s'],
                         method__transformer_depth=c['depth'], method__num_latents=c['latents'], replay__batch_size=1,
                         rlbench__cameras=cams, rlbench__camera_resolution=[c['H'], c['W']])
    agent = lu.create_agent(cfg)
    qa = agent._pose_agent._qattention_agents[0]
    enc = qa._perceiver_encoder
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 0), strict=False)
    agent.build(training=False, device=0)
    emb, tok = T(g[pre + 'lang_goal_emb']).to(DEV), T(g[pre + 'lang_token_embs']).to(DEV)
    seen = {}

    def text_encoder(tokens):                          # stands where upstream calls CLIP (agent :665)
        seen['tokens'] = tokens.detach().cpu()
        return emb, tok
    qa.set_text_encoder(text_encoder)
    rs = synthetic.make_replay_sample(1, cams, (c['H'], c['W']), c['V'], c['low_dim'], seed=21)
    obs = {}
    for cam in cams:
        obs['%s_rgb' % cam] = rs['%s_rgb' % cam].to(DEV)                         # [1,1,3,H,W], 0..255: PreprocessAgent normalises
        obs['%s_point_cloud' % cam] = rs['%s_point_cloud' % cam].to(DEV)
        obs['%s_camera_extrinsics' % cam] = T(g[pre + cam + '_ext']).to(DEV)
        obs['%s_camera_intrinsics' % cam] = T(g[pre + cam + '_int']).to(DEV)
    obs['low_dim_state'] = rs['low_dim_state'].to(DEV)
    obs['lang_goal_tokens'] = T(g[pre + 'lang_goal_tokens']).to(DEV)
    raw = {k: v.clone() for k, v in obs.items()}          # act() normalises the rgb entries of the dict it is given, in place
    res = agent.act(0, obs, deterministic=True)
    assert torch.equal(seen['tokens'].long().reshape(-1), T(g[pre + 'lang_goal_tokens']).long().reshape(-1))
    # discrete outputs: identical
    assert np.array_equal(res.observation_elements['trans_action_indicies'], g[pre + 'trans_action_indicies'])
    assert np.array_equal(res.observation_elements['rot_grip_action_indicies'], g[pre + 'rot_grip_action_indicies'])
    assert np.array_equal(res.info['voxel_idx_depth0'].cpu().numpy(), g[pre + 'coords'])
    for cam in cams:
        assert list(res.observation_elements['%s_pixel_coord' % cam]) == list(g[pre + cam + '_pixel_coord'])
    assert res.replay_elements == {'demo': False}
    # continuous 9-vector and the attention coordinate (bounds_min + res * idx + res / 2, agent :724)
    assert np.abs(np.asarray(res.action, np.float64) - g[pre + 'continuous_action']).max() < 1e-6
    assert np.abs(res.observation_elements['attention_coordinate_layer_0'] - g[pre + 'attention_coordinate']).max() < 1e-6
    # softmaxed translation Q: the reference's top-16 cells and 4096 sampled cells
    q = res.info['q_depth0'].reshape(1, -1).float().cpu()
    ti = T(g[pre + 'q_top_idx']).long()
    e_top = float((torch.gather(q, 1, ti) - T(g[pre + 'q_top_vals'])).abs().max())
    e_smp = float((q[:, T(g[pre + 'q_sample_idx']).long()] - T(g[pre + 'q_sample'])).abs().max())
    assert abs(float(q.double().sum()) - float(g[pre + 'q_sum'])) < 1e-4
    peak = float(T(g[pre + 'q_top_vals']).max())
    print('act %s: softmax(q_trans) top-16 err %.2e (peak %.3e), sample err %.2e' % (tag, e_top, peak, e_smp))
    assert e_top < 1e-4 * max(1.0, peak) and e_top < 2e-4 * peak + 1e-7 and e_smp < 1e-6
    # rotation / grip / collision heads, softmaxed the way act() does (agent :394-416)
    vox = res.info['voxel_grid_depth0']
    assert vox.shape == (1, 10, c['V'], c['V'], c['V'])
    o2 = {k: ((v.float() / 255.0) * 2.0 - 1.0 if 'rgb' in k else v.float()) for k, v in raw.items() if torch.is_tensor(v)}
    ob = [[o2['%s_rgb' % cam][0], o2['%s_point_cloud' % cam][0]] for cam in cams]
    pc = [o2['%s_point_cloud' % cam][0] for cam in cams]
    q_t, q_rg, q_c, _ = qa._q(ob, o2['low_dim_state'][0], pc, emb, tok, qa._coordinate_bounds, None, None)
    e_rg = float((qa._softmax_q_rot_grip(q_rg).cpu() - T(g[pre + 'q_rot_grip_softmax'])).abs().max())
    e_c = float((qa._softmax_ignore_collision(q_c).cpu() - T(g[pre + 'q_collision_softmax'])).abs().max())
    assert e_rg < 1e-4 and e_c < 1e-4, (e_rg, e_c)


def test_act_keeps_prepared_weights_until_they_change():
    """An evaluation agent keeps the prepared forms of its weights (bf16 planes / fragments, the polyphase W_eff) from one act() to the
    next -- no weight-preparation launch in the second call, identical outputs -- and drops them when the parameters change
    (load_state_dict bumps their version counters): the third call equals a fresh agent's with the new weights."""
    from voxactb_amd import _lib, ops
    cams = ['front', 'wrist']
    cfg = lu.default_cfg(method__voxel_sizes=[20], method__voxel_patch_size=5, method__voxel_patch_stride=5, method__transformer_depth=2,
                         method__num_latents=256, replay__batch_size=1, rlbench__cameras=cams, rlbench__camera_resolution=[32, 32])

    def make(seed):
        agent = lu.create_agent(cfg)
        enc = agent._pose_agent._qattention_agents[0]._perceiver_encoder
        enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, seed), strict=False)
        agent.build(training=False, device=0)
        return agent

    rs = synthetic.make_replay_sample(1, cams, (32, 32), 20, 4, seed=5)
    base = {k: v.to(DEV) for k, v in rs.items()
            if k.endswith(('_rgb', '_point_cloud', '_camera_extrinsics', '_camera_intrinsics')) or k == 'low_dim_state'}
    for cam in cams:
        base['%s_camera_extrinsics' % cam][0, 0, 2, 3] = -1.0
    base['lang_goal_emb'] = rs['lang_goal_emb'][0].to(DEV)
    base['lang_token_embs'] = rs['lang_token_embs'][0].to(DEV)

    def act(agent):
        res = agent.act(0, {k: v.clone() for k, v in base.items()}, deterministic=True)
        return res.info['q_depth0'].clone()

    a = make(0)
    eng = a._pose_agent._qattention_agents[0]._q.encoder.engine()
    assert eng.freeze_weight_prep
    q1 = act(a)
    assert eng._prep_sig is not None and ops.CACHE_OWNER is eng
    _lib.TIMER = _lib.KernelTimer()
    try:
        q2 = act(a)
        labels = set(_lib.TIMER.summary())
    finally:
        _lib.TIMER = None
    assert torch.equal(q1, q2)
    assert not ({'vxb_split_bf16_batch_f32', 'vxb_polyphase_weights_f32', 'vxb_gather_cvt_f32'} & labels), labels
    enc = a._pose_agent._qattention_agents[0]._q.encoder
    enc.load_state_dict(ow.hashed_state_dict({n: tuple(p.shape) for n, p in enc.named_parameters()}, 1), strict=False)
    q3 = act(a)
    assert not torch.equal(q3, q1)
    assert torch.equal(q3, act(make(1)))
    # another engine's forward takes the shared caches over: the first agent prepares again and still answers the same
    assert torch.equal(act(a), q3)
    # a long evaluation run: the address-keyed caches of prepared weights and the device memory stay where they are after the second call
    # (round-4 advisor finding: every act() re-laid the small conv weights into fresh temporaries, and gemm_wfrag cached a fragment copy
    # per temporary -- ~8 MB per call, never released while the preparation was reused)
    act(a)
    torch.cuda.synchronize()
    n_f, n_w, mem = len(ops._FCACHE), len(ops._WCACHE), torch.cuda.memory_allocated()
    for _ in range(20):
        q = act(a)
    torch.cuda.synchronize()
    assert torch.equal(q, q3)
    assert (len(ops._FCACHE), len(ops._WCACHE)) == (n_f, n_w), (len(ops._FCACHE), n_f, len(ops._WCACHE), n_w)
    assert torch.cuda.memory_allocated() <= mem + (1 << 20), (torch.cuda.memory_allocated(), mem)
