"""GPU: the bf16 matrix-core throughput mode (VOXACTB_PRECISION=bf16 / engine.precision='bf16') against the fp32 reference
fixtures.  bf16 carries 8 mantissa bits, so this mode is NOT held to the 1e-4 Q-value bound (that is the fp32 mode's
contract, tests/test_encoder_gpu.py); it must stay within bf16 rounding of it and pick (nearly) the same actions."""
import numpy as np
import pytest
import torch

from voxactb_amd import ops
from tests.test_encoder_gpu import build_encoder, batch, T, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('fixture,cams', [('f3_encoder_c1', ['front'])])
def test_bf16_mode_close_to_fp32(golden, fixture, cams):
    g = golden(fixture)
    enc, _ = build_encoder(g, bool(g['cfg_arm']))
    eng = enc.engine()
    eng.precision = 'bf16'
    rs = batch(g, cams)
    V, B = int(g['cfg_V']), int(g['cfg_B'])
    grid = T(g['grid']).to(DEV)
    outs, cache = eng.forward(grid, rs['low_dim_state'].to(DEV), rs['lang_token_embs'].to(DEV), training=False, save=True)
    q, qr = outs[0].cpu().reshape(B, -1), T(g['q_trans']).reshape(B, -1)
    err = float((q - qr).abs().max())
    scale = float(qr.abs().max())
    print('bf16 mode: q_trans max-abs err %.3e (|q| max %.3f), rot_grip err %.3e' %
          (err, scale, float((outs[1].cpu() - T(g['rot_grip'])).abs().max())))
    assert err < 0.03 * scale + 0.02
    # the fp32 argmax voxel must be among the top-5 of the bf16 logits
    top = q.topk(5, dim=1).indices
    assert all(int(qr[b].argmax()) in top[b].tolist() for b in range(B))
    at = rs['trans_action_indicies'].long()
    lab = ((at[:, 0] * V + at[:, 1]) * V + at[:, 2]).int().to(DEV)
    dq = torch.empty((B, V ** 3), device=DEV)
    l_t, _, _ = ops.ce_big(outs[0].view(B, -1), lab, dq, 1.0 / B)
    labs = torch.cat([rs['rot_grip_action_indicies'].int(), rs['ignore_collisions'].int()[:, :1]], 1).to(DEV).contiguous()
    d_o = torch.empty_like(cache['o'])
    l_h, _ = ops.ce_rows(cache['o'], [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)], labs, d_o, 1.0 / B)
    loss = float((l_t + l_h.sum(1)).mean())
    assert abs(loss - float(g['loss'])) < 0.05, (loss, float(g['loss']))
    for p in enc.parameters():
        p.grad = None
    eng.backward(cache, dq, d_o, None)
    P = dict(enc.named_parameters())
    worst = 1.0
    for n in [str(x) for x in g['grad_names']]:
        key = 'grad__' + n
        if key in g.files:
            ref = T(g[key]).flatten()
            mine = P[n].grad.detach().cpu().flatten()
            if float(ref.norm()) > 1e-4:
                cos = float(torch.dot(ref, mine) / (ref.norm() * mine.norm() + 1e-20))
                worst = min(worst, cos)
    print('bf16 mode: worst gradient cosine vs fp32 reference %.4f' % worst)
    assert worst > 0.95
    assert ops.PRECISION == 'fp32'       # the mode never leaks out of the engine call


def test_bf16_update_steps_run(golden):
    from tests.test_agent_gpu import make_agent, raw_batch
    g = golden('f6_update_traces')
    agent, _ = make_agent(g, 'a')
    qa = agent._pose_agent._qattention_agents[0]
    qa._q.encoder.engine().precision = 'bf16'
    ref = g['a_losses'][:, 0]
    got = [float(agent.update(s, raw_batch(g, 'a', 10 + s))['total_losses']) for s in range(3)]
    print(got, ref)
    assert np.abs(np.array(got) - ref).max() < 0.1
