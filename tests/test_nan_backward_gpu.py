"""A NaN born in the backward pass must not turn into finite weights.

The reference's backward is plain autograd (qattention_peract_bc_agent.py:578-590): a non-finite loss gradient propagates to every
parameter upstream of it and the optimizer writes NaN weights -- visible at once.  The default precision of this framework converts
gradient operands to fp16 with a saturating v_med3_f32, which maps NaN to -65504: without care a NaN would become finite garbage that the
optimizer applies silently.  Two mechanisms keep the reference's behaviour (csrc/common.h: vxb_sat_f16, vxb_amax_word; nn_ops.hip:
absmax_final_kernel / wgrad_finish_kernel): the operand scale taken from a tensor's largest magnitude becomes NaN when the tensor holds a
NaN or inf (every consumer multiplies by it), and the kernels whose scale is one step old convert with a NaN-preserving saturation.

Checked at configs[1] geometry (fixture f5g's batch: the kernels bench.py's headline dispatches, also through the wide kernels): a NaN
injected into d(q_trans) or into the MLP heads' gradient reaches every parameter the reference's autograd would reach."""
import numpy as np
import pytest
import torch

from tests.test_c2_reference_gpu import DEV, _setup
from voxactb_amd import ops

pytestmark = pytest.mark.gpu

HEADS = ('dense0.', 'dense1.', 'rot_grip_collision_ff.')


@pytest.mark.parametrize('wide_dispatch', [False, True], indirect=True, ids=['default', 'wide'])
@pytest.mark.parametrize('site', ['dq_trans', 'd_heads'])
def test_nan_in_a_loss_gradient_reaches_every_upstream_parameter(golden, site, wide_dispatch):
    g = golden('f5g_encoder_c2_grads')
    enc, rs, grid, arm, V, B = _setup(g)
    eng = enc.engine()
    eng.precision = 'bf16x3'
    outs, cache = eng.forward(grid, rs['low_dim_state'].to(DEV), rs['lang_token_embs'].to(DEV), training=False, save=True)
    at = rs['trans_action_indicies'].long()
    lab = ((at[:, 0] * V + at[:, 1]) * V + at[:, 2]).int().to(DEV)
    dq = torch.empty((B, V ** 3), device=DEV)
    ops.ce_big(outs[0].view(B, -1), lab, dq, 1.0 / B)
    labs = torch.cat([rs['rot_grip_action_indicies'].int(), rs['ignore_collisions'].int()[:, :1]], 1).to(DEV).contiguous()
    d_o = torch.empty_like(cache['o'])
    ops.ce_rows(cache['o'], [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)], labs, d_o, 1.0 / B)
    if site == 'dq_trans':
        dq[0, (37 * V + 41) * V + 43] = float('nan')          # one voxel of the translation head's gradient
    else:
        d_o[0, 5] = float('nan')                                # one logit of the rotation head's gradient
    for p in enc.parameters():
        p.grad = None
    eng.backward(cache, dq, d_o, None)
    # what the reference's autograd reaches: from d(q_trans) everything except the MLP heads (they do not depend on q_trans); from the
    # heads' gradient everything except trans_decoder (its gradient is dq (x) u)
    finite = []
    for n, p in enc.named_parameters():
        reached = not n.startswith(HEADS) if site == 'dq_trans' else not n.startswith('trans_decoder.')
        bad = int((~torch.isfinite(p.grad)).sum()) if p.grad is not None else 0
        if reached and bad == 0:
            finite.append(n)
        if not reached:
            assert bad == 0, (n, 'a gradient that does not depend on the poisoned value must stay finite')
    print('%s%s: parameters whose gradient stayed finite although the reference would hand them NaN: %s'
          % (site, '|wide' if wide_dispatch else '', finite))
    assert not finite, finite


def test_a_nan_step_poisons_the_weights_like_the_reference(golden):
    """update() with a NaN in the batch's proprioception: the loss is NaN and so are the weights after the step (reference: NaN loss ->
    NaN gradients -> LAMB writes NaN) -- never a finite loss over silently corrupted weights."""
    from tests.test_agent_gpu import make_agent, raw_batch
    g = golden('f6_update_traces')
    agent, _ = make_agent(g, 'a')
    batch = raw_batch(g, 'a', 10)
    batch['low_dim_state'] = batch['low_dim_state'].clone()
    batch['low_dim_state'][0, 0, 0] = float('nan')
    r = agent.update(0, batch)
    assert not np.isfinite(float(r['total_losses']))
    qa = agent._pose_agent._qattention_agents[0]
    n_bad = sum(int((~torch.isfinite(p)).sum()) for p in qa._q.parameters())
    assert n_bad > 0
