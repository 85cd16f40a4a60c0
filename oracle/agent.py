"""Oracle: QFunction / QAttentionPerActBCAgent numerics (TEST INFRASTRUCTURE).

Restates, from /root/reference/peract/agents/peract_bc/qattention_peract_bc_agent.py:
  argmax helpers        :57-80
  QFunction.forward     :82-135
  losses of update()    :511-582
  act() post-processing :394-416, :710-724
and /root/reference/peract/helpers/optim/lamb.py:60-124 (LAMB step).
"""
import torch
import torch.nn.functional as F

from . import perceiver as operc
from . import voxel_grid as ovox


def argmax_3d(q_trans):
    """agent :57-63 (assumes d == h == w, as the reference does)."""
    b, c, d, h, w = q_trans.shape
    idxs = q_trans.reshape(b, c, -1).argmax(-1)
    t = torch.div(idxs, h, rounding_mode='trunc')
    return torch.cat([torch.div(t, d, rounding_mode='trunc'), t % w, idxs % w], 1)


def choose_highest_action(q_trans, q_rot_grip, q_collision, rotation_resolution=5):
    """agent :65-80."""
    coords = argmax_3d(q_trans)
    n = int(360 // rotation_resolution)
    q_rot = torch.stack(torch.split(q_rot_grip[:, :-2], n, dim=1), dim=1)
    rot_and_grip = torch.cat([q_rot[:, 0:1].argmax(-1), q_rot[:, 1:2].argmax(-1),
                              q_rot[:, 2:3].argmax(-1),
                              q_rot_grip[:, -2:].argmax(-1, keepdim=True)], -1)
    ignore_collision = q_collision[:, -2:].argmax(-1, keepdim=True)
    return coords, rot_and_grip, ignore_collision


def qfunction_forward(P, pcd_list, rgb_list, proprio, lang_token_embs, bounds, V, **enc_kw):
    """agent :82-135.  Returns (q_trans, q_rot_grip, q_collision, voxel_grid[B,10,V,V,V][, arm])."""
    coords, feats = ovox.flatten_cameras(pcd_list, rgb_list)
    grid = ovox.voxelize(coords, feats, bounds, V)                 # :96-97
    grid_cf = grid.permute(0, 4, 1, 2, 3).detach()                 # :100
    outs = operc.forward(P, grid_cf, proprio, lang_token_embs, **enc_kw)
    return outs[:3] + (grid_cf,) + outs[3:]


def losses(q_trans, q_rot_grip, q_collision, action_trans, action_rot_grip,
           action_ignore_collisions, arm_out=None, action_label=None, num_rot=72):
    """agent :517-578 -- CE against integer labels (one-hot -> argmax is the identity).
    Returns (total, dict of per-head means)."""
    bs = q_trans.shape[0]
    V = q_trans.shape[-1]
    t = action_trans.long()
    flat_label = (t[:, 0] * V + t[:, 1]) * V + t[:, 2]
    l_trans = F.cross_entropy(q_trans.reshape(bs, -1), flat_label, reduction='none')
    r = action_rot_grip.long()
    l_rot = (F.cross_entropy(q_rot_grip[:, 0 * num_rot:1 * num_rot], r[:, 0], reduction='none')
             + F.cross_entropy(q_rot_grip[:, 1 * num_rot:2 * num_rot], r[:, 1], reduction='none')
             + F.cross_entropy(q_rot_grip[:, 2 * num_rot:3 * num_rot], r[:, 2], reduction='none'))
    l_grip = F.cross_entropy(q_rot_grip[:, 3 * num_rot:], r[:, 3], reduction='none')
    l_coll = F.cross_entropy(q_collision, action_ignore_collisions.long()[:, 0], reduction='none')
    comb = l_trans + l_rot + l_grip + l_coll
    parts = {'trans': l_trans.mean(), 'rot': l_rot.mean(), 'grip': l_grip.mean(), 'collision': l_coll.mean()}
    if arm_out is not None:
        l_arm = F.cross_entropy(arm_out, action_label.long()[:, 0], reduction='none')
        comb = comb + l_arm
        parts['arm'] = l_arm.mean()
    return comb.mean(), parts


def lamb_step(w, g, m, v, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=1e-6):
    """lamb.py:94-122 for one tensor; returns (w', m', v', trust_ratio).  No bias correction."""
    m = m * beta1 + (1 - beta1) * g
    v = v * beta2 + (1 - beta2) * g * g
    weight_norm = w.pow(2).sum().sqrt().clamp(0, 10)
    step = m / (v.sqrt() + eps)
    if weight_decay != 0:
        step = step + weight_decay * w
    adam_norm = step.pow(2).sum().sqrt()
    if weight_norm == 0 or adam_norm == 0:
        trust = 1.0
    else:
        trust = float(weight_norm / adam_norm)
    return w - lr * trust * step, m, v, trust


def softmax_heads(q_trans, q_rot_grip, q_collision, num_rot=72):
    """agent :394-416."""
    sh = q_trans.shape
    qt = F.softmax(q_trans.reshape(sh[0], -1), dim=1).reshape(sh)
    qr = torch.cat([F.softmax(q_rot_grip[:, i * num_rot:(i + 1) * num_rot], dim=1) for i in range(3)]
                   + [F.softmax(q_rot_grip[:, 3 * num_rot:], dim=1)], dim=1)
    qc = F.softmax(q_collision, dim=1)
    return qt, qr, qc


def attention_coordinate(bounds, coords, V):
    """agent :668, :723-724."""
    res = (bounds[:, 3:] - bounds[:, :3]) / V
    return bounds[:, :3] + res * coords.int() + res / 2


def train_steps(P, batches, V, n_steps, lr=5e-4, weight_decay=1e-6, **enc_kw):
    """Functional update() loop (agent :486-582, no augmentation, dropout 0).
    P: dict of leaf tensors (modified in place).  batches: list of dicts with keys
    pcd, rgb (lists of [B,3,H,W]), proprio, lang_token_embs, bounds, trans, rot_grip,
    ignore_collisions[, label].  Returns list of per-step dicts (loss + parts)."""
    names = list(P.keys())
    for k in names:
        P[k] = P[k].clone().requires_grad_(True)
    m = {k: torch.zeros_like(P[k]) for k in names}
    v = {k: torch.zeros_like(P[k]) for k in names}
    trace = []
    for i in range(n_steps):
        bt = batches[i % len(batches)]
        outs = qfunction_forward(P, bt['pcd'], bt['rgb'], bt['proprio'], bt['lang_token_embs'],
                                 bt['bounds'], V, **enc_kw)
        arm = outs[4] if len(outs) > 4 else None
        total, parts = losses(outs[0], outs[1], outs[2], bt['trans'], bt['rot_grip'],
                              bt['ignore_collisions'], arm, bt.get('label'))
        grads = torch.autograd.grad(total, [P[k] for k in names])
        with torch.no_grad():
            for k, g in zip(names, grads):
                w2, m[k], v[k], _ = lamb_step(P[k].detach(), g, m[k], v[k], lr=lr, weight_decay=weight_decay)
                P[k] = w2.requires_grad_(True)
        rec = {'total': float(total)}
        rec.update({kk: float(vv) for kk, vv in parts.items()})
        rec['grad_norms'] = {k: float(g.norm()) for k, g in zip(names, grads)}
        trace.append(rec)
    return trace
