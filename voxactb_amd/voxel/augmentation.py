"""SE(3) augmentation of a replay batch, entirely on the device
(reference: peract/voxel/augmentation.py:68-185 `apply_se3_augmentation`, :7-65 `perturb_se3`).

What the reference does per training step: draw a translation shift and a discrete yaw (roll / pitch) per sample, compose
it with the keyframe gripper pose, re-discretise the action labels on the host (one scipy call per sample, three
device->host copies per attempt, retried while any label leaves the grid), then rewrite every camera's point cloud.

Here:
  * `se3_augmentation_plan` uploads the random draws of up to `attempts` attempts (torch CPU generator, like upstream's
    `rand_dist` / `rand_discrete`) and launches ONE small kernel (`vxb_se3_relabel_f32`, csrc/se3_relabel.hip) that does the
    pose algebra, both discretisations and the retry vote on the device.  It returns device tensors: the new labels and a
    [B, 15] rigid transform.  No host round trip; an exhausted retry budget poisons the labels (-> NaN loss) and sets a
    status word the agent checks one step later.
  * the point clouds are NOT rewritten: the transform is applied inside the voxelizer's point load
    (`VoxelGrid.voxelize_cameras(..., xform=)`), so the 4 perturbed clouds never exist.
  * `apply_se3_augmentation` keeps the reference signature and return values for callers that want the clouds
    (one `vxb_se3_points_f32` pass per camera).

There is no CPU path: CPU tensors raise `VoxactbHipError`.
"""
import torch

from .._lib import VoxactbHipError, call, require_cuda

MAX_ATTEMPTS = 100        # augmentation.py:119
MAX_ATTEMPTS_2ROBOTS = 400        # augmentation.py:239


_GEN = None          # dedicated CPU generator of the augmentation draws (seeded from torch's global seed at first use)
_GEN_SEED = None
_PINNED = {}         # (attempts, bs) -> ring of pinned staging buffers for the draws


def _generator():
    """Augmentation draws come from their OWN generator, seeded once from the global seed (so `torch.manual_seed(s)` before the
    first update() still governs them), instead of from the global CPU generator: unrelated `torch.rand` calls of the host
    program no longer shift the augmentation stream, and the dropout seed (drawn from the global generator) is independent."""
    global _GEN, _GEN_SEED
    if _GEN is None or _GEN_SEED != torch.initial_seed():
        # (re-derived whenever the global seed changes: a `torch.manual_seed(s)` on resume or per epoch governs the augmentation stream
        # again, as it does the reference's draws from the global generator, utils.rand_dist / rand_discrete)
        _GEN_SEED = torch.initial_seed()
        _GEN = torch.Generator()
        _GEN.manual_seed((_GEN_SEED ^ 0x5E3A06) & 0x7FFFFFFFFFFFFFFF)
    return _GEN


def seed_augmentation(seed: int):
    """Explicit seed of the augmentation stream (independent of torch's global generator until the next torch.manual_seed)."""
    global _GEN, _GEN_SEED
    _GEN_SEED = torch.initial_seed()
    _GEN = torch.Generator()
    _GEN.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)


def _draws(bs, rot_aug_range, rot_aug_resolution, attempts, device=None):
    """Random numbers of `attempts` attempts: shifts uniform in (-1, 1), integer angle steps in [-n, n] per axis with
    n = range // resolution (all zero for an axis whose range is smaller than one step).  The reference draws one attempt at a
    time (utils.rand_dist / rand_discrete, augmentation.py:123-141); all attempts are drawn up front here because the retry
    loop runs on the device.  With `device` the numbers are written into a ring of PINNED buffers and uploaded with
    non-blocking copies (a pageable .to(device) on the compute stream would serialise the host with the GPU every step)."""
    gen = _generator()
    if device is None or torch.device(device).type != 'cuda':
        unit = 2.0 * torch.rand((attempts, bs, 3), generator=gen) - 1.0
        steps = torch.zeros((attempts, bs, 3), dtype=torch.int32)
        for axis in range(3):
            n = int(rot_aug_range[axis] // rot_aug_resolution)
            if n > 0:
                steps[:, :, axis] = torch.randint(-n, n + 1, (attempts, bs), dtype=torch.int32, generator=gen)
        return unit, steps
    ring = _PINNED.get((attempts, bs))
    if ring is None:
        ring = _PINNED[(attempts, bs)] = dict(slot=0, bufs=[
            [torch.empty((attempts, bs, 3), dtype=torch.float32).pin_memory(), torch.zeros((attempts, bs, 3), dtype=torch.int32).pin_memory(),
             None] for _ in range(4)])
    buf = ring['bufs'][ring['slot']]
    ring['slot'] = (ring['slot'] + 1) % len(ring['bufs'])
    if buf[2] is not None:
        buf[2].synchronize()                     # the upload out of this staging pair four calls ago has completed
    buf[0].uniform_(-1.0, 1.0, generator=gen)
    for axis in range(3):
        n = int(rot_aug_range[axis] // rot_aug_resolution)
        if n > 0:
            buf[1][:, :, axis].random_(-n, n + 1, generator=gen)
        else:
            buf[1][:, :, axis].zero_()       # (the ring is keyed by (attempts, bs) only: another agent's rpy range may have filled this axis)
    unit, steps = buf[0].to(device, non_blocking=True), buf[1].to(device, non_blocking=True)
    buf[2] = torch.cuda.Event()
    buf[2].record(torch.cuda.current_stream(device))
    return unit, steps


def se3_augmentation_plan(action_gripper_pose, action_rot_grip, bounds, layer, trans_aug_range, rot_aug_range,
                          rot_aug_resolution, voxel_size, rot_resolution, device, attempts=MAX_ATTEMPTS, draws=None):
    """-> (trans_idx [B,3] int32, rot_grip_idx [B,4] int32, xform [B,15] float32, status [1] int32), all on `device`.

    `draws = (shift_unit [K,B,3] float32 in (-1,1), rpy_steps [K,B,3] int32)` replaces the random generator (tests)."""
    require_cuda(action_gripper_pose, action_rot_grip, bounds)
    bs = action_gripper_pose.shape[0]
    dev = action_gripper_pose.device
    if draws is None:
        draws = _draws(bs, rot_aug_range, rot_aug_resolution, attempts, dev)
    unit = draws[0].to(device=dev, dtype=torch.float32).contiguous()
    steps = draws[1].to(device=dev, dtype=torch.int32).contiguous()
    K = unit.shape[0]
    pose = action_gripper_pose.float().contiguous()
    rot_grip = action_rot_grip.to(torch.int32).contiguous()
    bnd = bounds.float().reshape(-1, 6).contiguous()
    if bnd.shape[0] not in (1, bs):
        raise VoxactbHipError('bounds must have 1 or B rows')
    aug = [float(v) for v in torch.as_tensor(trans_aug_range, dtype=torch.float64).reshape(-1)[:3]]
    trans_idx = torch.empty((bs, 3), dtype=torch.int32, device=dev)
    rot_idx = torch.empty((bs, 4), dtype=torch.int32, device=dev)
    xform = torch.empty((bs, 15), dtype=torch.float32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    call('vxb_se3_relabel_f32', pose, rot_grip, bnd, bnd.shape[0], int(layer), unit, steps, K, bs, aug[0], aug[1], aug[2],
         float(rot_aug_resolution), int(voxel_size), float(rot_resolution), trans_idx, rot_idx, xform, status)
    return trans_idx, rot_idx, xform, status


def transform_point_clouds(pcd, xform):
    """perturb_se3 (reference :7-65) with the [B,15] transform of the plan: one fused pass per camera."""
    out = []
    for p in pcd:
        require_cuda(p)
        pc = p.float().contiguous()
        o = torch.empty_like(pc)
        call('vxb_se3_points_f32', pc, o, xform, pc.shape[0], pc.numel() // (pc.shape[0] * 3))
        out.append(o)
    return out


def perturb_se3(pcd, trans_shift_4x4, rot_shift_4x4, action_gripper_4x4, bounds):
    """Reference name, signature and result (peract/voxel/augmentation.py:7-65): the point clouds of every camera moved by
    p' = (p - t) R + clamp(t + shift, scene extent), t = the keyframe gripper position.  The [B, 15] transform record of the fused kernel
    is put together from the three 4 x 4 matrices (B x 15 values of plumbing); the clouds go through vxb_se3_points_f32."""
    require_cuda(*pcd)
    dev = pcd[0].device
    bs = pcd[0].shape[0]
    bounds = bounds.to(dev).float().reshape(-1, 6)
    t = action_gripper_4x4[:, 0:3, 3].to(dev).float()
    shift = trans_shift_4x4[:, 0:3, 3].to(dev).float()
    rot = rot_shift_4x4.to(dev).float()
    lo, hi = bounds[:, 0:3].amin(0), bounds[:, 3:6].amax(0)                  # (:45-47: the extent over ALL rows of the bounds)
    c = torch.minimum(torch.maximum(t + shift, lo), hi) + rot[:, 3, 0:3]     # (row vectors: the matrix's last ROW adds on, zero for a rotation)
    xform = torch.cat([rot[:, 0:3, 0:3].reshape(bs, 9), t, c], dim=1).contiguous()
    return transform_point_clouds(pcd, xform)


def apply_se3_augmentation(pcd, action_gripper_pose, action_trans, action_rot_grip, bounds, layer, trans_aug_range,
                           rot_aug_range, rot_aug_resolution, voxel_size, rot_resolution, device):
    """Reference signature and return values (perturbed action_trans, action_rot_grip, pcd).  Unlike the fused path of the
    agent this one materialises the clouds and checks the retry status at once (one device->host word)."""
    trans_idx, rot_idx, xform, status = se3_augmentation_plan(
        action_gripper_pose.to(device), action_rot_grip.to(device), bounds.to(device), layer, trans_aug_range, rot_aug_range,
        rot_aug_resolution, voxel_size, rot_resolution, device)
    if int(status.item()) < 0:
        raise Exception('Failing to perturb action and keep it within bounds.')
    return trans_idx.to(action_trans.dtype if action_trans.dtype in (torch.int32, torch.int64) else torch.int64), \
        rot_idx.to(torch.int64), transform_point_clouds([p.to(device) for p in pcd], xform)


def se3_augmentation_plan_2robots(pose_right, rot_grip_right, pose_left, rot_grip_left, bounds, layer, trans_aug_range,
                                  rot_aug_range, rot_aug_resolution, voxel_size, rot_resolution, device,
                                  attempts=MAX_ATTEMPTS_2ROBOTS, draws=None):
    """Both arms of the `one_policy_more_heads` baseline under ONE perturbation (reference :187-348): the same shift and
    rotation for both keyframe poses, re-drawn while either arm's translation label leaves the grid, the cloud transform
    centred on the right arm.  -> (trans_right, rot_grip_right, trans_left, rot_grip_left, xform [B,15], status [1])."""
    require_cuda(pose_right, rot_grip_right, pose_left, rot_grip_left, bounds)
    bs = pose_right.shape[0]
    dev = pose_right.device
    if draws is None:
        draws = _draws(bs, rot_aug_range, rot_aug_resolution, attempts, dev)
    unit = draws[0].to(device=dev, dtype=torch.float32).contiguous()
    steps = draws[1].to(device=dev, dtype=torch.int32).contiguous()
    bnd = bounds.float().reshape(-1, 6).contiguous()
    if bnd.shape[0] not in (1, bs):
        raise VoxactbHipError('bounds must have 1 or B rows')
    aug = [float(v) for v in torch.as_tensor(trans_aug_range, dtype=torch.float64).reshape(-1)[:3]]
    out = [torch.empty((bs, n), dtype=torch.int32, device=dev) for n in (3, 4, 3, 4)]
    xform = torch.empty((bs, 15), dtype=torch.float32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    call('vxb_se3_relabel_pair_f32', pose_right.float().contiguous(), rot_grip_right.to(torch.int32).contiguous(),
         pose_left.float().contiguous(), rot_grip_left.to(torch.int32).contiguous(), bnd, bnd.shape[0], int(layer), unit, steps,
         unit.shape[0], bs, aug[0], aug[1], aug[2], float(rot_aug_resolution), int(voxel_size), float(rot_resolution),
         out[0], out[1], out[2], out[3], xform, status)
    return out[0], out[1], out[2], out[3], xform, status


def apply_se3_augmentation_2Robots(pcd, action_gripper_pose_right, action_trans_right, action_rot_grip_right,
                                   action_gripper_pose_left, action_trans_left, action_rot_grip_left, bounds, layer,
                                   trans_aug_range, rot_aug_range, rot_aug_resolution, voxel_size, rot_resolution, device):
    """Reference signature and return values (:187-201, :348)."""
    tr, rr, tl, rl, xform, status = se3_augmentation_plan_2robots(
        action_gripper_pose_right.to(device), action_rot_grip_right.to(device), action_gripper_pose_left.to(device),
        action_rot_grip_left.to(device), bounds.to(device), layer, trans_aug_range, rot_aug_range, rot_aug_resolution,
        voxel_size, rot_resolution, device)
    if int(status.item()) < 0:
        raise Exception('Failing to perturb action and keep it within bounds.')
    return tr.to(torch.int64), rr.to(torch.int64), tl.to(torch.int64), rl.to(torch.int64), \
        transform_point_clouds([p.to(device) for p in pcd], xform)
