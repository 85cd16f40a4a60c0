"""SE(3) augmentation of point clouds + relabelling of the discrete action
(reference: peract/voxel/augmentation.py:7-185; helpers peract/helpers/utils.py:63-116,501-508).

Same algorithm, same random streams (CPU `torch.rand` / `torch.randint`, utils.py:501-508), same quirks:
every sample is discretised with `bounds[0]` when layer == 0 (:161-162), points are rotated as row vectors
(:41-42), the translated origin is clamped to the batch-wide bounds (:44-57), the whole batch is re-drawn while
any translation index is negative (:116), at most 100 attempts (:119-120).

The three pytorch3d==0.3.0 helpers the reference imports (not vendored upstream) are restated from their published
definition -- see oracle/se3.py for the pinning caveat.  The 4x4 / label arithmetic runs on the host (B tiny
evaluations, one device -> host copy of the poses); the point transform is one fused HIP kernel per camera.  Fusing it
into the voxelizer's point load is the first "next" row of SURVEY.md section 8(f).
"""
import numpy as np
import torch
from scipy.spatial.transform import Rotation


def rand_dist(size, min=-1.0, max=1.0):
    return (max - min) * torch.rand(size) + min


def rand_discrete(size, min=0, max=1):
    if min == max:
        return torch.zeros(size)
    return torch.randint(min, max + 1, size)


def normalize_quaternion(quat):
    return np.array(quat) / np.linalg.norm(quat, axis=-1, keepdims=True)


def quaternion_to_discrete_euler(quaternion, resolution):
    euler = Rotation.from_quat(quaternion).as_euler('xyz', degrees=True) + 180
    assert np.min(euler) >= 0 and np.max(euler) <= 360
    disc = np.around((euler / resolution)).astype(int)
    disc[disc == int(360 / resolution)] = 0
    return disc


def discrete_euler_to_quaternion(discrete_euler, resolution):
    euluer = (discrete_euler * resolution) - 180
    return Rotation.from_euler('xyz', euluer, degrees=True).as_quat()


def point_to_voxel_index(point, voxel_size, coord_bounds):
    bb_mins = np.array(coord_bounds[0:3])
    bb_maxs = np.array(coord_bounds[3:])
    dims_m_one = np.array([voxel_size] * 3) - 1
    bb_ranges = bb_maxs - bb_mins
    res = bb_ranges / (np.array([voxel_size] * 3) + 1e-12)
    return np.minimum(np.floor((point - bb_mins) / (res + 1e-12)).astype(np.int32), dims_m_one)


def quaternion_to_matrix(q):
    """pytorch3d 0.3.0 semantics: real-first quaternion, scaled by 2/|q|^2."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def euler_angles_to_matrix(e, convention='XYZ'):
    mats = []
    for axis, a in zip(convention, torch.unbind(e, -1)):
        c, s = torch.cos(a), torch.sin(a)
        one, zero = torch.ones_like(a), torch.zeros_like(a)
        f = {'X': (one, zero, zero, zero, c, -s, zero, s, c),
             'Y': (c, zero, s, zero, one, zero, -s, zero, c),
             'Z': (c, -s, zero, s, c, zero, zero, zero, one)}[axis]
        mats.append(torch.stack(f, -1).reshape(a.shape + (3, 3)))
    return mats[0] @ mats[1] @ mats[2]


def matrix_to_quaternion(m):
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]

    def sp(x):
        return torch.sqrt(torch.clamp(x, min=0))
    o0 = 0.5 * sp(1 + m00 + m11 + m22)
    x, y, z = 0.5 * sp(1 + m00 - m11 - m22), 0.5 * sp(1 - m00 + m11 - m22), 0.5 * sp(1 - m00 - m11 + m22)
    return torch.stack((o0, torch.copysign(x, m[..., 2, 1] - m[..., 1, 2]), torch.copysign(y, m[..., 0, 2] - m[..., 2, 0]),
                        torch.copysign(z, m[..., 1, 0] - m[..., 0, 1])), -1)


def perturb_se3(pcd, trans_shift_4x4, rot_shift_4x4, action_gripper_4x4, bounds):
    """reference :7-65.  pcd: list of [bs,3,H,W].  The pose matrices / bounds may live on the host (that is where
    apply_se3_augmentation does its B tiny 4x4 evaluations) while the clouds are on the device: rotation, gripper position
    and the clamped new centre go up as ONE [bs, 15] tensor and every camera is transformed by one fused kernel
    (vxb_se3_points_f32) instead of a reshape / subtract / bmm / transpose / add chain."""
    bs = pcd[0].shape[0]
    if bounds.shape[0] != bs:
        bounds = bounds.repeat(bs, 1)
    lo = torch.stack([bounds[:, 0].min(), bounds[:, 1].min(), bounds[:, 2].min()])
    hi = torch.stack([bounds[:, 3].max(), bounds[:, 4].max(), bounds[:, 5].max()])
    t_grip = action_gripper_4x4[:, 0:3, 3]
    centre = torch.max(torch.min(t_grip + trans_shift_4x4[:, 0:3, 3], hi), lo)
    R = rot_shift_4x4[:, :3, :3]
    if pcd[0].is_cuda:
        from .._lib import call
        xf = torch.cat([R.reshape(bs, 9), t_grip, centre], dim=1).float().contiguous().to(pcd[0].device)
        out = []
        for p in pcd:
            pc = p.float().contiguous()
            o = torch.empty_like(pc)
            call('vxb_se3_points_f32', pc, o, xf, bs, pc.numel() // (bs * 3))
            out.append(o)
        return out
    out = []
    for p in pcd:
        flat = p.reshape(bs, 3, -1) - t_grip.unsqueeze(-1)
        rot = torch.bmm(flat.transpose(2, 1), R).transpose(2, 1)
        out.append((rot + centre.unsqueeze(-1)).reshape(p.shape))
    return out


def apply_se3_augmentation(pcd, action_gripper_pose, action_trans, action_rot_grip, bounds, layer, trans_aug_range,
                           rot_aug_range, rot_aug_resolution, voxel_size, rot_resolution, device):
    """reference :68-185.  The pose / label arithmetic (B 4x4 matrices, one scipy call per sample) runs on the HOST --
    upstream evaluates it with ~80 tiny device kernels and several device->host copies per attempt, which on this path
    was 5 ms of an otherwise idle GPU at the start of every step; the random streams were host-side already.  Only the
    point clouds are touched on the device."""
    bs = pcd[0].shape[0]
    host = torch.device('cpu')
    pose_h = action_gripper_pose.detach().to(host).float()
    bounds_h = bounds.detach().to(host).float()
    grip_np = action_rot_grip.detach().cpu().numpy()
    identity_4x4 = torch.eye(4).unsqueeze(0).repeat(bs, 1, 1)
    q_wxyz = torch.cat((pose_h[:, 6].unsqueeze(1), pose_h[:, 3:6]), dim=1)
    action_gripper_4x4 = identity_4x4.detach().clone()
    action_gripper_4x4[:, :3, :3] = quaternion_to_matrix(q_wxyz)
    action_gripper_4x4[:, 0:3, 3] = pose_h[:, :3]
    perturbed_trans = torch.full(tuple(action_trans.shape), -1.)
    perturbed_rot_grip = torch.full(tuple(action_rot_grip.shape), -1.)
    bounds_np = bounds_h.numpy()
    attempts = 0
    while torch.any(perturbed_trans < 0):
        attempts += 1
        if attempts > 100:
            raise Exception('Failing to perturb action and keep it within bounds.')
        trans_range = (bounds_h[:, 3:] - bounds_h[:, :3]) * trans_aug_range.to(host)
        trans_shift = trans_range * rand_dist((bs, 3))
        trans_shift_4x4 = identity_4x4.detach().clone()
        trans_shift_4x4[:, 0:3, 3] = trans_shift
        steps = [int(r // rot_aug_resolution) for r in rot_aug_range]
        rpy = [rand_discrete((bs, 1), min=-n, max=n) * np.deg2rad(rot_aug_resolution) for n in steps]
        rot_shift_3x3 = euler_angles_to_matrix(torch.cat(rpy, dim=1).float(), "XYZ")
        rot_shift_4x4 = identity_4x4.detach().clone()
        rot_shift_4x4[:, :3, :3] = rot_shift_3x3
        perturbed = torch.bmm(action_gripper_4x4, rot_shift_4x4)
        perturbed[:, 0:3, 3] += trans_shift
        p_trans = perturbed[:, 0:3, 3].numpy()
        q = matrix_to_quaternion(perturbed[:, :3, :3])
        q_xyzw = torch.cat([q[:, 1:], q[:, 0].unsqueeze(1)], dim=1).numpy()
        trans_idx, rot_grip_idx = [], []
        for b in range(bs):
            bnp = bounds_np[b if layer > 0 else 0]
            trans_idx.append(point_to_voxel_index(p_trans[b], voxel_size, bnp).tolist())
            quat = normalize_quaternion(q_xyzw[b])
            if quat[-1] < 0:
                quat = -quat
            rot_grip_idx.append(quaternion_to_discrete_euler(quat, rot_resolution).tolist() + [int(grip_np[b, 3])])
        perturbed_trans = torch.from_numpy(np.array(trans_idx))
        perturbed_rot_grip = torch.from_numpy(np.array(rot_grip_idx))
    pcd = perturb_se3(pcd, trans_shift_4x4, rot_shift_4x4, action_gripper_4x4, bounds_h)
    return perturbed_trans.to(device=device), perturbed_rot_grip.to(device=device), pcd
