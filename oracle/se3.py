"""Oracle: SE(3) augmentation (TEST INFRASTRUCTURE).

Restates /root/reference/peract/voxel/augmentation.py:7-185 and the helpers it
uses from /root/reference/peract/helpers/utils.py (:63-64 normalize_quaternion,
:92-97 quaternion_to_discrete_euler, :104-116 point_to_voxel_index).

PINNED BY A REFERENCE RUN (fixture F8): tests/golden/make_golden.py path-loads the reference's own
voxel/augmentation.py and runs `apply_se3_augmentation` / `apply_se3_augmentation_2Robots` with scripted draws
(shared and per-sample bounds, layer 0 / 1, a forced whole-batch retry, two arms); `augment` below reproduces its
labels exactly and its transformed points bit for bit.

ONE PIECE STAYS UNPINNED UPSTREAM: the reference calls pytorch3d==0.3.0
(peract/requirements.txt:11; call sites augmentation.py:106,142,152), which is
neither vendored under /root/reference nor installed here, so that reference run uses the three helpers
below, restated from pytorch3d 0.3.0's published definitions:
  quaternion_to_matrix(wxyz)      : real-first, scaled by 2/|q|^2
  euler_angles_to_matrix(a,'XYZ') : Rx(a0) @ Ry(a1) @ Rz(a2)
  matrix_to_quaternion            : 0.5*sqrt(max(0, 1 +- m00 +- m11 +- m22)) with
                                    copysign from the off-diagonal differences
These three are checked by invariants only (orthonormality, round trip, scipy agreement).
The random draws (`torch.rand` / `torch.randint` on CPU, augmentation.py:123-141)
are passed in explicitly so that tests are deterministic.
"""
import numpy as np
import torch
from scipy.spatial.transform import Rotation


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _axis_rot(axis, a):
    c, s = torch.cos(a), torch.sin(a)
    one, zero = torch.ones_like(a), torch.zeros_like(a)
    if axis == 'X':
        f = (one, zero, zero, zero, c, -s, zero, s, c)
    elif axis == 'Y':
        f = (c, zero, s, zero, one, zero, -s, zero, c)
    else:
        f = (c, -s, zero, s, c, zero, zero, zero, one)
    return torch.stack(f, -1).reshape(a.shape + (3, 3))


def euler_angles_to_matrix(e, convention='XYZ'):
    ms = [_axis_rot(c, a) for c, a in zip(convention, torch.unbind(e, -1))]
    return ms[0] @ ms[1] @ ms[2]


def matrix_to_quaternion(m):
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]

    def sp(x):
        return torch.sqrt(torch.clamp(x, min=0))
    o0 = 0.5 * sp(1 + m00 + m11 + m22)
    x = 0.5 * sp(1 + m00 - m11 - m22)
    y = 0.5 * sp(1 - m00 + m11 - m22)
    z = 0.5 * sp(1 - m00 - m11 + m22)
    o1 = torch.copysign(x, m[..., 2, 1] - m[..., 1, 2])
    o2 = torch.copysign(y, m[..., 0, 2] - m[..., 2, 0])
    o3 = torch.copysign(z, m[..., 1, 0] - m[..., 0, 1])
    return torch.stack((o0, o1, o2, o3), -1)


def point_to_voxel_index(point, voxel_size, coord_bounds):
    """utils.py:104-116 (float64 numpy; clipped only from above)."""
    bb_mins = np.array(coord_bounds[0:3])
    bb_maxs = np.array(coord_bounds[3:])
    dims_m_one = np.array([voxel_size] * 3) - 1
    res = (bb_maxs - bb_mins) / (np.array([voxel_size] * 3) + 1e-12)
    return np.minimum(np.floor((point - bb_mins) / (res + 1e-12)).astype(np.int32), dims_m_one)


def quaternion_to_discrete_euler(quat_xyzw, resolution):
    """utils.py:92-97."""
    euler = Rotation.from_quat(quat_xyzw).as_euler('xyz', degrees=True) + 180
    disc = np.around(euler / resolution).astype(int)
    disc[disc == int(360 / resolution)] = 0
    return disc


def discrete_euler_to_quaternion(disc, resolution):
    """utils.py:100-102."""
    return Rotation.from_euler('xyz', (np.asarray(disc) * resolution) - 180, degrees=True).as_quat()


def perturb_points(pcd_list, shift, rot3, grip_trans, bounds):
    """augmentation.py:7-65.  pcd_list: [B,3,H,W] each; shift [B,3]; rot3 [B,3,3];
    grip_trans [B,3]; bounds [1|B,6].  p' = ((p - t)^T R)^T + clamp(t + shift)."""
    bs = pcd_list[0].shape[0]
    if bounds.shape[0] != bs:
        bounds = bounds.repeat(bs, 1)
    lo = torch.stack([bounds[:, 0].min(), bounds[:, 1].min(), bounds[:, 2].min()])
    hi = torch.stack([bounds[:, 3].max(), bounds[:, 4].max(), bounds[:, 5].max()])
    centre = torch.max(torch.min(grip_trans + shift, hi), lo)          # :49-57
    out = []
    for p in pcd_list:
        flat = p.reshape(bs, 3, -1) - grip_trans.unsqueeze(-1)         # :37
        rot = torch.bmm(flat.transpose(2, 1), rot3).transpose(2, 1)    # :41-42 (row-vector convention)
        out.append((rot + centre.unsqueeze(-1)).reshape(p.shape))      # :60-62
    return out


def augment(pcd_list, gripper_pose, rot_grip, bounds, shift_unit, rpy_steps,
            trans_aug_range, rot_aug_resolution, voxel_size, rot_resolution, layer=0):
    """One attempt of augmentation.py:116-177 with explicit random draws.
    gripper_pose [B,7] xyz+quat(xyzw); shift_unit [B,3] in (-1,1); rpy_steps [B,3] ints.
    Returns (trans_idx [B,3] int64, rot_grip_idx [B,4] int64, perturbed pcd list, ok flag)."""
    bs = pcd_list[0].shape[0]
    T = torch.eye(4).repeat(bs, 1, 1)
    q_wxyz = torch.cat([gripper_pose[:, 6:7], gripper_pose[:, 3:6]], 1)
    T[:, :3, :3] = quaternion_to_matrix(q_wxyz)                         # :106
    T[:, :3, 3] = gripper_pose[:, :3]
    # `trans_aug_range` is a float64 tensor upstream (agent :186, torch.from_numpy(np.array(...))): the range and the shift
    # are float64 (type promotion), the 4x4 that carries the shift to perturb_se3 is float32 again
    trans_range = (bounds[:, 3:] - bounds[:, :3]) * torch.as_tensor(trans_aug_range, dtype=torch.float64)
    shift64 = trans_range * shift_unit                                  # :123-124
    shift = shift64.float()                                             # :125-126 (assignment into the fp32 identity)
    ang = rpy_steps.float() * np.deg2rad(rot_aug_resolution)           # :133-141
    R3 = euler_angles_to_matrix(ang, 'XYZ')                              # :142
    R4 = torch.eye(4).repeat(bs, 1, 1)
    R4[:, :3, :3] = R3
    Tp = torch.bmm(T, R4)                                               # :147
    Tp[:, :3, 3] += shift64                                             # :148 (computed in float64, stored as fp32)
    tr = Tp[:, :3, 3].numpy()
    qw = matrix_to_quaternion(Tp[:, :3, :3])                            # :152
    q_xyzw = torch.cat([qw[:, 1:], qw[:, 0:1]], 1).numpy()
    ti, ri = [], []
    for b in range(bs):
        bnp = bounds[b if layer > 0 else 0].numpy()                    # :161-162
        ti.append(point_to_voxel_index(tr[b], voxel_size, bnp).tolist())
        quat = q_xyzw[b] / np.linalg.norm(q_xyzw[b], axis=-1, keepdims=True)
        if quat[-1] < 0:
            quat = -quat
        ri.append(quaternion_to_discrete_euler(quat, rot_resolution).tolist() + [int(rot_grip[b, 3])])
    ti = torch.from_numpy(np.array(ti))
    ri = torch.from_numpy(np.array(ri))
    ok = not bool(torch.any(ti < 0))                                    # :116
    pcd = perturb_points(pcd_list, shift, R3, gripper_pose[:, :3], bounds)  # :182
    return ti, ri, pcd, ok
