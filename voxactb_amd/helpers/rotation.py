"""Host-side label arithmetic of the discrete action space (numpy, float64)
(reference: peract/helpers/utils.py:63-64 normalize_quaternion, :92-97 quaternion_to_discrete_euler, :100-102
discrete_euler_to_quaternion, :104-116 point_to_voxel_index, :126-136 point_to_pixel_index).

The reference goes through scipy's `Rotation`; these are closed forms of the same conventions -- extrinsic x-y-z Euler
angles, quaternions as (x, y, z, w) -- vectorised over leading axes, so the replay fill and `act()` need no scipy object
per keyframe.  tests/test_rotation_cpu.py checks them against scipy on random and boundary inputs; the device twin of the
first two lives in csrc/se3_relabel.hip.
"""
import numpy as np


def normalize_quaternion(quat):
    q = np.asarray(quat, dtype=np.float64) if not isinstance(quat, np.ndarray) else quat
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def quaternion_to_euler_xyz(quat_xyzw):
    """(x, y, z, w) -> extrinsic x-y-z angles in radians, each in (-pi, pi]; gimbal lock puts the twist in the first angle."""
    q = np.asarray(quat_xyzw, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r00 = 1.0 - 2.0 * (y * y + z * z)
    r10 = 2.0 * (x * y + w * z)
    r20 = 2.0 * (x * z - w * y)
    r21 = 2.0 * (y * z + w * x)
    r22 = 1.0 - 2.0 * (x * x + y * y)
    sb = np.clip(-r20, -1.0, 1.0)
    e0 = np.arctan2(r21, r22)
    e1 = np.arcsin(sb)
    e2 = np.arctan2(r10, r00)
    lock = np.abs(sb) > 1.0 - 1e-14
    if np.any(lock):
        r01 = 2.0 * (x * y - w * z)
        r11 = 1.0 - 2.0 * (x * x + z * z)
        e0 = np.where(lock, np.arctan2(np.where(sb > 0, r01, -r01), r11), e0)
        e2 = np.where(lock, 0.0, e2)
    return np.stack([e0, e1, e2], axis=-1)


def quaternion_to_discrete_euler(quaternion, resolution):
    """bins of `resolution` degrees over [0, 360): round-half-even of (angle + 180) / resolution, the top bin is bin 0."""
    deg = np.degrees(quaternion_to_euler_xyz(quaternion)) + 180.0
    disc = np.rint(deg / resolution).astype(int)
    disc[disc == int(360 / resolution)] = 0
    return disc


def discrete_euler_to_quaternion(discrete_euler, resolution):
    """bin indices -> (x, y, z, w) of Rz(c) Ry(b) Rx(a), angles = index * resolution - 180 degrees."""
    half = np.radians(np.asarray(discrete_euler, dtype=np.float64) * resolution - 180.0) * 0.5
    ca, cb, cc = np.cos(half[..., 0]), np.cos(half[..., 1]), np.cos(half[..., 2])
    sa, sb, sc = np.sin(half[..., 0]), np.sin(half[..., 1]), np.sin(half[..., 2])
    # q = qz(c) * qy(b) * qx(a)
    return np.stack([sa * cb * cc - ca * sb * sc,
                     ca * sb * cc + sa * cb * sc,
                     ca * cb * sc - sa * sb * cc,
                     ca * cb * cc + sa * sb * sc], axis=-1)


def point_to_voxel_index(point, voxel_size, coord_bounds):
    """float64 floor((p - min) / (res + 1e-12)) with res = (max - min) / (V + 1e-12); clipped from above only (a point
    below the lower bound yields a negative index, which the augmentation uses as its 'out of bounds' signal)."""
    bounds = np.asarray(coord_bounds)
    lo, hi = bounds[..., 0:3], bounds[..., 3:6]
    res = (hi - lo) / (np.full(3, voxel_size) + 1e-12)
    idx = np.floor((np.asarray(point) - lo) / (res + 1e-12)).astype(np.int32)
    return np.minimum(idx, voxel_size - 1)


def point_to_pixel_index(point, extrinsics, intrinsics):
    """world point -> (px, py) of the pinhole camera (extrinsics = camera-to-world 4x4), with the reference's mirrored
    image convention: p = 2 c - int(-f * (X / Z) + c) per axis."""
    cam = np.linalg.inv(extrinsics).dot(np.array([point[0], point[1], point[2], 1.0]))
    out = []
    for axis in (0, 1):
        f, c = intrinsics[axis, axis], intrinsics[axis, 2]
        out.append(2 * c - int(-f * (cam[axis] / cam[2]) + c))
    return out[0], out[1]
