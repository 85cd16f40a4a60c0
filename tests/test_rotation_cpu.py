"""CPU: the closed-form label helpers (voxactb_amd/helpers/rotation.py) against scipy's Rotation, which is what the
reference calls (peract/helpers/utils.py:92-116), on random and on bin-centre / gimbal-lock inputs."""
import warnings

import numpy as np
from scipy.spatial.transform import Rotation

from voxactb_amd.helpers import rotation as R


def _ref_disc(q, res):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        e = Rotation.from_quat(q).as_euler('xyz', degrees=True) + 180
    d = np.around(e / res).astype(int)
    d[d == int(360 / res)] = 0
    return d


def test_quaternion_to_discrete_euler_matches_scipy():
    g = np.random.default_rng(0)
    q = g.standard_normal((20000, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[q[:, 3] < 0] *= -1
    for res in (5, 3, 15):
        assert np.array_equal(R.quaternion_to_discrete_euler(q, res), _ref_disc(q, res))
    assert np.array_equal(R.quaternion_to_discrete_euler(q[0], 5), _ref_disc(q[0], 5))       # single quaternion


def test_discrete_euler_roundtrip_and_quaternion_signs():
    g = np.random.default_rng(1)
    disc = g.integers(0, 72, (5000, 3))
    disc[:50, 1] = 54            # pitch = +90 degrees: gimbal lock
    disc[50:100, 1] = 18         # pitch = -90 degrees
    mine = R.discrete_euler_to_quaternion(disc, 5)
    ref = Rotation.from_euler('xyz', disc * 5 - 180, degrees=True).as_quat()
    assert np.abs(mine - ref).max() < 1e-15                    # same sign convention as scipy, not only the same rotation
    assert np.array_equal(R.quaternion_to_discrete_euler(ref, 5), _ref_disc(ref, 5))


def test_point_to_voxel_index_and_pixel_index():
    g = np.random.default_rng(2)
    b = np.array([-0.3, -0.5, 0.6, 0.7, 0.5, 1.6], np.float32)
    for _ in range(200):
        p = (b[:3] + g.uniform(-0.1, 1.1, 3) * (b[3:] - b[:3])).astype(np.float32)
        res = (b[3:] - b[:3]) / (np.array([100] * 3) + 1e-12)
        want = np.minimum(np.floor((p - b[:3]) / (res + 1e-12)).astype(np.int32), np.array([100] * 3) - 1)
        assert np.array_equal(R.point_to_voxel_index(p, 100, b), want)
    ext = np.eye(4)
    ext[:3, :3] = Rotation.from_euler('xyz', [0.3, -0.2, 0.5]).as_matrix()
    ext[:3, 3] = [0.1, -0.4, 1.3]
    K = np.array([[-110.85, 0, 64.0], [0, -110.85, 64.0], [0, 0, 1.0]])
    pt = np.array([0.25, 0.1, 0.9])
    cam = np.linalg.inv(ext).dot(np.append(pt, 1.0))
    px = 2 * K[0, 2] - int(-K[0, 0] * (cam[0] / cam[2]) + K[0, 2])
    py = 2 * K[1, 2] - int(-K[1, 1] * (cam[1] / cam[2]) + K[1, 2])
    assert R.point_to_pixel_index(pt, ext, K) == (px, py)
