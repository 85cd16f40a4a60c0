"""GPU parity: HIP voxelizer (through the C ABI) vs golden fixtures and the oracle.  Bit-exact in ALL channels."""
import numpy as np
import pytest
import torch

from oracle import voxel_grid as ovox
from voxactb_amd import synthetic
from voxactb_amd.voxel.voxel_grid import VoxelGrid

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(params=['tiles', 'table'], autouse=True)
def chain(request):
    """every test runs on both point chains: the tile-routed one (default) and the table-based fallback"""
    from voxactb_amd import _lib
    assert _lib.lib().vxb_voxelize_select_chain(1 if request.param == 'table' else 0) == 0
    yield request.param
    _lib.lib().vxb_voxelize_select_chain(0)


def T(a):
    return torch.from_numpy(np.asarray(a))


def same(a, b):
    return torch.equal(torch.nan_to_num(a.cpu()), torch.nan_to_num(b.cpu()))


def run(coords, feats, bounds, V):
    B, N, _ = coords.shape
    vg = VoxelGrid(bounds[0].tolist(), V, DEV, B, 0 if feats is None else feats.shape[-1], N)
    out = vg.coords_to_bounding_voxel_grid(coords.to(DEV), None if feats is None else feats.to(DEV), bounds.to(DEV))
    torch.cuda.synchronize()
    return out, vg


def test_golden_kats(golden):
    g = golden('f1_voxel_kats')
    out, _ = run(T(g['kat_coords']), T(g['kat_feats']), T(g['kat_bounds']), int(g['kat_V']))
    assert same(out, T(g['kat_grid']))
    for c in range(int(g['n_cases'])):
        out, _ = run(T(g['c%d_coords' % c]), T(g['c%d_feats' % c]), T(g['c%d_bounds' % c]), int(g['c%d_V' % c]))
        ref = T(g['c%d_grid' % c])
        assert same(out[..., -1], ref[..., -1]), 'occupancy case %d' % c
        assert same(out, ref), 'case %d' % c


def cams_batch(B, cams, H, W, V, seed):
    rs = synthetic.make_replay_sample(B, cams, (H, W), V, 4, seed=seed)
    pcd = [rs['%s_point_cloud' % c][:, 0] for c in cams]
    rgb = [(rs['%s_rgb' % c][:, 0] / 255.0) * 2.0 - 1.0 for c in cams]
    return pcd, rgb


def test_c1_fixture_and_camera_path(golden):
    g = golden('f3_encoder_c1')
    pcd, rgb = cams_batch(1, ['front'], 64, 64, 32, seed=1)
    bounds = torch.tensor([synthetic.SCENE_BOUNDS])
    vg = VoxelGrid(synthetic.SCENE_BOUNDS, 32, DEV, 1, 3, 64 * 64)
    out = vg.voxelize_cameras([p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb], bounds.to(DEV))
    assert same(out, T(g['grid']))
    out2 = vg.voxelize_cameras([p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb])   # default bounds, 2nd call
    assert same(out2, T(g['grid']))


@pytest.mark.parametrize('B,ncam,HW,V,per_sample', [(2, 2, 16, 8, False), (3, 4, 32, 50, True), (16, 4, 128, 100, False),
                                                     (2, 3, 48, 25, True)])
def test_vs_oracle(B, ncam, HW, V, per_sample):
    cams = synthetic.CAMERAS4[:ncam]
    pcd, rgb = cams_batch(B, cams, HW, HW, V, seed=7)
    if per_sample:
        g = np.random.default_rng(3)
        c = np.array(synthetic.SCENE_BOUNDS[:3]) + g.uniform(0.3, 0.7, (B, 3))
        bounds = torch.tensor(np.concatenate([c - 0.3, c + 0.3], 1), dtype=torch.float32)
    else:
        bounds = torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    ref = ovox.voxelize(coords, feats, bounds, V)
    vg = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, ncam * HW * HW)
    out = vg.voxelize_cameras([p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb], bounds.to(DEV))
    assert same(out[..., -1], ref[..., -1]), 'occupancy'
    assert same(out[..., 6:9], ref[..., 6:9]), 'index channels'
    assert same(out, ref), 'mean channels'
    out_flat, _ = run(coords, feats, bounds, V)
    assert same(out_flat, ref)
    # run-to-run determinism
    out3 = vg.voxelize_cameras([p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb], bounds.to(DEV))
    assert torch.equal(out, out3)


@pytest.mark.parametrize('B,ncam,HW,V', [(2, 4, 144, 50), (1, 4, 160, 32), (1, 4, 384, 40)])
def test_point_counts_beyond_the_headline_size(B, ncam, HW, V):
    """round-5 advisor: the merged heavy | light tile launch serves at most 64 route chunks of 1024 points per sample -- exactly the
    headline's 4 x 128 x 128 -- and the tile-routed chain at most 512 chunks.  4 x 144^2 = 82 944 and 4 x 160^2 = 102 400 points (81 / 100
    chunks: the all-heavy tile chain), 4 x 384^2 = 589 824 >= 2^19 (the table-based chain takes over): same grids as the oracle, bit for bit,
    stateless and through the incremental (persistent-buffer) path."""
    cams = synthetic.CAMERAS4[:ncam]
    pcd, rgb = cams_batch(B, cams, HW, HW, V, seed=11)
    bounds = torch.tensor([synthetic.SCENE_BOUNDS])
    coords, feats = ovox.flatten_cameras(pcd, rgb)
    ref = ovox.voxelize(coords, feats, bounds, V)
    vg = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, ncam * HW * HW)
    out = vg.voxelize_cameras([p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb], bounds.to(DEV))
    assert same(out, ref)
    vgp = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, ncam * HW * HW, persistent=2)
    pcd2, rgb2 = cams_batch(B, cams, HW, HW, V, seed=12)
    ref2 = ovox.voxelize(*ovox.flatten_cameras(pcd2, rgb2), bounds, V)
    for i in range(4):            # both persistent buffers, each updated in place once
        a, b_, r = (pcd, rgb, ref) if i % 2 == 0 else (pcd2, rgb2, ref2)
        o = vgp.voxelize_cameras([p.to(DEV) for p in a], [q.to(DEV) for q in b_], bounds.to(DEV))
        assert same(o, r), i


def test_edge_cases():
    V = 16
    bounds = torch.tensor([[0., 0., 0., 1., 1., 1.]])
    # (a) nothing inside -> all cells empty
    p = torch.full((2, 100, 3), 5.0)
    out, _ = run(p, torch.ones(2, 100, 3), bounds, V)
    assert same(out, ovox.voxelize(p, torch.ones(2, 100, 3), bounds, V)) and float(out[..., -1].sum()) == 0
    # (b) every point in ONE cell (long path, count >> 16), order-sensitive values
    g = np.random.default_rng(0)
    p = torch.from_numpy((0.5 + 0.01 * g.uniform(0, 1, (1, 5000, 3))).astype(np.float32))
    f = torch.from_numpy((g.standard_normal((1, 5000, 3)) * 10 ** g.uniform(-3, 3, (1, 5000, 1))).astype(np.float32))
    out, _ = run(p, f, bounds, V)
    assert same(out, ovox.voxelize(p, f, bounds, V))
    # (c) counts straddling the short/long threshold (15, 16, 17, 18 points per cell) + no features
    pts = []
    for i, cnt in enumerate((1, 2, 3, 15, 16, 17, 18, 33, 64, 65)):
        base = np.array([0.03 + 0.0625 * i, 0.5, 0.5], np.float32)
        pts.append(base + 0.01 * g.uniform(0, 1, (cnt, 3)).astype(np.float32))
    p = torch.from_numpy(np.concatenate(pts)[None])
    perm = torch.from_numpy(g.permutation(p.shape[1]))
    p = p[:, perm]
    f = torch.from_numpy(g.standard_normal((1, p.shape[1], 3)).astype(np.float32))
    out, _ = run(p, f, bounds, V)
    assert same(out, ovox.voxelize(p, f, bounds, V))
    out, _ = run(p, None, bounds, V)
    assert same(out, ovox.voxelize(p, torch.zeros(1, p.shape[1], 0), bounds, V))
    # (d) odd V (scalar fill path) and feature_size 1
    out, _ = run(p, f[..., :1], bounds, 7)
    assert same(out, ovox.voxelize(p, f[..., :1], bounds, 7))
    # (e) a slab too large for LDS whose cells end exactly on / straddle the 4096-pair chunk boundary
    for n0 in (4095, 4096, 4097):
        pa = 0.51 + 0.001 * g.uniform(0, 1, (n0, 3))
        pb = pa[:1500] + np.array([0.0, 0.0625, 0.0])
        pc = pa[:700] + np.array([0.0, 0.0, 0.0625])
        p = torch.from_numpy(np.concatenate([pa, pb, pc])[None].astype(np.float32))
        p = p[:, torch.from_numpy(g.permutation(p.shape[1]))]
        f = torch.from_numpy(g.standard_normal((1, p.shape[1], 3)).astype(np.float32))
        out, _ = run(p, f, bounds, V)
        assert same(out, ovox.voxelize(p, f, bounds, V)), n0


def test_tile_chain_edges():
    """sizes around the tile-routed chain's units: chunks of 2048 points, tiles of 8 x 8 x 16 cells, one heavy cell that
    holds a whole sample, a grid smaller than one tile, point counts that are not a multiple of anything"""
    g = np.random.default_rng(11)
    bounds = torch.tensor([[0., 0., 0., 1., 1., 1.]])
    for V, N in ((4, 1), (6, 63), (8, 2047), (16, 2048), (16, 2049), (24, 6000), (40, 4097)):
        p = torch.from_numpy(g.uniform(-0.05, 1.05, (2, N, 3)).astype(np.float32))
        f = torch.from_numpy(g.standard_normal((2, N, 3)).astype(np.float32))
        out, _ = run(p, f, bounds, V)
        assert same(out, ovox.voxelize(p, f, bounds, V)), (V, N)
    # a whole 4-camera sample (65 536 points) in ONE cell next to a second sample spread over the grid
    V, N = 20, 65536
    p = torch.from_numpy(g.uniform(0.0, 1.0, (2, N, 3)).astype(np.float32))
    p[0] = 0.5125 + 0.02 * p[0]
    f = torch.from_numpy((g.standard_normal((2, N, 3)) * 10 ** g.uniform(-2, 2, (2, N, 1))).astype(np.float32))
    out, _ = run(p, f, bounds, V)
    assert same(out, ovox.voxelize(p, f, bounds, V))
    # features: 0 .. 4 on the tile chain, 5 falls back to the table chain
    p = torch.from_numpy(g.uniform(0.0, 1.0, (1, 3000, 3)).astype(np.float32))
    for F in (0, 1, 2, 4, 5):
        f = torch.from_numpy(g.standard_normal((1, 3000, F)).astype(np.float32))
        out, _ = run(p, f if F else None, bounds, 12)
        assert same(out, ovox.voxelize(p, f, bounds, 12)), F


def test_fused_rigid_transform_equals_transform_then_voxelize():
    """`xform` of vxb_voxelize_f32 (the SE(3) augmentation folded into the point load) against vxb_se3_points_f32 followed
    by a plain voxelization, and against the oracle's perturb_points (reference augmentation.py:36-62)."""
    from oracle import se3 as ose3
    from voxactb_amd.voxel import augmentation as aug
    B, H, W, V = 3, 24, 20, 20
    cams = ['front', 'wrist']
    pcd, rgb = cams_batch(B, cams, H, W, V, seed=5)
    g = np.random.default_rng(2)
    ang = torch.from_numpy(g.uniform(-0.8, 0.8, (B, 3)).astype(np.float32))
    R = ose3.euler_angles_to_matrix(ang, 'XYZ')
    t = torch.from_numpy((np.array(synthetic.SCENE_BOUNDS[:3]) + g.uniform(0.3, 0.7, (B, 3))).astype(np.float32))
    c = t + torch.from_numpy(g.uniform(-0.1, 0.1, (B, 3)).astype(np.float32))
    xf = torch.cat([R.reshape(B, 9), t, c], 1).contiguous().to(DEV)
    bounds = torch.tensor([synthetic.SCENE_BOUNDS])
    vg = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, len(cams) * H * W)
    fused = vg.voxelize_cameras([p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb], bounds.to(DEV), xform=xf)
    moved = aug.transform_point_clouds([p.to(DEV) for p in pcd], xf)
    plain = vg.voxelize_cameras(moved, [r.to(DEV) for r in rgb], bounds.to(DEV))
    assert torch.equal(fused, plain)
    # the kernel's points vs the reference formula (p - t) R + c evaluated by the oracle on the host
    want = ose3.perturb_points(pcd, c - t, R, t, torch.tensor([[-10., -10., -10., 10., 10., 10.]]))
    for w_, m_ in zip(want, moved):
        assert float((m_.cpu() - w_).abs().max()) < 2e-6
    ref = ovox.voxelize(*ovox.flatten_cameras([m.cpu() for m in moved], rgb), bounds, V)
    assert same(fused, ref)


def test_persistent_buffers_are_updated_in_place_to_the_same_grids():
    """VoxelGrid(persistent=2): results of five calls with different clouds, colours and per-sample bounds equal the
    fresh-buffer results (incremental update = reset the cells occupied two calls ago + write the new ones)."""
    B, H, W, V = 3, 32, 32, 40
    cams = ['front', 'wrist']
    fresh = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, len(cams) * H * W)
    pers = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, len(cams) * H * W, persistent=2)
    g = np.random.default_rng(4)
    seen = []
    for it in range(5):
        pcd, rgb = cams_batch(B, cams, H, W, V, seed=20 + it)
        c = np.array(synthetic.SCENE_BOUNDS[:3]) + g.uniform(0.3, 0.7, (B, 3))
        bounds = torch.tensor(np.concatenate([c - 0.35, c + 0.35], 1), dtype=torch.float32).to(DEV)
        pcd, rgb = [p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb]
        want = fresh.voxelize_cameras(pcd, rgb, bounds)
        got = pers.voxelize_cameras(pcd, rgb, bounds)
        assert torch.equal(got, want), it
        seen.append(got.data_ptr())
    assert len(set(seen)) == 2 and seen[0] == seen[2] == seen[4] and seen[1] == seen[3]      # two buffers, used in turn
    # an empty cloud after a full one: every previously occupied cell is reset
    far = [torch.full((B, 3, H, W), 9.0, device=DEV) for _ in cams]
    for _ in range(2):
        got = pers.voxelize_cameras(far, rgb, bounds)
        assert torch.equal(got, fresh.voxelize_cameras(far, rgb, bounds)) and float(got[..., -1].sum()) == 0


def test_persistent_grid_check_catches_a_caller_that_wrote_into_a_returned_grid(monkeypatch, chain):
    """VOXACTB_VOXEL_CHECK: the reused buffers are handed out as views; the debug check compares an incremental update with a full
    rewrite and raises when a holder has modified a grid in an EMPTY cell (which the incremental reset does not touch)."""
    from voxactb_amd.voxel import voxel_grid as vgm
    if chain == 'table':
        pytest.skip('the table-based fallback chain rewrites the whole grid on every call: nothing incremental to check')
    B, H, W, V = 2, 16, 16, 24
    cams = ['front']
    monkeypatch.setattr(vgm, 'CHECK_EVERY', 1)
    pers = VoxelGrid(synthetic.SCENE_BOUNDS, V, DEV, B, 3, H * W, persistent=1)
    pcd, rgb = cams_batch(B, cams, H, W, V, seed=31)
    pcd, rgb = [p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb]
    g0 = pers.voxelize_cameras(pcd, rgb)
    pers.voxelize_cameras(pcd, rgb)                         # incremental, checked, fine
    empty = torch.nonzero(g0[..., -1] == 0)[0]
    g0[empty[0], empty[1], empty[2], empty[3], 0] = 123.0      # a caller scribbles into the view it was given
    with pytest.raises(Exception, match='differs from a full rewrite'):
        pers.voxelize_cameras(pcd, rgb)


def test_overlapped_launch_orders_give_the_same_grid():
    from voxactb_amd import _lib
    pcd, rgb = cams_batch(2, ['front', 'wrist'], 64, 64, 50, seed=8)
    pcd, rgb = [p.to(DEV) for p in pcd], [r.to(DEV) for r in rgb]
    vg = VoxelGrid(synthetic.SCENE_BOUNDS, 50, DEV, 2, 3, 2 * 64 * 64)
    ref = vg.voxelize_cameras(pcd, rgb)
    for order in (3, 5):
        assert _lib.lib().vxb_voxelize_select_chain(order) == 0
        assert torch.equal(vg.voxelize_cameras(pcd, rgb), ref), order


def test_depth_images_give_the_grid_of_the_reference_point_clouds(golden):
    """RGB-D input: voxelize_depth(depth buffers, cameras) == voxelize(point clouds PyRep computed from them) -- fixture F10
    holds clouds from PyRep's own code -- bit for bit, for a 0..1 depth buffer with near / far and for metres."""
    g = golden('f10_depth_clouds')
    B, H, W, nc = int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), int(g['cfg_ncam'])
    near, far = float(g['near']), float(g['far'])
    d01 = [torch.stack([T(g['b%d_c%d_depth01' % (b, c)]) for b in range(B)]).to(DEV) for c in range(nc)]
    clouds = [torch.stack([T(g['b%d_c%d_cloud' % (b, c)]).permute(2, 0, 1) for b in range(B)]).contiguous().to(DEV) for c in range(nc)]
    ext = torch.stack([torch.stack([T(g['b%d_c%d_ext' % (b, c)]) for c in range(nc)]) for b in range(B)])
    K = torch.stack([torch.stack([T(g['b%d_c%d_int' % (b, c)]) for c in range(nc)]) for b in range(B)])
    rgb = [torch.rand(B, 3, H, W, device=DEV) for _ in range(nc)]
    lo = torch.stack([c.amin(dim=(0, 2, 3)) for c in clouds]).amin(0).cpu()
    hi = torch.stack([c.amax(dim=(0, 2, 3)) for c in clouds]).amax(0).cpu()
    bounds = torch.cat([lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo)]).unsqueeze(0).to(DEV)
    V = 24
    vg = VoxelGrid(bounds[0].tolist(), V, DEV, B, 3, nc * H * W)
    want = vg.voxelize_cameras(clouds, rgb, bounds)
    assert float(want[..., -1].sum()) > 50
    got = vg.voxelize_depth(d01, rgb, ext, K, bounds, near_far=(near, far))
    assert torch.equal(got, want)
    metres = [(np.float32(near) + d.cpu().numpy() * np.float32(far - near)).astype(np.float32) for d in d01]
    got_m = vg.voxelize_depth([torch.from_numpy(m).to(DEV) for m in metres], rgb, ext, K, bounds)
    assert torch.equal(got_m, want)


def test_errors():
    from voxactb_amd import _lib
    vg = VoxelGrid([0, 0, 0, 1, 1, 1], 8, DEV, 2, 3, 16)
    with pytest.raises(_lib.VoxactbHipError):
        vg.coords_to_bounding_voxel_grid(torch.zeros(2, 16, 3, device=DEV), torch.zeros(2, 16, 3, device=DEV),
                                         torch.zeros(3, 6, device=DEV))   # bounds rows not in {1, B}
