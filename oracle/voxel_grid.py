"""Oracle: VoxelGrid.coords_to_bounding_voxel_grid (TEST INFRASTRUCTURE).

Restates /root/reference/peract/voxel/voxel_grid.py:148-198 (+ :106-146 for
the scatter-mean) with plain PyTorch CPU ops.  Sums are accumulated
sequentially in point-id order, which is what `scatter_add_` does on CPU
(voxel_grid.py:108) -- the HIP voxelizer reproduces that order, so all ten
channels are compared bit-for-bit.
"""
import torch

MIN_DENOMINATOR = 1e-12  # voxel_grid.py:11


def flatten_cameras(pcd_list, rgb_list):
    """qattention_peract_bc_agent.py:85-93: [B,3,H,W] per camera -> [B,N,3] with
    point id n = cam*H*W + h*W + w."""
    b = pcd_list[0].shape[0]
    coords = torch.cat([p.permute(0, 2, 3, 1).reshape(b, -1, 3) for p in pcd_list], 1)
    feats = torch.cat([p.permute(0, 2, 3, 1).reshape(b, -1, p.shape[1]) for p in rgb_list], 1)
    return coords, feats


def voxel_indices(coords: torch.Tensor, bounds: torch.Tensor, V: int) -> torch.Tensor:
    """voxel_grid.py:153-163.  coords [B,N,3] f32, bounds [1|B,6] f32 ->
    int32 [B,N,3] cell index in the (V+2)^3 grid, clamped to [0, V+1]."""
    bounds = bounds.float()
    bb_mins = bounds[..., 0:3]
    bb_maxs = bounds[..., 3:6]
    bb_ranges = bb_maxs - bb_mins
    dims_orig = torch.full((1, 3), V, dtype=torch.int32)
    res = bb_ranges / (dims_orig.float() + MIN_DENOMINATOR)          # :157
    den = res + MIN_DENOMINATOR                                      # :158
    shifted = bb_mins - res                                          # :160
    fl = torch.floor((coords - shifted.unsqueeze(1)) / den.unsqueeze(1)).int()  # :161-162
    idx = torch.min(fl, torch.full((1, 3), V + 1, dtype=torch.int32))           # :163
    idx = torch.max(idx, torch.zeros((1, 3), dtype=torch.int32))               # :164
    return idx


def voxelize(coords: torch.Tensor, feats: torch.Tensor, bounds: torch.Tensor, V: int) -> torch.Tensor:
    """coords [B,N,3], feats [B,N,F], bounds [1|B,6] -> [B,V,V,V,3+F+3+1] (NDHWC).
    Channel order (voxel_grid.py:196-198): mean xyz, mean feats, idx/V (3), occupancy."""
    coords = coords.float()
    B, N, _ = coords.shape
    W = V + 2
    idx = voxel_indices(coords, bounds, V).long()
    vals = torch.cat([coords, feats.float(), torch.ones(B, N, 1)], -1)   # :166-177
    C = vals.shape[-1]
    flat = ((torch.arange(B).view(B, 1) * W + idx[..., 0]) * W + idx[..., 1]) * W + idx[..., 2]
    flat = flat.reshape(-1)
    sums = torch.zeros(B * W * W * W, C)
    sums.index_add_(0, flat, vals.reshape(-1, C))            # sequential in point order (:108)
    cnt = torch.zeros(B * W * W * W)
    cnt.index_add_(0, flat, torch.ones(B * N))               # :119-120
    cnt.clamp_(min=1)                                        # :121
    mean = sums / cnt.unsqueeze(-1)                          # :123-124
    vox = mean.view(B, W, W, W, C)[:, 1:-1, 1:-1, 1:-1]      # :184
    occ = (vox[..., -1:] > 0).float()                        # :192
    ar = torch.arange(V, dtype=torch.float32)
    ig = torch.stack(torch.meshgrid(ar, ar, ar, indexing='ij'), -1)   # index grid (:86-93)
    ig = (ig / float(V)).unsqueeze(0).expand(B, V, V, V, 3)          # :197
    return torch.cat([vox[..., :-1], ig, occ], -1).contiguous()


def occupied_cells(grid: torch.Tensor) -> torch.Tensor:
    """[B,V,V,V,C] -> int64 [n,4] (b,x,y,z) of occupied cells, lexicographic."""
    return torch.nonzero(grid[..., -1] > 0)


def depth_to_point_cloud(depth, extrinsics, intrinsics, near=None, far=None):
    """RGB-D input path (PyRep/pyrep/objects/vision_sensor.py:155-175, :381-412; RLBench/rlbench/utils.py:205-207).
    depth [H,W] float32 (metres, or a 0..1 buffer when near / far are given), extrinsics [4,4] camera-to-world, intrinsics
    [3,3]  ->  world points [H,W,3] float32, exactly as the stored `<cam>_point_cloud` observations are produced:
    pixel grid (x = column, y = row) times depth in float32, then the inverse projection in float64."""
    import numpy as np
    depth = np.asarray(depth, dtype=np.float32)
    if near is not None:
        depth = (np.float32(near) + depth * np.float32(far - near)).astype(np.float32)          # utils.py:205-207 (fp32 array math)
    H, W = depth.shape
    xs = np.tile(np.arange(W, dtype=np.float32), (H, 1))
    ys = np.tile(np.arange(H, dtype=np.float32)[:, None], (1, W))
    pc = np.stack([xs * depth, ys * depth, depth], -1)                                            # :163-164 (float32)
    extrinsics = np.asarray(extrinsics, dtype=np.float64)
    R_inv = extrinsics[:3, :3].T                                                                  # :165-167
    R_inv_C = R_inv @ extrinsics[:3, 3:4]
    proj = np.asarray(intrinsics, dtype=np.float64) @ np.concatenate((R_inv, -R_inv_C), -1)        # :168-169
    inv = np.linalg.inv(np.concatenate([proj, [[0, 0, 0, 1]]]))[0:3]                                # :170-172
    homo = np.concatenate([pc.astype(np.float64), np.ones((H, W, 1))], -1).reshape(H * W, 4).T      # :406-409
    return (inv @ homo).T.reshape(H, W, 3).astype(np.float32), inv
