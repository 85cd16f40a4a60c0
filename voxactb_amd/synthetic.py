"""Seeded synthetic replay batches with the schema `launch_utils.create_replay`
declares (reference: peract/agents/peract_bc/launch_utils.py:56-145; shapes after
`offline_train_runner.py:140`: every element is (B, 1, *shape)).

There is no RLBench data on the GPU box, so bench.py / tests / smoke() draw their
inputs from here: per camera an RGB image uniform in
[0, 255] and a point cloud that mimics RGB-D geometry -- 70 % of the pixels on
three planes and two boxes inside the scene bounds, 20 % clustered inside a 5 cm
ball (wrist-camera contention on a few voxels), 10 % outside the bounds (dropped
by the voxelizer's border crop).  numpy Philox keyed by (seed, name) so that the
same batch can be regenerated anywhere.
"""
import zlib
import numpy as np
import torch

SCENE_BOUNDS = [-0.3, -0.5, 0.6, 0.7, 0.5, 1.6]   # peract/conf/config.yaml:15
CAMERAS4 = ['front', 'left_shoulder', 'right_shoulder', 'wrist']


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=(zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))


def synthetic_point_cloud(g, B, H, W, bounds):
    """[B,3,H,W] float32 world-metre point cloud."""
    lo = np.asarray(bounds[:3], np.float64)
    hi = np.asarray(bounds[3:], np.float64)
    ext = hi - lo
    n = H * W
    pts = np.empty((B, n, 3), np.float64)
    for b in range(B):
        kind = g.uniform(0, 1, n)
        u = g.uniform(0, 1, (n, 3))
        p = lo + u * ext
        # three planes (table z, back wall x, side wall y)
        m = kind < 0.30
        p[m, 2] = lo[2] + 0.15 * ext[2] + 0.002 * g.standard_normal(m.sum())
        m = (kind >= 0.30) & (kind < 0.42)
        p[m, 0] = hi[0] - 0.05 * ext[0] + 0.002 * g.standard_normal(m.sum())
        m = (kind >= 0.42) & (kind < 0.52)
        p[m, 1] = lo[1] + 0.08 * ext[1] + 0.002 * g.standard_normal(m.sum())
        # two boxes
        m = (kind >= 0.52) & (kind < 0.62)
        p[m] = lo + (0.3 + 0.12 * u[m]) * ext
        m = (kind >= 0.62) & (kind < 0.70)
        p[m] = lo + (0.6 + 0.08 * u[m]) * ext
        # cluster within a 5 cm ball
        m = (kind >= 0.70) & (kind < 0.90)
        c = lo + g.uniform(0.25, 0.75, 3) * ext
        d = g.standard_normal((m.sum(), 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-9
        p[m] = c + d * (0.05 * g.uniform(0, 1, (m.sum(), 1)) ** (1 / 3))
        # outside the bounds
        m = kind >= 0.90
        side = g.integers(0, 2, (m.sum(), 3))
        off = g.uniform(0.01, 0.5, (m.sum(), 3)) * ext
        p[m] = np.where(side == 0, lo - off, hi + off)
        pts[b] = p
    return np.ascontiguousarray(pts.astype(np.float32).transpose(0, 2, 1).reshape(B, 3, H, W))


def make_replay_sample(batch_size=16, cameras=CAMERAS4, image_size=(128, 128), voxel_size=100,
                       low_dim_size=4, seed=0, scene_bounds=SCENE_BOUNDS, arm_pred_loss=False,
                       crop_target_obj_voxel=False, crop_radius=0.3, n_depths=1, keyframes_near_target=False):
    """dict[str -> torch tensor (B,1,...)] as handed to PreprocessAgent.update()."""
    B = batch_size
    H, W = image_size
    out = {}
    lo = np.asarray(scene_bounds[:3], np.float32)
    hi = np.asarray(scene_bounds[3:], np.float32)
    for cam in cameras:
        g = _rng(seed, cam)
        rgb = g.integers(0, 256, (B, 3, H, W)).astype(np.float32)
        pcd = synthetic_point_cloud(g, B, H, W, scene_bounds)
        out['%s_rgb' % cam] = torch.from_numpy(rgb).unsqueeze(1)
        out['%s_point_cloud' % cam] = torch.from_numpy(pcd).unsqueeze(1)
        out['%s_camera_extrinsics' % cam] = torch.eye(4).repeat(B, 1, 1, 1)
        out['%s_camera_intrinsics' % cam] = torch.eye(3).repeat(B, 1, 1, 1)
    g = _rng(seed, 'labels')
    out['trans_action_indicies'] = torch.from_numpy(
        g.integers(0, voxel_size, (B, 1, 3 * n_depths)).astype(np.int32))
    rg = np.concatenate([g.integers(0, 72, (B, 1, 3)), g.integers(0, 2, (B, 1, 1))], -1)
    out['rot_grip_action_indicies'] = torch.from_numpy(rg.astype(np.int32))
    out['ignore_collisions'] = torch.from_numpy(g.integers(0, 2, (B, 1, 1)).astype(np.int32))
    pos = lo + g.uniform(0.3, 0.7, (B, 3)).astype(np.float32) * (hi - lo)
    q = g.standard_normal((B, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    out['gripper_pose'] = torch.from_numpy(np.concatenate([pos, q], 1)).unsqueeze(1)
    out['lang_goal_emb'] = torch.from_numpy(g.standard_normal((B, 1, 1024)).astype(np.float32))
    out['lang_token_embs'] = torch.from_numpy(g.standard_normal((B, 1, 77, 512)).astype(np.float32))
    out['low_dim_state'] = torch.from_numpy(g.uniform(0, 1, (B, 1, low_dim_size)).astype(np.float32))
    if arm_pred_loss:
        out['label'] = torch.from_numpy(g.integers(0, 2, (B, 1, 1)).astype(np.int32))
    if crop_target_obj_voxel:
        c = lo + g.uniform(0.35, 0.65, (B, 3)).astype(np.float32) * (hi - lo)
        if keyframes_near_target:
            # one task: the target object sits at about the same place in every sample and the keyframe gripper poses are
            # within 10 cm of it -- what the SE(3) augmentation needs to succeed under crop bounds (it discretises EVERY
            # sample with the first sample's bounds, reference augmentation.py:161-162)
            c = (c[:1] + g.uniform(-0.05, 0.05, (B, 3))).astype(np.float32)
            pos = (c + g.uniform(-0.1, 0.1, (B, 3))).astype(np.float32)
            out['gripper_pose'] = torch.from_numpy(np.concatenate([pos, q], 1)).unsqueeze(1)
        tb = np.concatenate([c - crop_radius, c + crop_radius], 1).astype(np.float32)
        out['target_object_scene_bounds'] = torch.from_numpy(tb).unsqueeze(1)
    out['action'] = torch.zeros(B, 1, 8)
    out['reward'] = torch.zeros(B, 1)
    out['terminal'] = torch.zeros(B, 1, dtype=torch.int8)
    out['timeout'] = torch.zeros(B, 1, dtype=torch.bool)
    out['indices'] = torch.arange(B, dtype=torch.int32).unsqueeze(1)
    out['demo'] = torch.ones(B, dtype=torch.bool)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Name-hashed parameter generator: fixtures cannot ship 133 MB of weights, so the fixture generator (which loads them into
# the REFERENCE module), the tests and bench.py's reference-digest check derive every tensor from (parameter name, seed)
# with numpy's Philox counter RNG.  Magnitudes follow the reference's init families so activations stay O(1)
# (perceiver_lang_io.py:197-205,234; network_utils.py:140-154,263-276): conv / dense with lrelu: kaiming-uniform bound
# sqrt(6 / ((1 + a^2) fan_in)); plain nn.Linear: 1 / sqrt(fan_in); LayerNorm: weight 1 + 0.1 u, bias 0.1 u (perturbed on
# purpose so the affine paths are exercised); biases: small non-zero values; latents / pos_encoding: N(0, 1).
# ----------------------------------------------------------------------------------------------------------------------
LRELU_SLOPE = 0.02  # network_utils.py:12


def _hrng(name: str, seed: int):
    key = zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))

def hashed_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    g = _hrng(name, seed)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = name.split('.')[-1]
    if name in ('latents', 'pos_encoding'):
        a = g.standard_normal(n)
    elif '.norm' in name or name.startswith('norm'):
        u = g.uniform(-1.0, 1.0, n)
        a = (1.0 + 0.1 * u) if leaf == 'weight' else 0.1 * u
    elif leaf == 'weight':
        fan_in = int(np.prod(shape[1:]))
        if 'conv3d' in name or name.endswith('linear.weight'):
            bound = np.sqrt(6.0 / ((1.0 + LRELU_SLOPE ** 2) * fan_in))
        else:
            bound = 1.0 / np.sqrt(fan_in)
        a = g.uniform(-bound, bound, n)
    elif leaf == 'bias':
        a = 0.05 * g.uniform(-1.0, 1.0, n)
    else:
        a = g.uniform(-1.0, 1.0, n)
    return torch.from_numpy(a.astype(np.float32).reshape(shape))


def hashed_state_dict(shapes: dict, seed: int = 0) -> dict:
    """shapes: {param_name: shape}.  Returns {param_name: fp32 tensor}."""
    return {k: hashed_tensor(k, v, seed) for k, v in shapes.items()}


def hashed_clip_text_state_dict(width=512, layers=12, vocab=49408, context=77, embed_dim=1024, seed: int = 0) -> dict:
    """Name-hashed weights for the TEXT half of a CLIP checkpoint (state-dict keys of helpers/clip/core/clip.py; magnitudes
    like its initialize_parameters, :366-393): the RN50 checkpoint itself is not available offline, so parity of the HIP text
    encoder is pinned on the reference network run with these weights."""
    shapes = {'token_embedding.weight': (vocab, width), 'positional_embedding': (context, width),
              'ln_final.weight': (width,), 'ln_final.bias': (width,), 'text_projection': (width, embed_dim)}
    for i in range(layers):
        pre = 'transformer.resblocks.%d.' % i
        shapes.update({pre + 'ln_1.weight': (width,), pre + 'ln_1.bias': (width,), pre + 'attn.in_proj_weight': (3 * width, width),
                       pre + 'attn.in_proj_bias': (3 * width,), pre + 'attn.out_proj.weight': (width, width),
                       pre + 'attn.out_proj.bias': (width,), pre + 'ln_2.weight': (width,), pre + 'ln_2.bias': (width,),
                       pre + 'mlp.c_fc.weight': (4 * width, width), pre + 'mlp.c_fc.bias': (4 * width,),
                       pre + 'mlp.c_proj.weight': (width, 4 * width), pre + 'mlp.c_proj.bias': (width,)})
    out = {}
    for name, shape in shapes.items():
        g = _hrng('clip:' + name, seed)
        n = int(np.prod(shape))
        leaf = name.split('.')[-1]
        if '.ln_' in name or name.startswith('ln_'):
            u = g.uniform(-1.0, 1.0, n)
            a = (1.0 + 0.1 * u) if leaf == 'weight' else 0.1 * u
        elif name == 'token_embedding.weight':
            a = 0.02 * g.standard_normal(n)
        elif name == 'positional_embedding':
            a = 0.01 * g.standard_normal(n)
        elif name == 'text_projection':
            a = width ** -0.5 * g.standard_normal(n)
        elif leaf in ('weight', 'in_proj_weight'):
            a = g.uniform(-1.0, 1.0, n) / np.sqrt(shape[1])
        else:
            a = 0.05 * g.uniform(-1.0, 1.0, n)
        out[name] = torch.from_numpy(a.astype(np.float32).reshape(shape))
    return out


def hashed_uniform(name: str, shape, lo=0.0, hi=1.0, seed: int = 0) -> torch.Tensor:
    g = _hrng('u:' + name, seed)
    a = g.uniform(lo, hi, int(np.prod(shape)))
    return torch.from_numpy(a.astype(np.float32).reshape(tuple(shape)))


def hashed_normal(name: str, shape, seed: int = 0) -> torch.Tensor:
    g = _hrng('n:' + name, seed)
    a = g.standard_normal(int(np.prod(shape)))
    return torch.from_numpy(a.astype(np.float32).reshape(tuple(shape)))


def hashed_int(name: str, shape, lo: int, hi: int, seed: int = 0) -> torch.Tensor:
    """integers in [lo, hi)."""
    g = _hrng('i:' + name, seed)
    a = g.integers(lo, hi, int(np.prod(shape)))
    return torch.from_numpy(a.astype(np.int64).reshape(tuple(shape)))


def projection_signs(name: str, numel: int, k: int, device='cpu') -> torch.Tensor:
    """+-1 vector number k for the tensor called `name`: a multiplicative hash of the element index (pure integer torch ops, so the
    fixture generator on the CPU and the GPU test produce the same signs)."""
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) & 0x7FFFFFFF
    i = torch.arange(numel, dtype=torch.int64, device=device)
    x = (i * 0x9E3779B1 + (h + 1) * 0x85EBCA77 + (k + 1) * 0xC2B2AE3D) & 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & 0xFFFFFFFF
    x = ((x ^ (x >> 15)) * 0x846CA68B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return ((x >> 7) & 1).to(torch.float64) * 2.0 - 1.0


def project(t: torch.Tensor, name: str, nproj: int) -> torch.Tensor:
    """nproj random +-1 projections of a tensor, in float64 (on the tensor's device)."""
    f = t.detach().reshape(-1).double()
    return torch.stack([(f * projection_signs(name, f.numel(), k, f.device)).sum() for k in range(nproj)]).cpu()


def projection_error(pa: torch.Tensor, pb: torch.Tensor) -> torch.Tensor:
    """estimate of ||a - b|| from the projections of a and b: E[(s . (a - b))^2] = ||a - b||^2 for independent +-1 signs."""
    return ((pa.double() - pb.double()) ** 2).mean().sqrt()
