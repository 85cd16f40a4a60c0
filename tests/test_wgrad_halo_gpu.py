"""LDS-halo 3x3x3 weight-gradient kernel (wgrad_halo.hip): 'bf16' against the generic transposed-read bf16 kernel (same
rounded operands, different fp32 summation order) and 'bf16x3' against a float64 PyTorch reference at the bound of the
exact-fp32 kernels; ragged edges (S not a multiple of the 2x8x8 tile), two sources, zero padding and the
depth-to-space dY gather of the polyphase up-conv."""
import pytest
import torch
import torch.nn.functional as F

from voxactb_amd import ops
from .test_ops_gpu import rnd, close, cl, ref_conv, DEV

pytestmark = pytest.mark.gpu


def _generic(fn):
    ops.HALO_CONV = False
    try:
        return fn()
    finally:
        ops.HALO_CONV = True


@pytest.mark.parametrize('C0,C1,N,S,B', [(64, 64, 64, 19, 2), (16, 0, 128, 17, 1), (32, 0, 64, 16, 3)])
def test_wgrad_halo_bf16_matches_generic(C0, C1, N, S, B):
    a = cl(rnd(B, C0, S, S, S)).to(DEV)
    c = cl(rnd(B, C1, S, S, S, seed=5)).to(DEV) if C1 else None
    dy = cl(rnd(B, N, S, S, S, seed=3)).to(DEV)
    ref = _generic(lambda: ops.conv3d_wgrad(a, dy, N, B, S, S, 3, -1, src1=c, force_bf16=True))
    got = ops.conv3d_wgrad(a, dy, N, B, S, S, 3, -1, src1=c, force_bf16=True)
    close(got, ref, 3e-5, 'halo wgrad bf16 vs generic')
    # explicit split counts, including one that does not divide the tile count
    for ns in (1, 7):
        close(ops.conv3d_wgrad(a, dy, N, B, S, S, 3, -1, src1=c, force_bf16=True, nsplit=ns), ref, 3e-5, 'nsplit %d' % ns)


@pytest.mark.parametrize('C0,C1,N,S', [(64, 64, 64, 18), (32, 0, 64, 21)])
def test_wgrad_halo_x3_matches_fp64(C0, C1, N, S):
    B = 2
    a, c = rnd(B, C0, S, S, S), (rnd(B, C1, S, S, S, seed=5) if C1 else None)
    dy = rnd(B, N, S, S, S, seed=3)
    xin = torch.cat([a, c], 1) if C1 else a
    W = torch.zeros(N, C0 + C1, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    ref_conv(xin.double(), W, None).backward(dy.double())
    got = ops.conv3d_wgrad(cl(a).to(DEV), cl(dy).to(DEV), N, B, S, S, 3, -1, src1=cl(c).to(DEV) if C1 else None,
                           force_bf16='bf16x3')
    close(got, ops.conv_weight_fwd(W.grad.float()), 2e-5, 'halo wgrad x3 vs fp64')


def test_wgrad_halo_zero_padding_dgrad_geometry():
    """zero padding with S_out = S_in + 2, off = -2 (the geometry a data-gradient conv's own weight gradient would use)."""
    B, C, N, S = 2, 32, 64, 16
    a = cl(rnd(B, C, S, S, S)).to(DEV)
    dy = cl(rnd(B, N, S + 2, S + 2, S + 2, seed=3)).to(DEV)
    ref = _generic(lambda: ops.conv3d_wgrad(a, dy, N, B, S, S + 2, 3, -2, replicate=False, force_bf16=True))
    got = ops.conv3d_wgrad(a, dy, N, B, S, S + 2, 3, -2, replicate=False, force_bf16=True)
    close(got, ref, 3e-5, 'halo wgrad zero pad')


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
def test_wgrad_halo_depth_to_space_dy(mode):
    """dY gathered from the fine grid of a depth-to-space output (polyphase up-conv): N = s^3 * 64 columns."""
    B, C, G, s = 1, 32, 16, 2
    z = cl(rnd(B, C, G, G, G)).to(DEV)
    dyf = cl(rnd(B, 64, G * s, G * s, G * s, seed=3)).to(DEV)
    N = s ** 3 * 64
    ref = _generic(lambda: ops.conv3d_wgrad(z, dyf, N, B, G, G, 3, -1, d2s=(s, 64), force_bf16=mode))
    got = ops.conv3d_wgrad(z, dyf, N, B, G, G, 3, -1, d2s=(s, 64), force_bf16=mode)
    close(got, ref, 3e-5, 'halo wgrad d2s ' + mode)


# ------------------------------------------------------------------------------------------------ fp16 single-product mode
def _f16(fn):
    ops.WGRAD_PRECISION = 'fp16'
    try:
        return fn()
    finally:
        ops.WGRAD_PRECISION = ''


@pytest.mark.parametrize('C0,C1,N,S,gain', [(64, 64, 64, 18, 1.0), (32, 0, 64, 21, 3e-9), (16, 0, 128, 17, 4e6)])
def test_wgrad_halo_fp16_matches_fp64(C0, C1, N, S, gain):
    """one fp16 product per term with the gradient operand scaled on the device: against float64 at the 2^-12 operand
    rounding averaged over the B S^3 voxels of the reduction, for gradients of ordinary, tiny (3e-9: far below fp16's
    smallest normal 6e-5) and huge (4e6: far above its largest 65504) magnitude -- the scale comes from the tensor itself."""
    B = 2
    a, c = rnd(B, C0, S, S, S), (rnd(B, C1, S, S, S, seed=5) if C1 else None)
    dy = rnd(B, N, S, S, S, seed=3) * gain
    dy[0, :, 0, 0, 0] *= 50.0                                     # a few spikes 50x above the bulk, like SpatialSoftmax3D's gradient
    xin = torch.cat([a, c], 1) if C1 else a
    W = torch.zeros(N, C0 + C1, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    ref_conv(xin.double(), W, None).backward(dy.double())
    ref = ops.conv_weight_fwd(W.grad.float())
    got = _f16(lambda: ops.conv3d_wgrad(cl(a).to(DEV), cl(dy).to(DEV), N, B, S, S, 3, -1, src1=cl(c).to(DEV) if C1 else None,
                                        force_bf16='bf16x3'))
    err = float((got.cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err < 4e-4, err                                        # (bf16x3: 2e-5; plain bf16 would be ~3e-3)
    x3 = ops.conv3d_wgrad(cl(a).to(DEV), cl(dy).to(DEV), N, B, S, S, 3, -1, src1=cl(c).to(DEV) if C1 else None, force_bf16='bf16x3')
    assert float((x3.cpu() - ref).abs().max()) / float(ref.abs().max()) < 3e-5


def test_wgrad_halo_fp16_depth_to_space_tap_masks_and_saturation():
    """the polyphase up-conv's gradient (fine-grid dY, per-phase tap masks) in fp16; activations beyond fp16's range saturate
    at 65504 instead of turning the gradient into inf / NaN."""
    B, C, G, s, k = 1, 64, 20, 5, 5
    z = cl(rnd(B, C, G, G, G)).to(DEV)
    dyf = cl(rnd(B, 64, G * s, G * s, G * s, seed=3) * 1e-6).to(DEV)
    N = s ** 3 * 64
    st = ops.polyphase_structure(k, s, DEV)
    kw = dict(d2s=(s, 64), force_bf16='bf16x3', phase_mask=st['phase_mask_t'], flops_frac=st['frac'])
    ref = ops.conv3d_wgrad(z, dyf, N, B, G, G, 3, -1, **kw)
    got = _f16(lambda: ops.conv3d_wgrad(z, dyf, N, B, G, G, 3, -1, **kw))
    assert float((got - ref).abs().max()) / float(ref.abs().max()) < 1e-3
    assert torch.equal(got == 0, ref == 0)                        # the structurally zero (tap, phase) blocks stay exactly zero
    z[0, 3, 3, 3, :] = 1e9
    big = _f16(lambda: ops.conv3d_wgrad(z, dyf, N, B, G, G, 3, -1, **kw))
    assert bool(torch.isfinite(big).all())


@pytest.mark.parametrize('C0,C1,N,S,gain', [(128, 0, 64, 20, 1.0), (32, 32, 64, 17, 3e-9), (16, 0, 128, 16, 4e6)])
def test_wgrad_k5_as_shifted_blocks_fp16_matches_fp64(C0, C1, N, S, gain):
    """5x5x5 stride-1 weight gradient as eight shifted 3x3x3 blocks of the LDS-halo kernel (vxb_conv3_wgrad_halo5_f16_f32):
    every one of the 125 taps against F.conv3d's float64 autograd on the replicate-padded input (pad 2), ragged grids, two
    sources, tiny / huge gradients; and against the generic gather kernel (bf16x3) it replaces."""
    B = 2
    a, c = rnd(B, C0, S, S, S), (rnd(B, C1, S, S, S, seed=5) if C1 else None)
    dy = rnd(B, N, S, S, S, seed=3) * gain
    dy[0, :, 0, 0, 0] *= 50.0
    xin = torch.cat([a, c], 1) if C1 else a
    W = torch.zeros(N, C0 + C1, 5, 5, 5, dtype=torch.float64, requires_grad=True)
    F.conv3d(F.pad(xin.double(), (2,) * 6, mode='replicate'), W).backward(dy.double())
    ref = ops.conv_weight_fwd(W.grad.float())
    args = (cl(a).to(DEV), cl(dy).to(DEV), N, B, S, S, 5, -2)
    kw = dict(src1=cl(c).to(DEV) if C1 else None, force_bf16='bf16x3')
    got = _f16(lambda: ops.conv3d_wgrad(*args, **kw))
    assert got.shape == ref.shape
    err = float((got.cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err < 4e-4, err
    gen = ops.conv3d_wgrad(*args, **kw)                      # (WGRAD_PRECISION is not 'fp16' here: the generic bf16x3 kernel)
    assert float((gen.cpu() - ref).abs().max()) / float(ref.abs().max()) < 3e-5
    assert float((got - gen).abs().max()) / float(ref.abs().max()) < 4e-4


def test_absmax_scale():
    for n, val in ((5, 3.0), (1 << 20, 1e-7), (12345, 7e5), (64, 0.0)):
        x = torch.rand(n, device=DEV) * val
        if n > 100:
            x[n // 3] = -val * 1.5                                # the maximum is a negative element
        sc = ops.absmax_scale(x).cpu()
        m = float(x.abs().max())
        if m == 0:
            assert sc.tolist() == [1.0, 1.0]
            continue
        assert 2.0 ** 14 <= m * float(sc[0]) < 2.0 ** 15 and float(sc[0]) * float(sc[1]) == 1.0
        assert float(torch.log2(sc[0])) == round(float(torch.log2(sc[0])))
    x = torch.ones(100, device=DEV)
    x[7] = float('nan')
    # non-finite input: BOTH factors NaN -- every kernel that multiplies by them then hands on NaN instead of the saturated (finite)
    # values its fp16 conversion would make of the tensor (round 5; tests/test_nan_backward_gpu.py)
    for bad in (float('nan'), float('inf'), -float('inf')):
        x[7] = bad
        assert bool(torch.isnan(ops.absmax_scale(x)).all())
