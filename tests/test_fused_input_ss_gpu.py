"""The input conv fused with the pooled features of its output (vxb_pointwise_ss3d_fwd_f32, vxb_pointwise_wgrad_ss3d_f32):
forward bit-identical to the two-kernel path, gradients against torch autograd of the reference formulation
(perceiver_lang_io.py:357 + :360; network_utils.py:773-809)."""
import pytest
import torch
import torch.nn.functional as F

from voxactb_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def ref_ss3d(x):
    """oracle restatement of network_utils.py:773-809 (SpatialSoftmax3D) + AdaptiveMaxPool3d(1) on x [B,C,D,H,W]."""
    from oracle import perceiver as operc
    return operc.spatial_softmax3d(x), x.amax(dim=(2, 3, 4))


@pytest.mark.parametrize('B,S', [(2, 12), (3, 13), (1, 20)])
def test_fused_forward_is_bit_identical(B, S):
    x = (rnd(B, S, S, S, 10, seed=S) * 0.5).to(DEV)
    W = (rnd(64, 10, seed=1) * 0.1).to(DEV)
    b = (rnd(64, seed=2) * 0.05).to(DEV)
    y0 = ops.pointwise_fwd(x, W, b)
    ss0 = ops.ss3d_max_fwd(y0, S ** 3 * 64, B, S, 64)
    y1, ss1 = ops.pointwise_ss3d_fwd(x, W, b, B, S)
    assert torch.equal(y0, y1)
    for a, c, name in zip(ss0, ss1, ('out_ss', 'out_max', 'stats', 'argmax')):
        assert torch.equal(a, c), name


@pytest.mark.parametrize('B,S', [(2, 12), (2, 17)])
def test_fused_weight_gradient(B, S):
    x = (rnd(B, S, S, S, 10, seed=3) * 0.5)
    W = (rnd(64, 10, seed=1) * 0.1).requires_grad_(True)
    b = (rnd(64, seed=2) * 0.05).requires_grad_(True)
    dy = rnd(B, S, S, S, 64, seed=4) * 0.01
    g_ss, g_mx = rnd(B, 192, seed=5), rnd(B, 64, seed=6)
    # reference formulation in float64 autograd
    y = F.leaky_relu(x.double() @ W.double().t() + b.double(), 0.02)
    ss, mx = ref_ss3d(y.permute(0, 4, 1, 2, 3))
    ((y * dy.double()).sum() + (ss * g_ss.double()).sum() + (mx * g_mx.double()).sum()).backward()
    xd, Wd, bd, dyd = x.to(DEV), W.detach().to(DEV), b.detach().to(DEV), dy.to(DEV)
    yd, (o_ss, o_mx, stats, arg) = ops.pointwise_ss3d_fwd(xd, Wd, bd, B, S)
    dW, db = torch.zeros(64, 10, device=DEV), torch.zeros(64, device=DEV)
    ops.pointwise_wgrad_ss3d(xd, yd, dyd, dW, db, B, S, stats, o_ss, arg, g_ss.to(DEV), g_mx.to(DEV))
    # and the unfused pair the engine used before
    dtot = dyd.clone()
    ops.ss3d_max_bwd(yd, S ** 3 * 64, B, S, 64, stats, o_ss, arg, g_ss.to(DEV), g_mx.to(DEV), dtot, S ** 3 * 64, accumulate=True)
    dW2, db2 = torch.zeros(64, 10, device=DEV), torch.zeros(64, device=DEV)
    ops.pointwise_wgrad(xd, yd, dtot, dW2, db2)
    for mine, pair, ref, name in ((dW, dW2, W.grad, 'dW'), (db, db2, b.grad, 'db')):
        scale = float(ref.abs().max())
        e_ref = float((mine.cpu().double() - ref.double()).abs().max()) / scale
        e_pair = float((mine - pair).abs().max()) / scale
        print(name, 'vs float64 autograd %.2e, vs the unfused kernels %.2e' % (e_ref, e_pair))
        assert e_ref < 2e-4 and e_pair < 2e-5, (name, e_ref, e_pair)


def test_fused_weight_gradient_with_padding_adjoint():
    """fold_src: the padding adjoint of a padded-domain data gradient gathered on the fly == fold_pad + the fused kernel."""
    B, S, pad = 2, 12, 2
    Sp = S + 2 * pad + 1                     # (the strided patchify gradient's buffer is wider than the valid extent)
    x = (rnd(B, S, S, S, 10, seed=3) * 0.5).to(DEV)
    W, b = (rnd(64, 10, seed=1) * 0.1).to(DEV), (rnd(64, seed=2) * 0.05).to(DEV)
    dy = (rnd(B, S, S, S, 64, seed=4) * 0.01).to(DEV)
    src = (rnd(B, Sp, Sp, Sp, 64, seed=8) * 0.01).to(DEV)
    g_ss, g_mx = rnd(B, 192, seed=5).to(DEV), rnd(B, 64, seed=6).to(DEV)
    y, (o_ss, o_mx, stats, arg) = ops.pointwise_ss3d_fwd(x, W, b, B, S)
    dW, db = torch.zeros(64, 10, device=DEV), torch.zeros(64, device=DEV)
    ops.pointwise_wgrad_ss3d(x, y, dy, dW, db, B, S, stats, o_ss, arg, g_ss, g_mx, fold_src=src, Sp=Sp, pad=pad)
    dtot = dy.clone()
    ops.fold_pad(src, Sp, 64, 0, dtot, B, S, 64, pad, accumulate=True)
    dW2, db2 = torch.zeros(64, 10, device=DEV), torch.zeros(64, device=DEV)
    ops.pointwise_wgrad_ss3d(x, y, dtot, dW2, db2, B, S, stats, o_ss, arg, g_ss, g_mx)
    for mine, ref, name in ((dW, dW2, 'dW'), (db, db2, 'db')):
        e = float((mine - ref).abs().max()) / float(ref.abs().max())
        assert e < 2e-5, (name, e)


@pytest.mark.parametrize('B,V,gain,wscale', [(2, 10, 1.0, 0.05), (1, 23, 3e-7, 0.05), (3, 12, 2e5, 0.05), (2, 10, 1.0, 1.0), (2, 10, 4e3, 1.0)])
def test_patchify_data_gradient_folded_into_the_input_weight_gradient(B, V, gain, wscale):
    """vxb_patch_dgrad_input_wgrad_f32: the patchify block's data gradient (k = stride = 5, replicate padding 2) never becomes a
    tensor -- its share of dW_in [64][10] / db_in [64] straight from dpatch, against torch autograd in float64: the gradient of the
    replicate-padded stride-5 conv w.r.t. d0, times LeakyReLU'(d0), times the voxel inputs.  Ragged grids (V = 23, 12: voxels the
    patches never reach get nothing), tiny / huge gradients (device-side power-of-two scale), accumulation into non-zero dW / db.
    wscale = 1.0 (round-5 advisor): patchify weights 20 x the released scale -- the data gradient G = sum of 64 products of dpatch
    (scaled to max 2^15) and weights goes through fp16 at 2^-4 of its scaled value: |G| 2^-4 < 65504 holds for max |W| < 0.5 in the worst
    case and with Gaussian weights of unit scale in practice (patch_wgrad.hip); same 2e-3 bound."""
    k, pad, C, Cin = 5, 2, 64, 10
    G = (V + 2 * pad - k) // k + 1
    d0 = rnd(B, C, V, V, V, seed=1)
    vox = rnd(B, Cin, V, V, V, seed=2)
    Wp = rnd(C, C, k, k, k, seed=3) * wscale
    dpatch = rnd(B, C, G, G, G, seed=4) * gain
    x = d0.double().requires_grad_(True)
    out = F.conv3d(F.pad(x, (pad,) * 6, mode='replicate'), Wp.double(), stride=k)
    assert out.shape[-1] == G
    wp64 = Wp.double().requires_grad_(True)
    out_w = F.conv3d(F.pad(d0.double(), (pad,) * 6, mode='replicate'), wp64, stride=k)
    dWp_ref, = torch.autograd.grad(out_w, wp64, dpatch.double())            # the patchify conv's own weight gradient (round 5: same launch)
    dd0, = torch.autograd.grad(out, x, dpatch.double())
    m = torch.where(d0 > 0, 1.0, ops.LRELU_SLOPE).double()
    g = (dd0 * m).permute(0, 2, 3, 4, 1).reshape(-1, C)                       # [voxels][64]
    xv = vox.double().permute(0, 2, 3, 4, 1).reshape(-1, Cin)
    dW_ref, db_ref = g.t() @ xv, g.sum(0)
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    dW = torch.full((C, Cin), 0.25 * gain, device=DEV)
    db = torch.full((C,), -0.5 * gain, device=DEV)
    keep = ops.PRECISION, ops.WGRAD_PRECISION
    ops.PRECISION, ops.WGRAD_PRECISION = 'bf16x3', 'fp16'
    try:
        assert ops.patch_dgrad_input_wgrad_ok(k, k, C, Cin)
        dWp = torch.full((C, C, k, k, k), 0.125 * gain, device=DEV)
        ops.patch_dgrad_input_wgrad(cl(dpatch), Wp.to(DEV), cl(d0), cl(vox), dW, db, B, V, G, k, pad, dWp=dWp)
        dW2, db2 = torch.zeros_like(dW), torch.zeros_like(db)
        ops.patch_dgrad_input_wgrad(cl(dpatch), Wp.to(DEV), cl(d0), cl(vox), dW2, db2, B, V, G, k, pad)      # (without it: the same dW / db bits)
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION = keep
    eW = float((dW.double().cpu() - 0.25 * gain - dW_ref).abs().max() / dW_ref.abs().max())
    eb = float((db.double().cpu() + 0.5 * gain - db_ref).abs().max() / db_ref.abs().max())
    assert eW < 2e-3 and eb < 2e-3, (eW, eb)
    assert float((dW2 - (dW - 0.25 * gain)).abs().max()) <= 2e-6 * float(dW2.abs().max())
    ep = float((dWp.double().cpu() - 0.125 * gain - dWp_ref).abs().max() / dWp_ref.abs().max())
    assert ep < 2e-3, ep


def test_patchify_weight_gradient_propagates_a_nan_of_d0():
    """round-5 advisor: the d0 operand of the fused patchify weight gradient goes through vxb_sat_f16 (NaN / inf stay non-finite) -- a
    plain v_med3 would turn a NaN activation into -65504 and hand the optimizer finite garbage where the reference's autograd hands on NaN."""
    B, V, k, pad, C, Cin = 1, 10, 5, 2, 64, 10
    G = (V + 2 * pad - k) // k + 1
    d0 = rnd(B, C, V, V, V, seed=1)
    d0[0, 7, 3, 4, 5] = float('nan')
    vox, Wp, dpatch = rnd(B, Cin, V, V, V, seed=2), rnd(C, C, k, k, k, seed=3) * 0.05, rnd(B, C, G, G, G, seed=4)
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    dW, db, dWp = torch.zeros((C, Cin), device=DEV), torch.zeros((C,), device=DEV), torch.zeros((C, C, k, k, k), device=DEV)
    keep = ops.PRECISION, ops.WGRAD_PRECISION
    ops.PRECISION, ops.WGRAD_PRECISION = 'bf16x3', 'fp16'
    try:
        ops.patch_dgrad_input_wgrad(cl(dpatch), Wp.to(DEV), cl(d0), cl(vox), dW, db, B, V, G, k, pad, dWp=dWp)
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION = keep
    assert bool(torch.isnan(dWp[:, 7]).any())                 # the weight gradient column of the NaN input channel
    assert bool(torch.isfinite(dWp[:, 8]).all())              # (and nothing else)
