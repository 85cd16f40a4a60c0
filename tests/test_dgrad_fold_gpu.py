"""GPU: data gradient of a 3x3x3 replicate-padded conv fused with the adjoint of its padding (vxb_conv3_dgrad_fold_f32)
against the two-kernel path (zero-padded conv on the padded domain + vxb_fold_pad_f32) and against autograd of the
PyTorch conv (float64) in bf16x3."""
import pytest
import torch
import torch.nn.functional as F

from voxactb_amd import ops
from .test_ops_gpu import rnd, close, cl, ref_conv, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('S,B', [(16, 2), (20, 1), (34, 1)])
def test_dgrad_fold_matches_two_kernel_path(mode, S, B):
    C, N = 64, 128
    dy = cl(rnd(B, C, S, S, S, seed=3)).to(DEV)
    W = rnd(C, N, 3, 3, 3, seed=1, scale=0.1).to(DEV)              # conv N -> C, so its data gradient has N columns
    y1 = cl(rnd(B, 64, S, S, S, seed=5)).to(DEV)
    base0 = cl(rnd(B, 64, S, S, S, seed=6)).to(DEV)
    ops.PRECISION = mode
    try:
        assert ops.dgrad_fold_ok(C, N, S)
        wd = ops.conv_weight_dgrad(W)
        dcat = ops.conv3d(dy, wd, N, B, S, S + 2, 3, -2, replicate=False)
        r0 = base0.clone()
        ops.fold_pad(dcat, S + 2, N, 0, r0, B, S, 64, 1, accumulate=True)
        r1 = torch.empty(B, S, S, S, 64, device=DEV)
        ops.fold_pad(dcat, S + 2, N, 64, r1, B, S, 64, 1, lrelu_of=y1)
        g0 = base0.clone()
        g1 = torch.full((B, S, S, S, 64), 3.0, device=DEV)
        ops.conv3_dgrad_fold(dy, wd, B, S, N, [(g0, True, None), (g1, False, y1)])
    finally:
        ops.PRECISION = 'fp32'
    close(g0, r0, 3e-5, 'fused fold, accumulate block ' + mode)
    close(g1, r1, 3e-5, 'fused fold, lrelu block ' + mode)


def test_dgrad_fold_x3_vs_autograd():
    B, S, C, N = 1, 16, 64, 64
    x = rnd(B, N, S, S, S, seed=2).double().requires_grad_(True)
    W = rnd(C, N, 3, 3, 3, seed=1, scale=0.1)
    dy = rnd(B, C, S, S, S, seed=3)
    ref_conv(x, W.double(), None).backward(dy.double())
    out = torch.zeros(B, S, S, S, 64, device=DEV)
    ops.PRECISION = 'bf16x3'
    try:
        ops.conv3_dgrad_fold(cl(dy).to(DEV), ops.conv_weight_dgrad(W.to(DEV)), B, S, N, [(out, False, None)])
    finally:
        ops.PRECISION = 'fp32'
    close(out, cl(x.grad.float()), 2e-5, 'fused dgrad + fold vs autograd')


@pytest.mark.parametrize('S,B,gain', [(16, 2, 1.0), (20, 1, 2e-8), (34, 1, 3e5)])
def test_dgrad_fold_leaf_block_on_single_fp16_products(S, B, gain):
    """`leaf_blocks=(0,)` with WGRAD_PRECISION = 'fp16': column block 0 (a gradient that only feeds a weight gradient) runs on single
    fp16 products with dy scaled on the device; block 1 (it propagates) runs on TWO fp16 products -- dy as an fp16 hi + lo pair, the
    weights as one fp16 value: only the weight rounding (2^-12 per weight) separates it from the bf16x3 result -- or, with
    DGRAD_PRECISION = 'bf16x3', stays bit-identical to the all-bf16x3 call; dy of ordinary, tiny and huge magnitude (the scale comes
    from the tensor)."""
    C, N = 64, 128
    dy = (cl(rnd(B, C, S, S, S, seed=3)) * gain).to(DEV)
    dy[0, 0, 0, 0, :8] *= 40.0
    W = rnd(C, N, 3, 3, 3, seed=1, scale=0.1).to(DEV)
    y1 = cl(rnd(B, 64, S, S, S, seed=5)).to(DEV)
    base0 = (cl(rnd(B, 64, S, S, S, seed=6)) * gain).to(DEV)
    ops.PRECISION = 'bf16x3'
    try:
        wd = ops.conv_weight_dgrad(W)
        r0, r1 = base0.clone(), torch.empty(B, S, S, S, 64, device=DEV)
        ops.conv3_dgrad_fold(dy, wd, B, S, N, [(r0, True, None), (r1, False, y1)])
        ops.WGRAD_PRECISION = 'fp16'
        g0, g1 = base0.clone(), torch.empty(B, S, S, S, 64, device=DEV)
        ops.conv3_dgrad_fold(dy, wd, B, S, N, [(g0, True, None), (g1, False, y1)], leaf_blocks=(0,))
        h0, h1 = base0.clone(), torch.empty(B, S, S, S, 64, device=DEV)
        ops.conv3_dgrad_fold(dy, wd, B, S, N, [(h0, True, None), (h1, False, y1)], leaf_blocks=(0,), dy_scale=ops.absmax_scale(dy))
        dp, ops.DGRAD_PRECISION = ops.DGRAD_PRECISION, 'bf16x3'
        k0, k1 = base0.clone(), torch.empty(B, S, S, S, 64, device=DEV)
        ops.conv3_dgrad_fold(dy, wd, B, S, N, [(k0, True, None), (k1, False, y1)], leaf_blocks=(0,))
        ops.DGRAD_PRECISION = dp
    finally:
        ops.PRECISION = 'fp32'
        ops.WGRAD_PRECISION = ''
    assert torch.equal(k1, r1) and torch.equal(k0, g0)                       # 'bf16x3' data gradients: the propagating block untouched
    assert torch.equal(g1, h1)
    if ops.DGRAD_PRECISION == 'fp16x2':
        e1 = float((g1 - r1).abs().max()) / float(r1.abs().max())
        assert 0 < e1 < 4e-4, e1                                             # weights rounded to 11 bits, dy carried in 22
    else:
        assert torch.equal(g1, r1)
    assert torch.equal(g0, h0)
    err = float((g0 - r0).abs().max()) / float((r0 - base0).abs().max())
    assert 0 < err < 1.5e-3, err                                             # 2^-12 per operand over a 27 x 64 term sum


def test_fused_operand_scales_equal_the_absmax_pass():
    """the fp16 operand scales taken inside the producers (the fused du kernel, the data gradient's fold epilogue) are the ones
    vxb_absmax_scale_f32 computes from the finished tensors"""
    B, S, C = 2, 20, 64
    dy = cl(rnd(B, C, S, S, S, seed=3) * 3e-4).to(DEV)
    W = rnd(C, 128, 3, 3, 3, seed=1, scale=0.1).to(DEV)
    y1 = cl(rnd(B, 64, S, S, S, seed=5)).to(DEV)
    ops.PRECISION, ops.WGRAD_PRECISION = 'bf16x3', 'fp16'
    try:
        g0, g1 = torch.zeros(B, S, S, S, 64, device=DEV), torch.empty(B, S, S, S, 64, device=DEV)
        bias = torch.full((64,), 0.25, device=DEV)
        sc = ops.conv3_dgrad_fold(dy, ops.conv_weight_dgrad(W), B, S, 128, [(g0, False, None), (g1, False, y1)], leaf_blocks=(0,),
                                  scale_blocks=(1,), colsum_into={1: bias})
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION = 'fp32', ''
    assert torch.equal(sc[1], ops.absmax_scale(g1))
    want = g1.double().sum(dim=(0, 1, 2, 3)).cpu() + 0.25          # column sums of the block as written, accumulated into `bias`
    assert float((bias.double().cpu() - want).abs().max()) < 1e-5 * float(want.abs().max())
    # the fused backward of u (c1 data gradient + SpatialSoftmax3D term + LeakyReLU' + bias sums)
    u = cl(rnd(B, 64, S, S, S, seed=8)).to(DEV)
    dq = rnd(B, S, S, S, seed=9).to(DEV) * 1e-5
    w = rnd(1, 64, 3, 3, 3, seed=10, scale=0.1).to(DEV)
    ss, mx, st, am = ops.ss3d_max_fwd(u, S ** 3 * 64, B, S, 64)
    gss, gmx = rnd(B, 192, seed=11).to(DEV) * 1e-3, rnd(B, 64, seed=12).to(DEV) * 1e-3
    du = torch.empty_like(u)
    db = torch.zeros(64, device=DEV)
    _, sc_du = ops.conv3_c1_dgrad_ss3d(dq, w, u, du, B, S, st, ss, am, gss, gmx, db, want_scale=True)
    assert torch.equal(sc_du, ops.absmax_scale(du))


@pytest.mark.parametrize('B,S,gain', [(2, 18, 1.0), (1, 22, 1e-6), (1, 100, 1.0), (2, 38, 1.0)])
def test_leaf_block_folded_into_the_input_weight_gradient(B, S, gain):
    """vxb_conv3_dgrad_fold_f16_wgin_f32: the d(d0) block of `final`'s data gradient is not stored -- its epilogue multiplies it with
    LeakyReLU'(d0) and the voxel inputs and accumulates dW_in [64][10] / db_in [64]: against the stored-tensor route (same fp16
    kernel, then the products in float64)."""
    from .test_ops_gpu import rnd, cl, DEV
    C = 64
    du = cl(rnd(B, C, S, S, S, seed=1) * gain).to(DEV)
    Wf = (rnd(C, 2 * C, 3, 3, 3, seed=2) * 0.05).to(DEV)
    d0 = cl(rnd(B, C, S, S, S, seed=3)).to(DEV)
    u0 = cl(rnd(B, C, S, S, S, seed=4)).to(DEV)
    vox = cl(rnd(B, 10, S, S, S, seed=5)).to(DEV)
    keep = ops.PRECISION, ops.WGRAD_PRECISION, ops.WGIN_FOLD
    ops.PRECISION, ops.WGRAD_PRECISION, ops.WGIN_FOLD = 'bf16x3', 'fp16', True      # (the fold is opt-in: VOXACTB_WGIN_FOLD=1)
    try:
        assert ops.wgin_fold_ok(C, 2 * C, S, 10)
        wt = ops.conv_weight_dgrad(Wf)
        dd0, du0 = torch.empty_like(d0), torch.empty_like(d0)
        ops.conv3_dgrad_fold(du, wt, B, S, 2 * C, [(dd0, False, None), (du0, False, u0)], leaf_blocks=(0,))
        g = (dd0.double() * torch.where(d0 > 0, 1.0, ops.LRELU_SLOPE).double()).view(-1, C)
        dW_ref, db_ref = g.t() @ vox.double().view(-1, 10), g.sum(0)
        dW, db = torch.full((C, 10), 0.5 * gain, device=DEV), torch.full((C,), -0.25 * gain, device=DEV)
        du0b = torch.empty_like(d0)
        ops.conv3_dgrad_fold(du, wt, B, S, 2 * C, [(None, False, None), (du0b, False, u0)], leaf_blocks=(0,),
                             wgin={0: (d0, vox, dW, db)})
        dW2, db2 = torch.full((C, 10), 0.5 * gain, device=DEV), torch.full((C,), -0.25 * gain, device=DEV)
        ops.conv3_dgrad_fold(du, wt, B, S, 2 * C, [(None, False, None), (du0b, False, u0)], leaf_blocks=(0,),
                             wgin={0: (d0, vox, dW2, db2)})
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.WGIN_FOLD = keep
    assert torch.equal(du0, du0b)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)        # fixed-order sums: the same bits every run
    eW = float((dW.double() - 0.5 * gain - dW_ref).abs().max() / dW_ref.abs().max())
    eb = float((db.double() + 0.25 * gain - db_ref).abs().max() / db_ref.abs().max())
    assert eW < 2e-5 and eb < 2e-5, (eW, eb)


@pytest.mark.parametrize('gain', [1.0, 3e-9, 4e6])
def test_two_product_data_gradients_match_bf16x3_up_to_the_weight_rounding(gain):
    """DGRAD_PRECISION = 'fp16x2' on both LDS-halo data gradients that propagate (final's d(u0) block and the polyphase up-conv's tap-list
    launch): with weights that ARE fp16 values the two-product result equals the bf16x3 one to its own rounding (the gradient operand
    is carried as an fp16 hi + lo pair: 22 bits); with arbitrary weights the difference is the 2^-12 weight rounding.  Gradients of
    ordinary, tiny and huge magnitude with a 50x spike (the operand scale comes from the tensor)."""
    B, C, k, s, G = 1, 64, 5, 5, 6
    Lh, R = ops.polyphase_tables(k, s)
    kl = 2 * R + 1
    du = (cl(rnd(B, C, G * s, G * s, G * s, seed=3)) * gain).to(DEV)
    du[0, 1, 2, 3, :4] *= 50.0
    Sp = G + 2 * R
    res = {}
    for exact_w in (True, False):
        Weff = rnd(kl ** 3 * C, s ** 3 * C, seed=1, scale=0.05).to(DEV)
        if exact_w:
            Weff = Weff.half().float()
        ops.PRECISION, ops.WGRAD_PRECISION = 'bf16x3', 'fp16'
        dp = ops.DGRAD_PRECISION
        try:
            wd = ops.polyphase_dgrad_weights_lowres(Weff, C, C, s, kl)
            ops.DGRAD_PRECISION = 'bf16x3'
            ref = ops.conv3_s2d(du, wd, C, B, G, Sp, -(kl - 1), s, C, poly_k=k)
            ops.DGRAD_PRECISION = 'fp16x2'
            got = ops.conv3_s2d(du, wd, C, B, G, Sp, -(kl - 1), s, C, poly_k=k)
            got2 = ops.conv3_s2d(du, wd, C, B, G, Sp, -(kl - 1), s, C, poly_k=k, dy_scale=ops.absmax_scale(du))
        finally:
            ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION = 'fp32', '', dp
        assert torch.equal(got, got2)
        res[exact_w] = float((got - ref).abs().max()) / float(ref.abs().max())
    assert res[True] < 3e-5, res                  # fp32 accumulation noise of K = 8000 x 17.6 terms, as between two bf16x3 associations
    assert 0 < res[False] < 4e-4, res
