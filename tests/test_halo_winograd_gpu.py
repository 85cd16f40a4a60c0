"""GPU: `final`'s forward with the filter's depth axis by Winograd F(2, 3) (vxb_conv3_halo_ss3d_wg_bf16x3_f32; conv_halo_bf16.hip, WG)
against a float64 PyTorch conv (the bound of the direct bf16x3 kernel: 2e-5 of the output maximum), against the direct kernel (the
transforms cost fp32 roundings of two-term input sums and three-term weight sums: a few 1e-7 of the maximum), and its epilogue statistics
against the statistics pass over ITS OWN output.  Reference module: helpers/network_utils.py:128-170 (Conv3DBlock), used at
perceiver_lang_io.py:462 with replicate padding, then SpatialSoftmax3D / max pool (:470)."""
import pytest
import torch
import torch.nn.functional as F

from voxactb_amd import ops
from .test_ops_gpu import rnd, cl, ref_conv, DEV

pytestmark = pytest.mark.gpu


def _run(d0, u0, W, bias, B, S, wino):
    old = ops.FINAL_WINOGRAD
    ops.FINAL_WINOGRAD = wino
    ops.PRECISION = 'bf16x3'
    try:
        return ops.conv3_ss3d_fwd(d0, u0, ops.conv_weight_fwd(W), bias, B, S)
    finally:
        ops.PRECISION = 'fp32'
        ops.FINAL_WINOGRAD = old


@pytest.mark.parametrize('S,B', [(16, 2), (18, 1), (20, 2), (22, 1), (28, 1), (36, 1), (50, 1)])       # 20, 28, 36: half tiles along h / w; 18, 22, 50: a two-deep last depth tile, 6- and 2-wide edges
def test_winograd_forward_against_float64_and_the_direct_kernel(S, B):
    C = 64
    a, c = rnd(B, C, S, S, S, seed=1), rnd(B, C, S, S, S, seed=2)
    W = rnd(C, 2 * C, 3, 3, 3, seed=3, scale=0.03)
    bias = rnd(C, seed=4)
    ref = cl(F.leaky_relu(ref_conv(torch.cat([a, c], 1).double(), W.double(), bias.double()), 0.02)).float()
    d0, u0 = cl(a).to(DEV), cl(c).to(DEV)
    got, (g_ss, g_max, g_stats, g_arg) = _run(d0, u0, W.to(DEV), bias.to(DEV), B, S, True)
    direct, _ = _run(d0, u0, W.to(DEV), bias.to(DEV), B, S, False)
    mx = float(ref.abs().max())
    e_w = float((got.cpu() - ref).abs().max()) / mx
    e_d = float((direct.cpu() - ref).abs().max()) / mx
    e_wd = float((got - direct).abs().max()) / mx
    print('S=%d: vs float64  winograd %.2e  direct %.2e | winograd vs direct %.2e (of the output maximum)' % (S, e_w, e_d, e_wd))
    assert e_w < 2e-5 and e_wd < 2e-5
    # the statistics of the epilogue are those of the tensor it wrote
    r_ss, r_max, r_stats, r_arg = ops.ss3d_max_fwd(got, S ** 3 * C, B, S, C)
    assert torch.equal(g_arg, r_arg) and torch.equal(g_max, r_max)
    assert torch.equal(g_stats[..., 0], r_stats[..., 0])
    assert float((g_stats[..., 1] - r_stats[..., 1]).abs().max() / r_stats[..., 1].abs().max()) < 2e-6
    assert float((g_ss - r_ss).abs().max()) < 2e-6


def test_winograd_at_the_headline_grid():
    """S = 100 (25 whole depth tiles, a half tile along h and w), B = 1: against the direct kernel."""
    S, B, C = 100, 1, 64
    d0 = cl(rnd(B, C, S, S, S, seed=1)).to(DEV)
    u0 = cl(rnd(B, C, S, S, S, seed=2)).to(DEV)
    W = rnd(C, 2 * C, 3, 3, 3, seed=3, scale=0.03).to(DEV)
    bias = rnd(C, seed=4).to(DEV)
    got, st_w = _run(d0, u0, W, bias, B, S, True)
    direct, st_d = _run(d0, u0, W, bias, B, S, False)
    mx = float(direct.abs().max())
    e = float((got - direct).abs().max()) / mx
    print('S=100: winograd vs direct %.2e of the output maximum' % e)
    assert e < 2e-5
    # (the pooled features of the two variants are NOT compared: at temperature 0.01 on random data a 6e-6 difference of u moves the
    # expected coordinates by 1e-3 -- each epilogue is checked against the statistics pass over its own output)
    r_ss, r_max, r_stats, r_arg = ops.ss3d_max_fwd(got, S ** 3 * C, B, S, C)
    assert torch.equal(st_w[3], r_arg) and torch.equal(st_w[1], r_max)
    assert float((st_w[0] - r_ss).abs().max()) < 2e-6


def test_unsupported_depths_fall_back_to_the_direct_kernel():
    S, B, C = 21, 1, 64                     # odd S: a depth pair would straddle the grid's end
    d0 = cl(rnd(B, C, S, S, S, seed=1)).to(DEV)
    u0 = cl(rnd(B, C, S, S, S, seed=2)).to(DEV)
    W = rnd(C, 2 * C, 3, 3, 3, seed=3, scale=0.03).to(DEV)
    bias = rnd(C, seed=4).to(DEV)
    a, _ = _run(d0, u0, W, bias, B, S, True)
    b, _ = _run(d0, u0, W, bias, B, S, False)
    assert torch.equal(a, b)


@pytest.mark.parametrize('S,B', [(16, 1), (18, 1), (20, 2), (100, 1)])      # S + 2 = 18 / 20 / 22 / 102: two-deep last depth tiles, 6-wide edges
def test_winograd_data_gradient_against_the_direct_kernel(S, B):
    """the propagating (fp16x2) half of `final`'s data gradient + padding adjoint (vxb_conv3_dgrad_fold_f16x2_wg_f32) against the direct
    launch and, for the small grids, a float64 conv_transpose3d folded by the adjoint of the replicate padding."""
    C, N = 64, 128
    dy = cl(rnd(B, C, S, S, S, seed=3)).to(DEV)
    W = rnd(C, N, 3, 3, 3, seed=1, scale=0.1).to(DEV)
    y1 = cl(rnd(B, 64, S, S, S, seed=5)).to(DEV)
    state = (ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION, ops.DGRAD_WINOGRAD, ops.LEAF_WINOGRAD)
    try:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION = 'bf16x3', 'fp16', 'fp16x2'
        wd = ops.conv_weight_dgrad(W)

        def run(wino):
            ops.DGRAD_WINOGRAD = ops.LEAF_WINOGRAD = wino
            g0, g1 = torch.ones(B, S, S, S, 64, device=DEV), torch.empty(B, S, S, S, 64, device=DEV)
            cs = torch.zeros(64, device=DEV)
            sc = ops.conv3_dgrad_fold(dy, wd, B, S, N, [(g0, True, None), (g1, False, y1)], leaf_blocks=(0,), scale_blocks=(1,),
                                      colsum_into={1: cs})
            return g1, cs, sc[1], g0
        d_g, d_cs, d_sc, d_leaf = run(False)
        w_g, w_cs, w_sc, w_leaf = run(True)
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION, ops.DGRAD_WINOGRAD, ops.LEAF_WINOGRAD = state
    # the leaf block (channels 0..63, single fp16 products, accumulated onto ones): both round dy AND the weights to 11 bits
    e_leaf = float((w_leaf - d_leaf).abs().max()) / float((d_leaf - 1.0).abs().max())
    print('S=%d: leaf block, winograd vs direct %.2e of the largest element' % (S, e_leaf))
    assert e_leaf < 2e-3
    mx = float(d_g.abs().max())
    e = float((w_g - d_g).abs().max()) / mx
    print('S=%d: winograd vs direct data gradient %.2e of the largest element' % (S, e))
    # (both round their weights to fp16, 2^-12 per product -- the direct one the taps, this one the transformed taps: they differ by that)
    assert e < 1e-3
    assert float((w_cs - d_cs).abs().max()) <= 1e-3 * float(d_cs.abs().max()) + 1e-3
    assert torch.equal(w_sc, d_sc) or float((w_sc[0] / d_sc[0])) in (0.5, 1.0, 2.0)       # the same power of two unless the maximum sits at a binade edge
    if S <= 20:
        # float64: dX_padded = conv_transpose3d(dy, W) on the (S + 2)^3 domain, folded back by the adjoint of the replicate padding,
        # times LeakyReLU'(y1) -- the channels 64..127 of the input (block 1)
        dyc = dy.permute(0, 4, 1, 2, 3).double().cpu()
        full = F.conv_transpose3d(dyc, W.double().cpu())[:, 64:]                       # [B, 64, S+2, S+2, S+2]
        for ax in (2, 3, 4):
            n = full.shape[ax]
            idx = torch.clamp(torch.arange(n) - 1, 0, n - 3)
            out = torch.zeros(full.shape[:ax] + (n - 2,) + full.shape[ax + 1:], dtype=full.dtype)
            out.index_add_(ax, idx, full)
            full = out
        ref = cl(full) * torch.where(y1.cpu() > 0, 1.0, 0.02).double()
        ew = float((w_g.cpu().double() - ref).abs().max()) / float(ref.abs().max())
        ed = float((d_g.cpu().double() - ref).abs().max()) / float(ref.abs().max())
        print('      vs float64: winograd %.2e  direct %.2e' % (ew, ed))
        assert ew < 1e-3 and ew < 1.5 * ed + 1e-5           # (the fp16 rounding of the WEIGHTS, 2^-12 per product, bounds both)


@pytest.mark.parametrize('S,B', [(20, 2), (22, 1), (36, 1)])      # half tiles along h / w; a two-deep last depth tile (idle wave rows take the barriers)
def test_weight_fragments_through_lds_are_bit_identical(S, B):
    """Round 6: the Winograd variants (bf16x3 forward, fp16x2 data gradient) fetch a tap's weight fragments once per workgroup into an
    LDS ring and read the halo out of a compact, slot-swizzled image (conv3_halo_body, BL); vxb_debug_set_halo_experiment(0x1000) keeps
    the per-wave fragment loads and the padded image of rounds 5 - 6.  Same products in the same order: equal bits."""
    from voxactb_amd import _lib
    C = 64
    d0, u0 = cl(rnd(B, C, S, S, S, seed=1)).to(DEV), cl(rnd(B, C, S, S, S, seed=2)).to(DEV)
    W = rnd(C, 2 * C, 3, 3, 3, seed=3, scale=0.03).to(DEV)
    bias = rnd(C, seed=4).to(DEV)
    dy = cl(rnd(B, C, S, S, S, seed=5)).to(DEV)
    y1 = cl(rnd(B, 64, S, S, S, seed=6)).to(DEV)
    Wd = rnd(C, 2 * C, 3, 3, 3, seed=7, scale=0.1).to(DEV)
    out = {}
    try:
        for bits in (0, 0x1000):
            _lib.lib().vxb_debug_set_halo_experiment(bits)
            fwd, st = _run(d0, u0, W, bias, B, S, True)
            ops.PRECISION, ops.WGRAD_PRECISION = 'bf16x3', 'fp16'
            try:
                assert ops.DGRAD_PRECISION == 'fp16x2' and ops.DGRAD_WINOGRAD
                g0 = torch.zeros(B, S, S, S, 64, device=DEV)
                g1 = torch.full((B, S, S, S, 64), 3.0, device=DEV)
                ops.begin_backward()
                ops.conv3_dgrad_fold(dy, ops.conv_weight_dgrad(Wd), B, S, 2 * C, [(g0, False, None), (g1, False, y1)], leaf_blocks=(0,))
            finally:
                ops.PRECISION, ops.WGRAD_PRECISION = 'fp32', ''
            out[bits] = (fwd, st[0], st[1], st[3], g1)
    finally:
        _lib.lib().vxb_debug_set_halo_experiment(0)
    for a, b in zip(out[0], out[0x1000]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('S,B', [(20, 1), (22, 1)])
def test_direct_kernels_with_the_fragment_ring_are_bit_identical(S, B):
    """The non-Winograd variants whose waves all multiply both column tiles (the direct bf16x3 forward, the fp16 d(d0) data gradient) take
    their weight fragments through the same LDS ring (conv3_halo_body, BN): equal bits with vxb_debug_set_halo_experiment(0x1000)."""
    from voxactb_amd import _lib
    C = 64
    d0, u0 = cl(rnd(B, C, S, S, S, seed=1)).to(DEV), cl(rnd(B, C, S, S, S, seed=2)).to(DEV)
    W = rnd(C, 2 * C, 3, 3, 3, seed=3, scale=0.03).to(DEV)
    bias = rnd(C, seed=4).to(DEV)
    dy = cl(rnd(B, C, S, S, S, seed=5)).to(DEV)
    y1 = cl(rnd(B, 64, S, S, S, seed=6)).to(DEV)
    Wd = rnd(C, 2 * C, 3, 3, 3, seed=7, scale=0.1).to(DEV)
    out = {}
    try:
        for bits in (0, 0x1000):
            _lib.lib().vxb_debug_set_halo_experiment(bits)
            fwd, st = _run(d0, u0, W, bias, B, S, False)                # the direct forward
            ops.PRECISION, ops.WGRAD_PRECISION = 'bf16x3', 'fp16'
            try:
                g0 = torch.zeros(B, S, S, S, 64, device=DEV)
                g1 = torch.full((B, S, S, S, 64), 3.0, device=DEV)
                ops.begin_backward()
                ops.conv3_dgrad_fold(dy, ops.conv_weight_dgrad(Wd), B, S, 2 * C, [(g0, False, None), (g1, False, y1)], leaf_blocks=(0,))
            finally:
                ops.PRECISION, ops.WGRAD_PRECISION = 'fp32', ''
            out[bits] = (fwd, st[0], st[1], st[3], g0, g1)
    finally:
        _lib.lib().vxb_debug_set_halo_experiment(0)
    for a, b in zip(out[0], out[0x1000]):
        assert torch.equal(a, b)
