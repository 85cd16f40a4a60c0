"""GPU: the block-sparse evaluation of the polyphase up-conv (network_utils.py:245-250: a fine phase only reaches the
low-res taps under its trilinear footprint, 17.6 of 27 weight blocks on average) against the dense evaluation of the
same weights.  The skipped products are exact zeros and the surviving ones keep their order: bit-identical."""
import numpy as np
import pytest
import torch

from voxactb_amd import ops
from .test_ops_gpu import rnd, cl, DEV

pytestmark = pytest.mark.gpu


def _weff(C, k, s, seed=1):
    W = rnd(C, C, k, k, k, seed=seed, scale=0.05).to(DEV)
    Lh, R = ops.polyphase_tables(k, s)
    Lt = torch.from_numpy(Lh).to(DEV)
    return ops.polyphase_weights(W, Lt, s, 2 * R + 1), Lt, R


def test_structure_matches_the_weights():
    """every block the structure calls zero IS zero in W_eff, and the non-zero fraction is 2.6^3 / 27 for k = s = 5."""
    C, k, s = 64, 5, 5
    Weff, _, R = _weff(C, k, s)
    kl = 2 * R + 1
    st = ops.polyphase_structure(k, s, DEV)
    blk = Weff.view(kl ** 3, C, s ** 3, C).abs().amax(dim=(1, 3)).cpu().numpy()          # [tap][phase]
    for ph in range(s ** 3):
        for tap in range(kl ** 3):
            if not (st['phase_mask'][ph] >> tap) & 1:
                assert blk[tap, ph] == 0.0, (tap, ph)
    assert abs(st['frac'] - 2.6 ** 3 / 27.0) < 1e-9
    assert sorted(st['order']) == list(range(s ** 3))
    tm = st['tile_mask'].cpu().tolist()
    for i, m in enumerate(tm):
        want = st['phase_mask'][st['order'][2 * i]] | (st['phase_mask'][st['order'][2 * i + 1]] if 2 * i + 1 < s ** 3 else 0)
        assert m == want and m != 0


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('k,s,G', [(5, 5, 6), (3, 2, 9)])
def test_sparse_forward_is_bit_identical_to_dense(mode, k, s, G):
    B, C = 2, 64
    Weff, _, R = _weff(C, k, s)
    kl = 2 * R + 1
    z = cl(rnd(B, C, G, G, G, seed=4)).to(DEV)
    bias = rnd(C, seed=2).to(DEV).repeat(s ** 3)
    ops.PRECISION = mode
    try:
        assert ops.polyphase_fwd_ok(C, C, kl, B, G)
        got = ops.conv3_polyphase_fwd(z, Weff, C, B, G, k, s, bias, act=ops.ACT_LRELU)
        ops.HALO_CONV = False
        ref = ops.conv3d(z, Weff, s ** 3 * C, B, G, G, kl, -R, bias=bias, act=ops.ACT_LRELU, d2s=(s, C))
    finally:
        ops.PRECISION = 'fp32'
        ops.HALO_CONV = True
    assert got.shape == ref.shape == (B, G * s, G * s, G * s, C)
    assert torch.equal(got, ref)


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
def test_sparse_weight_gradient_matches_dense_on_the_nonzero_blocks(mode):
    """the masked LDS-halo weight gradient returns the dense result on every structurally non-zero (tap, phase) block
    (same tiles, same order: bit-identical) and zeros elsewhere -- which is all polyphase_weights_bwd reads."""
    B, C, k, s, G = 1, 64, 5, 5, 16          # (the LDS-halo kernel takes grids of 16^3 and up)
    st = ops.polyphase_structure(k, s, DEV)
    kl, R = st['kl'], st['R']
    z = cl(rnd(B, C, G, G, G, seed=4)).to(DEV)
    dy = cl(rnd(B, C, G * s, G * s, G * s, seed=6)).to(DEV)
    dense = ops.conv3d_wgrad(z, dy, s ** 3 * C, B, G, G, kl, -R, d2s=(s, C), force_bf16=mode, nsplit=2)
    got = ops.conv3d_wgrad(z, dy, s ** 3 * C, B, G, G, kl, -R, d2s=(s, C), force_bf16=mode, nsplit=2,
                           phase_mask=st['phase_mask_t'], flops_frac=st['frac'])
    keep = torch.tensor([[(m >> t) & 1 for m in st['phase_mask']] for t in range(kl ** 3)], dtype=torch.bool, device=DEV)
    keep = keep[:, None, :, None].expand(kl ** 3, C, s ** 3, C).reshape(kl ** 3 * C, s ** 3 * C)
    assert torch.equal(got[keep], dense[keep])
    assert float(got[~keep].abs().max()) == 0.0
    # and through the adjoint of the weight construction: identical parameter gradients
    Lh, _ = ops.polyphase_tables(k, s)
    Lt = torch.from_numpy(Lh).to(DEV)
    dW0 = torch.zeros(C, C, k, k, k, device=DEV)
    dW1 = torch.zeros(C, C, k, k, k, device=DEV)
    ops.polyphase_weights_bwd(dense, Lt, dW0, s, kl)
    ops.polyphase_weights_bwd(got, Lt, dW1, s, kl)
    assert float((dW0 - dW1).abs().max()) <= 1e-6 * float(dW0.abs().max())


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
def test_sparse_data_gradient_is_bit_identical_to_dense(mode):
    """space-to-depth LDS-halo conv with per-phase tap lists (only the non-zero blocks of the flipped weights, lists
    padded to a multiple of three with zero-weight taps) against the same kernel visiting all 27 taps."""
    B, C, k, s, G = 2, 64, 5, 5, 6
    Weff, _, R = _weff(C, k, s)
    kl = 2 * R + 1
    du = cl(rnd(B, C, G * s, G * s, G * s, seed=3)).to(DEV)
    Sp = G + 2 * R
    ops.PRECISION = mode
    try:
        wd = ops.polyphase_dgrad_weights_lowres(Weff, C, C, s, kl)
        assert ops.s2d_halo_ok(kl, C, C)
        dense = ops.conv3_s2d(du, wd, C, B, G, Sp, -(kl - 1), s, C)
        ks_default, ops.S2D_KSPLIT = ops.S2D_KSPLIT, 1
        got = ops.conv3_s2d(du, wd, C, B, G, Sp, -(kl - 1), s, C, poly_k=k)
        # the reduction split over several workgroups per tile (the default in the step): same products, the partial sums of the
        # parts added in order -- equal up to the fp32 rounding of K = 8000 x 17.6 terms summed in a different association
        splits = []
        for ks in (2, 3, ks_default, 16):
            ops.S2D_KSPLIT = ks
            splits.append((ks, ops.conv3_s2d(du, wd, C, B, G, Sp, -(kl - 1), s, C, poly_k=k)))
    finally:
        ops.PRECISION = 'fp32'
        ops.S2D_KSPLIT = ks_default
    assert torch.equal(got, dense)
    for ks, sp in splits:
        kp = ops.s2d_kparts(k, s, DEV, 16 if mode == 'bf16x3' else 32, C // (16 if mode == 'bf16x3' else 32), ks).tolist()
        assert kp[0] == 0 and kp[-1] == s ** 3 * C // (16 if mode == 'bf16x3' else 32) and all(a <= b for a, b in zip(kp, kp[1:]))
        err = float((sp - dense).abs().max() / dense.abs().max())
        assert err < 3e-5, (ks, err)
    tt, ncls, total, rows = ops.s2d_taptab(k, s, DEV, 16, 4)
    assert ncls == 27 and total == rows.numel() and total % 3 == 0
    assert total == 4 * (8 * 9 + 36 * 12 + 54 * 18 + 27 * 27)          # 8-tap lists padded to 9


def test_weight_gather_equals_the_aten_layout_chains():
    """ops.gather_cvt with a traced index table (one pass, vxb_gather_cvt_f32) writes the same bits as the ATen chains it replaces: the
    forward's perm8 gather + hi / lo split + fragment shuffle of W_eff, and the data gradient's flip / permute / transpose / fp16 /
    fragment shuffle / tap-list gather (network_utils.py:245-250 backward)."""
    C, k, s = 64, 5, 5
    Weff, _, R = _weff(C, k, s, seed=5)
    kl = 2 * R + 1
    st = ops.polyphase_structure(k, s, DEV)
    N, K = s ** 3 * C, kl ** 3 * C
    # forward: planes in fragment order
    wt = Weff.t().view(s ** 3, C, K).index_select(0, st['perm8_long']).view(N, K)
    ref = ops.gemm_wfrag(ops.split_planes(wt, 2))
    idx = ops.traced_index(('test_polyf', k, s, C), tuple(Weff.shape),
                           lambda I: ops._wfrag_index(I.t().view(s ** 3, C, K).index_select(0, st['perm8_long']).view(N, K), Weff.numel()), DEV)
    got = ops.gather_cvt(Weff, idx, 1, Weff.numel())
    assert got.numel() == ref.numel() and torch.equal(got.view(torch.int16), ref.reshape(-1).view(torch.int16))
    # data gradient: fp16 fragments of the listed taps
    C0 = s ** 3 * C
    tt, ncls, total, rows = ops.s2d_taptab(k, s, DEV, 16, C // 16)
    wd = ops.polyphase_dgrad_weights_lowres(Weff, C, C, s, kl)
    f = ops.halo_wfrag_x2(wd.t().contiguous().half(), C0)
    ref2 = f.view(f.shape[0], f.shape[1] * 27, -1).index_select(1, rows).contiguous()

    def layout(I):
        g = ops.halo_wfrag_x2(ops.polyphase_dgrad_weights_lowres(I, C, C, s, kl).t().contiguous(), C0)
        return g.view(g.shape[0], g.shape[1] * 27, -1).index_select(1, rows).contiguous()
    got2 = ops.gather_cvt(Weff, ops.traced_index(('test_s2dw', k, s, C), tuple(Weff.shape), layout, DEV), 0)
    assert got2.numel() == ref2.numel() and torch.equal(got2.view(torch.int16), ref2.reshape(-1).view(torch.int16))
