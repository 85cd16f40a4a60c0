"""GPU: the part-tile modes of the LDS-halo conv (edge tiles that run only the M tiles holding output voxels: half tiles, 6-of-8 tiles with
the 2-wide strip, two-depth tiles with one depth per wave row; conv_halo_bf16.hip) against the same kernels with every tile run in full
(experiment bit 4 of vxb_debug_set_halo_experiment): the same products in the same order for every output voxel -> bit-identical, in the
forward (two sources, statistics epilogue), the data gradient + padding adjoint and the tap-list launch of the up-conv's data gradient."""
import pytest
import torch

from voxactb_amd import ops, _lib
from .test_ops_gpu import rnd, cl, DEV

pytestmark = pytest.mark.gpu


def _both(fn):
    L = _lib.lib()
    try:
        L.vxb_debug_set_halo_experiment(4)
        full = fn()
    finally:
        L.vxb_debug_set_halo_experiment(0)
    part = fn()
    return full, part


@pytest.mark.parametrize('S,B', [(12, 2), (13, 1), (20, 1), (22, 1), (28, 1), (100, 1)])
def test_forward_part_tiles(S, B):
    C = 64
    d0, u0 = cl(rnd(B, C, S, S, S, seed=1)).to(DEV), cl(rnd(B, C, S, S, S, seed=2)).to(DEV)
    W, bias = rnd(C, 2 * C, 3, 3, 3, seed=3, scale=0.03).to(DEV), rnd(C, seed=4).to(DEV)
    ops.PRECISION = 'bf16x3'
    try:
        wt = ops.conv_weight_fwd(W)
        full, part = _both(lambda: ops.conv3d(d0, wt, C, B, S, S, 3, -1, bias=bias, act=ops.ACT_LRELU, src1=u0))
        assert torch.equal(full, part)
        if ops.conv3_ss3d_ok(C, C, C, S):
            for wino in (False, True):              # the direct kernel, then the Winograd-along-depth one (even S; half tiles and two-deep last depth tiles)
                old = ops.FINAL_WINOGRAD
                ops.FINAL_WINOGRAD = wino
                try:
                    f2, p2 = _both(lambda: ops.conv3_ss3d_fwd(d0, u0, wt, bias, B, S))
                finally:
                    ops.FINAL_WINOGRAD = old
                assert torch.equal(f2[0], p2[0])
                assert wino or torch.equal(f2[0], full)
                for a, b in zip(f2[1], p2[1]):
                    assert torch.equal(a, b)
    finally:
        ops.PRECISION = 'fp32'


@pytest.mark.parametrize('S,B', [(16, 1), (18, 1), (20, 2), (100, 1)])
@pytest.mark.parametrize('dgrad', ['bf16x3', 'fp16x2'])
def test_fold_part_tiles(S, B, dgrad):
    C, N = 64, 128
    dy = cl(rnd(B, C, S, S, S, seed=3)).to(DEV)
    W = rnd(C, N, 3, 3, 3, seed=1, scale=0.1).to(DEV)
    y1 = cl(rnd(B, 64, S, S, S, seed=5)).to(DEV)
    state = (ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION)
    try:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION = 'bf16x3', 'fp16', dgrad
        wd = ops.conv_weight_dgrad(W)

        def run():
            g0, g1 = torch.ones(B, S, S, S, 64, device=DEV), torch.empty(B, S, S, S, 64, device=DEV)
            ops.conv3_dgrad_fold(dy, wd, B, S, N, [(g0, True, None), (g1, False, y1)], leaf_blocks=(0,))
            return g0, g1
        full, part = _both(run)
        assert torch.equal(full[0], part[0]) and torch.equal(full[1], part[1])
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION = state


@pytest.mark.parametrize('G,B', [(4, 2), (10, 1), (20, 1)])
@pytest.mark.parametrize('dgrad', ['bf16x3', 'fp16x2'])
def test_tap_list_part_tiles(G, B, dgrad):
    k, s, C = 5, 5, 64
    Lh, R = ops.polyphase_tables(k, s)
    kl = 2 * R + 1
    Weff = rnd(kl ** 3 * C, s ** 3 * C, seed=1, scale=0.05).to(DEV)
    du = cl(rnd(B, C, G * s, G * s, G * s, seed=3)).to(DEV)
    state = (ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION)
    try:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION = 'bf16x3', 'fp16', dgrad
        wd = ops.polyphase_dgrad_weights_lowres(Weff, C, C, s, kl)
        full, part = _both(lambda: ops.conv3_s2d(du, wd, C, B, G, G + 2 * R, -(kl - 1), s, C, poly_k=k))
        assert torch.equal(full, part)
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.DGRAD_PRECISION = state
