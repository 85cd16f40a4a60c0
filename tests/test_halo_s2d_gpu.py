"""LDS-halo conv with space-to-depth input (data gradient of the polyphase up-conv) and depth-to-space output (its
forward), against the generic implicit-GEMM kernels on the same operands."""
import pytest
import torch

from voxactb_amd import ops
from .test_ops_gpu import rnd, close, cl, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode,tol', [('bf16', 3e-5), ('bf16x3', 2e-5)])
@pytest.mark.parametrize('s,G', [(2, 9), (5, 4)])
def test_s2d_dgrad_matches_strided_form(mode, tol, s, G):
    B, C, kl, R = 2, 64, 3, 1
    Weff = rnd(kl ** 3 * C, s ** 3 * C, seed=1, scale=0.05).to(DEV)
    du = cl(rnd(B, C, G * s, G * s, G * s, seed=3)).to(DEV)
    Sp = G + 2 * R
    ops.PRECISION = mode
    try:
        wd = ops.polyphase_dgrad_weights(Weff, C, C, s, kl)
        ops.HALO_CONV = False
        ref = ops.conv3d(du, wd, C, B, G * s, Sp, s * kl, -s * (kl - 1), stride=s, replicate=False)
        ops.HALO_CONV = True
        assert ops.s2d_halo_ok(kl, C, C)
        got = ops.conv3_s2d(du, ops.polyphase_dgrad_weights_lowres(Weff, C, C, s, kl), C, B, G, Sp, -(kl - 1), s, C)
    finally:
        ops.PRECISION = 'fp32'
        ops.HALO_CONV = True
    close(got, ref, tol if mode == 'bf16' else 3e-5, 's2d halo dgrad ' + mode)
    if mode == 'bf16x3':
        exact = ops.conv3d(du, wd, C, B, G * s, Sp, s * kl, -s * (kl - 1), stride=s, replicate=False)   # fp32 matrix cores
        # K = 27 * s^3 * 64 products per output (216,000 at s = 5): both results carry sqrt(K)-growing rounding noise
        close(got, exact, 2 * tol, 's2d halo dgrad x3 vs exact fp32')


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
def test_d2s_forward_matches_generic(mode):
    B, C, G, s = 2, 64, 17, 2
    z = cl(rnd(B, C, G, G, G)).to(DEV)
    Weff = rnd(27 * C, s ** 3 * 64, seed=1, scale=0.05).to(DEV)
    bias = rnd(s ** 3 * 64, seed=2).to(DEV)
    ops.PRECISION = mode
    try:
        ops.HALO_CONV = False
        ref = ops.conv3d(z, Weff, s ** 3 * 64, B, G, G, 3, -1, bias=bias, act=ops.ACT_LRELU, d2s=(s, 64))
        ops.HALO_CONV = ops.HALO_D2S = True
        got = ops.conv3d(z, Weff, s ** 3 * 64, B, G, G, 3, -1, bias=bias, act=ops.ACT_LRELU, d2s=(s, 64))
    finally:
        ops.PRECISION = 'fp32'
        ops.HALO_CONV, ops.HALO_D2S = True, False
    assert got.shape == (B, G * s, G * s, G * s, 64)
    close(got, ref, 3e-5, 'd2s halo fwd ' + mode)
