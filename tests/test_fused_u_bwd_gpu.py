"""vxb_conv3_c1_dgrad_ss3d_f32: the backward of everything that reads u = final(...) in one pass (pooled-feature term of
SpatialSoftmax3D + max, translation head's data gradient, LeakyReLU', bias column sums) against the three separate kernels
(each of which is checked against torch in test_ops_gpu.py / test_c1_conv_gpu.py): du bit-identical, column sums to 1e-5."""
import pytest
import torch

from voxactb_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


@pytest.mark.parametrize('B,S,two', [(2, 12, False), (1, 20, False), (2, 8, True)])
def test_fused_u_backward(B, S, two):
    C = 64
    u = (rnd(B, S, S, S, C, seed=1) * 0.05).to(DEV)
    dq = (rnd(B, S, S, S, seed=2) * 0.1).to(DEV)
    dql = (rnd(B, S, S, S, seed=3) * 0.1).to(DEV)
    w = (rnd(1, C, 3, 3, 3, seed=4) * 0.05).to(DEV)
    wl = (rnd(1, C, 3, 3, 3, seed=5) * 0.05).to(DEV)
    g_ss, g_mx = rnd(B, 3 * C, seed=6).to(DEV), rnd(B, C, seed=7).to(DEV)
    ss, mx, stats, arg = ops.ss3d_max_fwd(u, S ** 3 * C, B, S, C)
    # the separate kernels, in the order the engine used them
    du0 = torch.empty_like(u)
    ops.ss3d_max_bwd(u, S ** 3 * C, B, S, C, stats, ss, arg, g_ss, g_mx, du0, S ** 3 * C)
    if two:
        ops.conv3_c1_dgrad(dql, wl, u, du0, B, S, accumulate=True, mask=False)
    ops.conv3_c1_dgrad(dq, w, u, du0, B, S, accumulate=True, mask=True)
    db0 = torch.full((C,), 0.5, device=DEV)
    ops.colsum(du0.view(-1, C), db0, accumulate=True)
    # fused
    assert ops.c1_dgrad_ss3d_ok(S, C)
    du1 = torch.empty_like(u)
    if two:
        ops.conv3_c1_dgrad(dql, wl, u, du1, B, S, accumulate=False, mask=False)
    db1 = torch.full((C,), 0.5, device=DEV)
    ops.conv3_c1_dgrad_ss3d(dq, w, u, du1, B, S, stats, ss, arg, g_ss, g_mx, db1, accumulate=two)
    assert torch.equal(du0, du1)
    ref = du0.double().sum((0, 1, 2, 3)) + 0.5
    scale = float(ref.abs().max())
    e0 = float((db0.double() - ref).abs().max()) / scale
    e1 = float((db1.double() - ref).abs().max()) / scale
    print('bias column sums vs float64: colsum kernel %.2e, fused %.2e' % (e0, e1))
    assert e1 < 1e-5
