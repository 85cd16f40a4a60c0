"""GPU: `final` conv with the SpatialSoftmax3D / max-pool statistics of its output taken in the conv's epilogue
(vxb_conv3_halo_ss3d_bf16x3_f32; perceiver_lang_io.py:462 + :470) against the two-kernel path (conv, then vxb_ss3d_max_fwd_f32 over
its output): the conv output must be bit-identical, the argmax identical, the pooled features equal up to the association of the
partial sums."""
import pytest
import torch

from voxactb_amd import ops
from .test_ops_gpu import rnd, cl, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('S,B', [(16, 2), (20, 2), (22, 1), (28, 1), (36, 1)])
def test_conv_epilogue_statistics_match_the_statistics_pass(S, B):
    C = 64
    d0 = cl(rnd(B, C, S, S, S, seed=1)).to(DEV)
    u0 = cl(rnd(B, C, S, S, S, seed=2)).to(DEV)
    W = rnd(C, 2 * C, 3, 3, 3, seed=3, scale=0.03).to(DEV)
    bias = rnd(C, seed=4).to(DEV)
    ops.PRECISION = 'bf16x3'
    wino, ops.FINAL_WINOGRAD = ops.FINAL_WINOGRAD, False        # the DIRECT kernel's claim (tests/test_halo_winograd_gpu.py: the other one)
    try:
        assert ops.conv3_ss3d_ok(C, C, C, S)
        wt = ops.conv_weight_fwd(W)
        ref = ops.conv3d(d0, wt, C, B, S, S, 3, -1, bias=bias, act=ops.ACT_LRELU, src1=u0)
        r_ss, r_max, r_stats, r_arg = ops.ss3d_max_fwd(ref, S ** 3 * C, B, S, C)
        got, (g_ss, g_max, g_stats, g_arg) = ops.conv3_ss3d_fwd(d0, u0, wt, bias, B, S)
    finally:
        ops.PRECISION = 'fp32'
        ops.FINAL_WINOGRAD = wino
    assert torch.equal(got, ref)
    assert torch.equal(g_arg, r_arg)
    assert torch.equal(g_max, r_max)
    assert torch.equal(g_stats[..., 0], r_stats[..., 0])                                     # the maxima of x / T are exact
    assert float((g_stats[..., 1] - r_stats[..., 1]).abs().max() / r_stats[..., 1].abs().max()) < 2e-6
    assert float((g_ss - r_ss).abs().max()) < 2e-6                                           # expected coordinates in [-1, 1]


def test_ties_go_to_the_lowest_voxel_index():
    """a constant output (zero weights, equal bias): every voxel ties for the maximum -> argmax 0, features = the grid's mean (0)."""
    S, B, C = 20, 1, 64
    d0 = torch.zeros(B, S, S, S, C, device=DEV)
    u0 = torch.zeros(B, S, S, S, C, device=DEV)
    W = torch.zeros(C, 2 * C, 3, 3, 3, device=DEV)
    bias = torch.full((C,), 0.25, device=DEV)
    ops.PRECISION = 'bf16x3'
    try:
        got, (g_ss, g_max, g_stats, g_arg) = ops.conv3_ss3d_fwd(d0, u0, ops.conv_weight_fwd(W), bias, B, S)
    finally:
        ops.PRECISION = 'fp32'
    assert torch.equal(got, torch.full_like(got, 0.25))
    assert int(g_arg.abs().max()) == 0 and torch.equal(g_max, torch.full_like(g_max, 0.25))
    assert float(g_ss.abs().max()) < 1e-5
