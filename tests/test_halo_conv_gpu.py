"""LDS-halo bf16 conv (conv_halo_bf16.hip) against the generic bf16 implicit-GEMM kernel and the fp32 kernel on
bf16-rounded operands: same products, fp32 accumulation in a different order -> 3e-5 relative."""
import pytest
import torch

from voxactb_amd import ops
from .test_ops_gpu import rnd, close, cl, bf, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('C0,C1,N,S', [(64, 0, 64, 16), (64, 64, 64, 19), (128, 0, 128, 17), (64, 0, 128, 21)])
def test_halo_forward_matches_generic(C0, C1, N, S):
    B = 2
    a = cl(rnd(B, C0, S, S, S)).to(DEV)
    c = cl(rnd(B, C1, S, S, S, seed=5)).to(DEV) if C1 else None
    W = rnd(N, C0 + C1, 3, 3, 3, seed=1, scale=0.1).to(DEV)
    b = rnd(N, seed=2).to(DEV)
    wb = ops.to_bf16_nk(ops.conv_weight_fwd(W))
    ops.HALO_CONV = False
    try:
        ref = ops.conv3d_bf16w(a, wb, N, B, S, S, 3, -1, bias=b, act=ops.ACT_LRELU, src1=c)
    finally:
        ops.HALO_CONV = True
    y = ops.conv3d_bf16w(a, wb, N, B, S, S, 3, -1, bias=b, act=ops.ACT_LRELU, src1=c)
    close(y, ref, 3e-5, 'halo fwd vs generic bf16')
    # and against the exact fp32 kernel on rounded operands
    ref32 = ops.conv3d(bf(a), ops.conv_weight_fwd(bf(W)), N, B, S, S, 3, -1, bias=b, act=ops.ACT_LRELU,
                       src1=bf(c) if C1 else None)
    close(y, ref32, 3e-5, 'halo fwd vs fp32 on rounded operands')


@pytest.mark.parametrize('Cin,Cout,S', [(128, 64, 16), (64, 64, 18)])
def test_halo_dgrad_matches_generic(Cin, Cout, S):
    """zero padding, S_out = S + 2, off = -2 (the data-gradient configuration)."""
    B = 2
    dy = cl(rnd(B, Cout, S, S, S, seed=3)).to(DEV)
    W = rnd(Cout, Cin, 3, 3, 3, seed=1, scale=0.1).to(DEV)
    wd = ops.to_bf16_nk(ops.conv_weight_dgrad(W))
    ops.HALO_CONV = False
    try:
        ref = ops.conv3d_bf16w(dy, wd, Cin, B, S, S + 2, 3, -2, replicate=False)
    finally:
        ops.HALO_CONV = True
    y = ops.conv3d_bf16w(dy, wd, Cin, B, S, S + 2, 3, -2, replicate=False)
    close(y, ref, 3e-5, 'halo dgrad vs generic bf16')


@pytest.mark.parametrize('C0,C1,N,S', [(64, 64, 64, 19), (32, 0, 128, 17), (64, 0, 64, 16)])
def test_halo_x3_matches_fp64_reference(C0, C1, N, S):
    """'bf16x3' halo kernel (forward, replicate padding, two sources) and its zero-padded data-gradient form against a
    float64 PyTorch conv: 2e-5 of the output max, the bound of the exact-fp32 kernels."""
    import torch.nn.functional as F
    from .test_ops_gpu import ref_conv
    B = 2
    a, c = rnd(B, C0, S, S, S), (rnd(B, C1, S, S, S, seed=5) if C1 else None)
    W = rnd(N, C0 + C1, 3, 3, 3, seed=1, scale=0.1)
    b = rnd(N, seed=2)
    xin = torch.cat([a, c], 1) if C1 else a
    ref = F.leaky_relu(ref_conv(xin.double(), W.double(), b.double()), 0.02).float()
    ops.PRECISION = 'bf16x3'
    try:
        y = ops.conv3d(cl(a).to(DEV), ops.conv_weight_fwd(W.to(DEV)), N, B, S, S, 3, -1, bias=b.to(DEV), act=ops.ACT_LRELU,
                       src1=cl(c).to(DEV) if C1 else None)
        close(y, cl(ref), 2e-5, 'x3 halo fwd')
        dy = rnd(B, N, S, S, S, seed=3)
        dxp = ops.conv3d(cl(dy).to(DEV), ops.conv_weight_dgrad(W.to(DEV)), C0 + C1, B, S, S + 2, 3, -2, replicate=False) \
            if (C0 + C1) in (64, 128) else None
    finally:
        ops.PRECISION = 'fp32'
    if dxp is not None:
        refd = F.conv_transpose3d(dy.double(), W.double()).float()      # full (padded-domain) data gradient
        close(dxp, cl(refd), 2e-5, 'x3 halo dgrad')
