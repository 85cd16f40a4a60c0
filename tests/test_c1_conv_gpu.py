"""GPU: the Cout = 1 3x3x3 conv of the translation head (float4 kernels for S % 4 == 0, scalar kernels otherwise) against
autograd of the PyTorch conv with replicate padding; interior and face voxels, accumulate and LeakyReLU-mask flags."""
import pytest
import torch
import torch.nn.functional as F

from voxactb_amd import ops
from .test_ops_gpu import rnd, close, cl, ref_conv, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,S', [(2, 12), (1, 16), (3, 4), (2, 7), (1, 20)])
def test_c1_forward_dgrad_wgrad(B, S):
    u = rnd(B, 64, S, S, S, seed=4).requires_grad_(True)
    w1, b1 = rnd(1, 64, 3, 3, 3, seed=5, scale=0.1).requires_grad_(True), rnd(1, seed=6).requires_grad_(True)
    q_ref = ref_conv(u.double(), w1.double(), b1.double())
    ud = cl(u.detach()).to(DEV)
    q = ops.conv3_c1_fwd(ud, w1.detach().to(DEV), b1.detach().to(DEV), B, S)
    close(q, q_ref[:, 0].float(), 2e-5, 'c1 fwd')
    dq = rnd(B, S, S, S, seed=7)
    (q_ref[:, 0] * dq.double()).sum().backward()
    # accumulate onto an existing gradient, then the LeakyReLU' mask of the producer
    base = rnd(B, S, S, S, 64, seed=8)
    du = base.to(DEV).clone()
    ops.conv3_c1_dgrad(dq.to(DEV), w1.detach().to(DEV), ud, du, B, S, accumulate=True, mask=True)
    want = (cl(u.grad).float() + base) * torch.where(cl(u.detach()) > 0, 1.0, ops.LRELU_SLOPE)
    close(du, want, 2e-5, 'c1 dgrad accumulate+mask')
    du2 = torch.full((B, S, S, S, 64), 7.0, device=DEV)
    ops.conv3_c1_dgrad(dq.to(DEV), w1.detach().to(DEV), ud, du2, B, S, accumulate=False, mask=False)
    close(du2, cl(u.grad).float(), 2e-5, 'c1 dgrad overwrite')
    dw, dbb = torch.zeros(1, 64, 3, 3, 3, device=DEV), torch.zeros(1, device=DEV)
    ops.conv3_c1_wgrad(ud, dq.to(DEV), dw, dbb, B, S)
    close(dw, w1.grad.float(), 3e-5, 'c1 wgrad')
    close(dbb, b1.grad.float(), 3e-5, 'c1 db')
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(dbb)
    ops.conv3_c1_wgrad(ud, dq.to(DEV), dw2, db2, B, S)
    assert torch.equal(dw, dw2) and torch.equal(dbb, db2)        # deterministic split reduction


@pytest.mark.parametrize('mode', ['bf16x3', 'bf16'])
@pytest.mark.parametrize('B,S', [(2, 12), (1, 16), (3, 4), (2, 7), (1, 21), (1, 2)])
def test_c1_matrix_core_forward_and_wgrad(B, S, mode):
    """taps as the MFMA column dimension (c1_mfma.hip): forward P = u w then a 27-point gather; weight gradient u^T Bq with
    the clamped-tap adjoint folded into Bq.  bf16x3 products: same 2e-5 / 3e-5 bounds as the exact kernels."""
    u = rnd(B, 64, S, S, S, seed=4).requires_grad_(True)
    w1, b1 = rnd(1, 64, 3, 3, 3, seed=5, scale=0.1).requires_grad_(True), rnd(1, seed=6).requires_grad_(True)
    q_ref = ref_conv(u.double(), w1.double(), b1.double())
    dq = rnd(B, S, S, S, seed=7)
    (q_ref[:, 0] * dq.double()).sum().backward()
    ud = cl(u.detach()).to(DEV)
    ops.PRECISION = mode
    try:
        assert ops.C1_MFMA
        q = ops.conv3_c1_fwd(ud, w1.detach().to(DEV), b1.detach().to(DEV), B, S)
        dw, dbb = torch.zeros(1, 64, 3, 3, 3, device=DEV), torch.zeros(1, device=DEV)
        ops.C1_WGRAD_F16 = False
        ops.conv3_c1_wgrad(ud, dq.to(DEV), dw, dbb, B, S)
        dw2, db2 = torch.zeros_like(dw), torch.zeros_like(dbb)
        ops.conv3_c1_wgrad(ud, dq.to(DEV), dw2, db2, B, S)
        # the shipped arithmetic of the default precision from 2^19 voxels on (here: at every size): one fp16 product per term, dq scaled by
        # a device-side power of two
        ops.C1_WGRAD_F16, ops.WGRAD_PRECISION, ops.C1_WGRAD_F16_MIN_VOXELS = True, 'fp16', 0
        dw16, db16 = torch.zeros_like(dw), torch.zeros_like(dbb)
        ops.conv3_c1_wgrad(ud, dq.to(DEV), dw16, db16, B, S)
        dw16b = torch.zeros_like(dw)
        ops.conv3_c1_wgrad(ud, (dq * 1e-7).to(DEV), dw16b, torch.zeros_like(dbb), B, S)       # softmax - onehot far from a peak: ~1e-7
        ops.C1_MFMA = False
        q_exact = ops.conv3_c1_fwd(ud, w1.detach().to(DEV), b1.detach().to(DEV), B, S)
    finally:
        ops.PRECISION, ops.C1_MFMA, ops.C1_WGRAD_F16, ops.WGRAD_PRECISION, ops.C1_WGRAD_F16_MIN_VOXELS = 'fp32', True, True, '', 1 << 19
    if mode == 'bf16x3':
        close(dw16, w1.grad.float(), 1e-3, 'c1 fp16 wgrad')
        close(dw16b * 1e7, w1.grad.float(), 1e-3, 'c1 fp16 wgrad, tiny dq')
        close(db16, b1.grad.float(), 3e-5, 'c1 fp16 db')
        assert not torch.equal(dw16, dw)
    close(q, q_ref[:, 0].float(), 2e-5, 'c1 mfma fwd')
    close(q, q_exact, 2e-5, 'c1 mfma fwd vs exact kernel')
    close(dw, w1.grad.float(), 3e-5, 'c1 mfma wgrad')
    close(dbb, b1.grad.float(), 3e-5, 'c1 mfma db')
    assert torch.equal(dw, dw2) and torch.equal(dbb, db2)
