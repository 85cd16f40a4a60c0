"""'bf16x3' split-product kernels (hi = bf16(a), lo = bf16(a - hi); hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32
accumulate) against plain PyTorch fp32 references of the same op.  Per-product error <= ~3 * 2^-18 |a||b|, so the
results are held to 2e-5 of the output's max -- the same bound the exact-fp32 kernels are tested with."""
import pytest
import torch
import torch.nn.functional as F

from voxactb_amd import ops
from .test_ops_gpu import rnd, close, cl, ref_conv, DEV

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(autouse=True)
def _x3_mode():
    ops.PRECISION = 'bf16x3'
    ops.new_step()
    yield
    ops.PRECISION = 'fp32'
    ops.new_step()


@pytest.mark.parametrize('M,N,K', [(256, 128, 64), (300, 72, 96), (1000, 512, 2048), (77, 64, 512)])
def test_gemm_x3(M, N, K):
    x, W, b, r = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    y = ops.gemm_bf16w(x.to(DEV), ops.split_bf16(W.to(DEV)), bias=b.to(DEV), act=ops.ACT_LRELU, residual=r.to(DEV))
    close(y, F.leaky_relu(x.double() @ W.double().t() + b.double(), 0.02).float() + r, TOL, 'x3 gemm')


def test_linear_and_linear_bwd_x3():
    M, N, K = 2048, 256, 512
    x, W, dy = rnd(M, K), rnd(N, K, seed=1, scale=0.1), rnd(M, N, seed=2)
    y = ops.linear(x.to(DEV), W.to(DEV))
    close(y, (x.double() @ W.double().t()).float(), TOL, 'x3 linear')
    dW = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    dx = torch.empty(M, K, device=DEV)
    ops.linear_bwd(x.to(DEV), W.to(DEV), dy.to(DEV), dW, db, dx)
    close(dx, (dy.double() @ W.double()).float(), TOL, 'x3 dx')
    close(dW, (dy.double().t() @ x.double()).float(), TOL, 'x3 dW')


@pytest.mark.parametrize('Cin,Cout,k,s,S', [(64, 64, 3, 1, 6), (128, 64, 5, 1, 5), (64, 64, 5, 5, 10), (32, 128, 3, 1, 4)])
def test_conv3d_x3(Cin, Cout, k, s, S):
    B = 2
    x = rnd(B, Cin, S, S, S)
    W = rnd(Cout, Cin, k, k, k, seed=1, scale=0.1)
    b = rnd(Cout, seed=2)
    ref = F.leaky_relu(ref_conv(x.double(), W.double(), b.double(), s), 0.02).float()
    G = ref.shape[-1]
    y = ops.conv3d(cl(x).to(DEV), ops.conv_weight_fwd(W.to(DEV)), Cout, B, S, G, k, -(k // 2), stride=s, bias=b.to(DEV),
                   act=ops.ACT_LRELU)
    close(y, cl(ref), TOL, 'x3 conv fwd')
    dy = rnd(B, Cout, G, G, G, seed=3)
    dWt = ops.conv3d_wgrad(cl(x).to(DEV), cl(dy).to(DEV), Cout, B, S, G, k, -(k // 2), stride=s)
    xr = x.double().requires_grad_(False)
    Wd = W.double().requires_grad_(True)
    ref_conv(xr, Wd, None, s).backward(dy.double())
    close(dWt, ops.conv_weight_fwd(Wd.grad.float()), TOL, 'x3 conv wgrad')
