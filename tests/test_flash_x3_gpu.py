"""GPU: 'bf16x3' fused attention (forward + backward) against float64 attention / autograd on the UNROUNDED fp32 operands.
Every operand, the probabilities and the score gradients are carried as hi + lo bf16 halves, so the kernels are held to
the bound of the exact-fp32 kernels (2e-5 of the output max), not to bf16 rounding."""
import pytest
import torch

from voxactb_amd import flash
from tests.test_ops_gpu import rnd, close, DEV

pytestmark = pytest.mark.gpu
TOL = 2e-5


def ref64(q, kv, do, B, H, Nq, Nk, scale):
    d = 64
    qr = q.double().requires_grad_(True)
    kvr = kv.double().requires_grad_(True)
    qh = qr.view(B, Nq, H, d).permute(0, 2, 1, 3)
    k = kvr[:, :H * d].view(B, Nk, H, d).permute(0, 2, 1, 3)
    v = kvr[:, H * d:].view(B, Nk, H, d).permute(0, 2, 1, 3)
    s = torch.einsum('bhid,bhjd->bhij', qh, k) * scale
    o = torch.einsum('bhij,bhjd->bhid', s.softmax(-1), v).permute(0, 2, 1, 3).reshape(B * Nq, H * d)
    (o * do.double()).sum().backward()
    return o.detach().float(), torch.logsumexp(s, -1).reshape(B * H, Nq).detach().float(), qr.grad.float(), kvr.grad.float()


@pytest.mark.parametrize('B,H,Nq,Nk', [(2, 1, 100, 141), (1, 8, 256, 256), (2, 2, 77, 64), (1, 1, 300, 8077)])
def test_flash_x3_forward_and_backward(B, H, Nq, Nk):
    scale = 0.125
    q, kv, do = rnd(B * Nq, H * 64), rnd(B * Nk, 2 * H * 64, seed=1), rnd(B * Nq, H * 64, seed=2)
    o_ref, lse_ref, dq_ref, dkv_ref = ref64(q, kv, do, B, H, Nq, Nk, scale)
    o, lse = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, scale, x3=True)
    close(lse, lse_ref, TOL, 'lse')
    close(o, o_ref, TOL, 'o')
    dq, dkv = flash.flash_attn_bwd(q.to(DEV), kv.to(DEV), o, do.to(DEV), lse, B, H, Nq, Nk, scale, x3=True)
    close(dq, dq_ref, TOL, 'dq')
    close(dkv[:, H * 64:], dkv_ref[:, H * 64:], TOL, 'dv')
    close(dkv[:, :H * 64], dkv_ref[:, :H * 64], TOL, 'dk')


def test_flash_x3_dropout_mask_matches_bf16_kernels():
    """same (seed, p) -> same keep mask in both precisions: outputs differ only by bf16 rounding."""
    B, H, Nq, Nk = 1, 2, 128, 512
    q, kv = rnd(B * Nq, H * 64), rnd(B * Nk, 2 * H * 64, seed=1)
    o3, _ = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, 0.125, p=0.25, seed=5, x3=True)
    o1, _ = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, 0.125, p=0.25, seed=5)
    o3b, _ = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, 0.125, p=0.25, seed=5, x3=True)
    assert torch.equal(o3, o3b)
    close(o1, o3, 2e-2, 'bf16 vs x3 with one mask')
