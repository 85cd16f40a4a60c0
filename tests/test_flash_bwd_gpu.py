"""GPU: fused bf16 attention backward vs autograd of the fp32 attention on the same bf16-rounded operands."""
import pytest
import torch

from voxactb_amd import flash
from tests.test_ops_gpu import rnd, close, bf, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,H,Nq,Nk', [(2, 1, 100, 141), (1, 8, 128, 192), (1, 2, 77, 64), (1, 1, 200, 1000)])
def test_flash_bwd_matches_autograd(B, H, Nq, Nk):
    d, scale = 64, 0.125
    q = rnd(B * Nq, H * d)
    kv = rnd(B * Nk, 2 * H * d, seed=1)
    do = rnd(B * Nq, H * d, seed=2)
    qr = bf(q).requires_grad_(True)
    kvr = bf(kv).requires_grad_(True)
    qh = qr.view(B, Nq, H, d).permute(0, 2, 1, 3)
    k = kvr[:, :H * d].view(B, Nk, H, d).permute(0, 2, 1, 3)
    v = kvr[:, H * d:].view(B, Nk, H, d).permute(0, 2, 1, 3)
    p = (torch.einsum('bhid,bhjd->bhij', qh, k) * scale).softmax(-1)
    o_ref = torch.einsum('bhij,bhjd->bhid', p, v).permute(0, 2, 1, 3).reshape(B * Nq, H * d)
    (o_ref * bf(do)).sum().backward()
    o, lse = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, scale)
    dq, dkv = flash.flash_attn_bwd(q.to(DEV), kv.to(DEV), o, do.to(DEV), lse, B, H, Nq, Nk, scale)
    close(dq, qr.grad, 2e-2, 'dq')
    close(dkv[:, H * d:], kvr.grad[:, H * d:], 2e-2, 'dv')
    close(dkv[:, :H * d], kvr.grad[:, :H * d], 2e-2, 'dk')


def test_flash_bwd_dropout_consistent_with_forward():
    """O is linear in V for a fixed dropout mask: <dO, O(V + E) - O(V)> must equal <dV, E>."""
    B, H, Nq, Nk, scale = 1, 2, 96, 160, 0.125
    q, kv, do = rnd(B * Nq, H * 64), rnd(B * Nk, 2 * H * 64, seed=1), rnd(B * Nq, H * 64, seed=2)
    E = torch.zeros_like(kv)
    E[:, H * 64:] = bf(rnd(B * Nk, H * 64, seed=3))
    qd, kvd, dod = q.to(DEV), bf(kv).to(DEV), do.to(DEV)
    o0, lse = flash.flash_attn_fwd(qd, kvd, B, H, Nq, Nk, scale, p=0.3, seed=11)
    o1, _ = flash.flash_attn_fwd(qd, bf(bf(kv) + E).to(DEV), B, H, Nq, Nk, scale, p=0.3, seed=11)
    dq, dkv = flash.flash_attn_bwd(qd, kvd, o0, dod, lse, B, H, Nq, Nk, scale, p=0.3, seed=11)
    lhs = float(((o1 - o0).double() * dod.double()).sum())
    rhs = float((dkv.double().cpu() * E.double()).sum())
    assert abs(lhs - rhs) < 3e-2 * max(abs(lhs), 1.0), (lhs, rhs)
