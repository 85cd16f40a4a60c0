"""GPU: direct-to-LDS fused attention forward (flash_fwd_dl.hip) against the register-staged forward kernel -- same
operand rounding, same dropout mask, same summation structure, so the two agree to fp32 rounding -- and, in bf16x3,
against float64 attention."""
import pytest
import torch

from voxactb_amd import flash
from tests.test_ops_gpu import rnd, close, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('x3', [False, True])
@pytest.mark.parametrize('B,H,Nq,Nk', [(2, 1, 100, 141), (1, 8, 256, 256), (2, 2, 77, 64), (1, 1, 300, 8077), (1, 2, 130, 65)])
def test_flash_fwd_dl_matches_staged_kernel(B, H, Nq, Nk, x3):
    q, kv = rnd(B * Nq, H * 64).to(DEV), rnd(B * Nk, 2 * H * 64, seed=1).to(DEV)
    for p, seed in ((0.0, 0), (0.2, 9)):
        o_ref, lse_ref = flash.flash_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, p=p, seed=seed, x3=x3)
        o, lse = flash.flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, 0.125, p=p, seed=seed, x3=x3)
        close(lse, lse_ref, 1e-6, 'lse p=%g' % p)
        close(o, o_ref, 2e-6, 'o p=%g' % p)


def test_flash_fwd_dl_x3_vs_fp64():
    B, H, Nq, Nk, scale = 1, 4, 200, 333, 0.125
    q, kv = rnd(B * Nq, H * 64), rnd(B * Nk, 2 * H * 64, seed=1)
    qh = q.double().view(B, Nq, H, 64).permute(0, 2, 1, 3)
    k = kv.double()[:, :H * 64].view(B, Nk, H, 64).permute(0, 2, 1, 3)
    v = kv.double()[:, H * 64:].view(B, Nk, H, 64).permute(0, 2, 1, 3)
    s = torch.einsum('bhid,bhjd->bhij', qh, k) * scale
    o_ref = torch.einsum('bhij,bhjd->bhid', s.softmax(-1), v).permute(0, 2, 1, 3).reshape(B * Nq, H * 64).float()
    o, lse = flash.flash_attn_fwd_dl(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, scale, x3=True)
    close(o, o_ref, 2e-5, 'o')
    close(lse, torch.logsumexp(s, -1).reshape(B * H, Nq).float(), 2e-5, 'lse')
