"""GPU: fused bf16 attention vs the plain PyTorch fp32 attention of the same (bf16-rounded) operands."""
import pytest
import torch

from voxactb_amd import flash
from tests.test_ops_gpu import rnd, close, bf, DEV

pytestmark = pytest.mark.gpu


def ref_attn(q, kv, B, H, Nq, Nk, scale):
    d = 64
    qh = bf(q * (scale * 1.4426950408889634)).view(B, Nq, H, d).permute(0, 2, 1, 3) / 1.4426950408889634
    k = bf(kv[:, :H * d]).view(B, Nk, H, d).permute(0, 2, 1, 3)
    v = bf(kv[:, H * d:]).view(B, Nk, H, d).permute(0, 2, 1, 3)
    s = torch.einsum('bhid,bhjd->bhij', qh, k)
    p = s.softmax(-1)
    o = torch.einsum('bhij,bhjd->bhid', p, v).permute(0, 2, 1, 3).reshape(B * Nq, H * d)
    return o, torch.logsumexp(s, -1).reshape(B * H, Nq), p


@pytest.mark.parametrize('B,H,Nq,Nk', [(2, 1, 100, 141), (1, 8, 256, 256), (2, 2, 77, 64), (1, 1, 300, 8077)])
def test_flash_fwd_matches_reference(B, H, Nq, Nk):
    q = rnd(B * Nq, H * 64)
    kv = rnd(B * Nk, 2 * H * 64, seed=1)
    o, lse = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, 0.125)
    o_ref, lse_ref, _ = ref_attn(q, kv, B, H, Nq, Nk, 0.125)
    close(lse, lse_ref, 2e-3, 'lse')          # scores carry bf16-rounded operands in both; P is rounded to bf16 in the kernel
    close(o, o_ref, 1e-2, 'o')


def test_flash_fwd_dropout_statistics_and_determinism():
    B, H, Nq, Nk = 1, 2, 128, 512
    q, kv = rnd(B * Nq, H * 64), rnd(B * Nk, 2 * H * 64, seed=1)
    # constant V = 1 in the first column: O[:, 0] = sum of kept probabilities / (1-p) -> mean ~ 1, spread small
    kv[:, H * 64:] = 0.0
    kv[:, H * 64] = 1.0
    kv[:, H * 64 + 64] = 1.0
    o1, _ = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, 0.125, p=0.25, seed=5)
    o2, _ = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, 0.125, p=0.25, seed=5)
    o3, _ = flash.flash_attn_fwd(q.to(DEV), kv.to(DEV), B, H, Nq, Nk, 0.125, p=0.25, seed=6)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    col = torch.cat([o1[:, 0], o1[:, 64]])
    assert abs(float(col.mean()) - 1.0) < 0.03, float(col.mean())
    assert 0.02 < float(col.std()) < 0.5
