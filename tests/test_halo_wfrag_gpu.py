"""GPU: the two weight paths of the LDS-halo conv -- B fragments straight from global memory (weights pre-shuffled into
fragment order by ops.halo_wfrag, no barrier in the tap loop) and weights staged through LDS per tap -- produce the
same sums in the same order: bit-identical outputs."""
import pytest
import torch

from voxactb_amd import ops
from .test_ops_gpu import rnd, cl, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('x3', [False, True])
@pytest.mark.parametrize('C0,C1,N,S', [(64, 64, 64, 19), (64, 0, 128, 17)])
def test_halo_weight_paths_agree_bitwise(C0, C1, N, S, x3):
    B = 2
    a = cl(rnd(B, C0, S, S, S)).to(DEV)
    c = cl(rnd(B, C1, S, S, S, seed=5)).to(DEV) if C1 else None
    W = rnd(N, C0 + C1, 3, 3, 3, seed=1, scale=0.1).to(DEV)
    wb = ops.split_bf16(ops.conv_weight_fwd(W).t().contiguous(), x3)
    outs = []
    for wd in (True, False):
        ops.HALO_WD = wd
        try:
            outs.append(ops.conv3d_bf16w(a, wb, N, B, S, S, 3, -1, act=ops.ACT_LRELU, src1=c))
        finally:
            ops.HALO_WD = True
    assert torch.equal(outs[0], outs[1])


def test_halo_wfrag_layout():
    """lane (lq, hi) of column tile j finds W[n = nb*64 + j*32 + lq][tap][chunk*CPC + ...] at its 16-byte slot."""
    N, Ct = 128, 64
    w = torch.arange(N * 27 * Ct, dtype=torch.float32).reshape(N, 27 * Ct)
    wb = (w % 251).to(torch.bfloat16).to(DEV)                      # small integers: exact in bf16
    wf = ops.halo_wfrag(wb, Ct).cpu().float()                      # (nb, ch, tap, j, f, hi, lq, e), chunk = 32 channels
    wv = (w % 251).reshape(N, 27, Ct)
    for nb, ch, tap, j, f, hi, lq in [(0, 0, 0, 0, 0, 0, 0), (1, 1, 26, 1, 1, 1, 31), (1, 0, 13, 0, 1, 0, 7)]:
        n = nb * 64 + j * 32 + lq
        k0 = ch * 32 + f * 16 + hi * 8
        assert torch.equal(wf[nb, ch, tap, j, f, hi, lq], wv[n, tap, k0:k0 + 8])
