"""GPU: bf16 matrix-core weight gradient (ds_read_b64_tr_b16 transposed operands) == fp32 kernel on bf16-rounded operands."""
import pytest
import torch

from voxactb_amd import ops
from tests.test_ops_gpu import rnd, close, cl, bf, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('Cin,Cout,k,s,S,B', [(64, 64, 3, 1, 6, 2), (128, 64, 5, 1, 5, 2), (64, 64, 5, 5, 10, 3), (64, 128, 3, 1, 7, 1),
                                               (64, 64, 3, 2, 8, 2)])
def test_wgrad_bf16_matches_rounded_fp32(Cin, Cout, k, s, S, B):
    p = k // 2
    G = (S + 2 * p - k) // s + 1
    x = cl(rnd(B, Cin, S, S, S)).to(DEV)
    dy = cl(rnd(B, Cout, G, G, G, seed=3)).to(DEV)
    ref = ops.conv3d_wgrad(bf(x), bf(dy), Cout, B, S, G, k, -p, stride=s, nsplit=2)
    got = ops.conv3d_wgrad(x, dy, Cout, B, S, G, k, -p, stride=s, nsplit=3, force_bf16=True)
    close(got, ref, 3e-5, 'bf16 wgrad')


def test_wgrad_bf16_two_sources_and_d2s():
    B, S = 2, 5
    a, c = cl(rnd(B, 64, S, S, S)).to(DEV), cl(rnd(B, 64, S, S, S, seed=5)).to(DEV)
    dy = cl(rnd(B, 64, S, S, S, seed=7)).to(DEV)
    ref = ops.conv3d_wgrad(bf(a), bf(dy), 64, B, S, S, 3, -1, src1=bf(c), nsplit=1)
    got = ops.conv3d_wgrad(a, dy, 64, B, S, S, 3, -1, src1=c, nsplit=2, force_bf16=True)
    close(got, ref, 3e-5, 'two-source')
    # polyphase: dY is the fine grid of a depth-to-space output
    k, s, G = 5, 5, 3
    L, R = ops.polyphase_tables(k, s)
    kl = 2 * R + 1
    z1 = cl(rnd(B, 64, G, G, G, seed=9)).to(DEV)
    dyf = cl(rnd(B, 64, G * s, G * s, G * s, seed=11)).to(DEV)
    ref = ops.conv3d_wgrad(bf(z1), bf(dyf), s ** 3 * 64, B, G, G, kl, -R, d2s=(s, 64), nsplit=1)
    got = ops.conv3d_wgrad(z1, dyf, s ** 3 * 64, B, G, G, kl, -R, d2s=(s, 64), nsplit=2, force_bf16=True)
    close(got, ref, 3e-5, 'polyphase d2s wgrad')
