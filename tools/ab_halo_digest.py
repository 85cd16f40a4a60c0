"""Digest of the LDS-halo conv outputs (forward, two concatenated sources, bf16x3 and bf16) at sizes whose last tiles are half
empty along h / w -- run with two builds (VOXACTB_HIP_LIB) and compare: the edge-tile variant must be bit-identical."""
import hashlib
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops  # noqa: E402


def main():
    dev = 'cuda:0'
    torch.manual_seed(0)
    for S, B in ((12, 2), (20, 2), (28, 1), (100, 1), (9, 1), (13, 1)):
        x0 = torch.randn(B, S, S, S, 64, device=dev)
        x1 = torch.randn(B, S, S, S, 64, device=dev)
        w = torch.randn(64, 27 * 128, device=dev) * 0.05
        bias = torch.randn(64, device=dev)
        for x3 in (True, False):
            wf = ops.split_bf16(w, x3)
            out = torch.full((B, S, S, S, 64), float('nan'), device=dev)
            ops.conv3d_bf16w(x0, wf, 64, B, S, S, 3, -1, bias=bias, act=ops.ACT_LRELU, src1=x1, out=out)
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            print('S=%d B=%d x3=%d %s' % (S, B, x3, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]))
    # data gradient + padding adjoint (S_out = S + 2: 22 and 102 end in a 6-wide tile, 20 in a 4-wide one, 18 in a 2-wide one)
    for S, B in ((20, 2), (18, 1), (16, 1), (100, 1)):
        dy = torch.randn(B, S, S, S, 64, device=dev)
        W = torch.randn(64, 128, 3, 3, 3, device=dev) * 0.1
        y1 = torch.randn(B, S, S, S, 64, device=dev)
        for mode in ('bf16x3', 'bf16'):
            ops.PRECISION = mode
            wd = ops.conv_weight_dgrad(W)
            g0 = torch.ones(B, S, S, S, 64, device=dev)
            g1 = torch.full((B, S, S, S, 64), float('nan'), device=dev)
            ops.conv3_dgrad_fold(dy, wd, B, S, 128, [(g0, True, None), (g1, False, y1)])
            torch.cuda.synchronize()
            assert torch.isfinite(g0).all() and torch.isfinite(g1).all()
            print('fold S=%d B=%d %s %s %s' % (S, B, mode, hashlib.sha256(g0.cpu().numpy().tobytes()).hexdigest()[:16],
                                               hashlib.sha256(g1.cpu().numpy().tobytes()).hexdigest()[:16]))
    # tap-list kernel: the polyphase up-conv's data gradient over the padded low-res grid (G + 2)
    for G, B in ((20, 1), (4, 2), (10, 1)):
        k, s, C = 5, 5, 64
        Lh, R = ops.polyphase_tables(k, s)
        kl = 2 * R + 1
        Weff = torch.randn(kl ** 3 * C, s ** 3 * C, device=dev) * 0.05
        du = torch.randn(B, G * s, G * s, G * s, C, device=dev)
        for mode in ('bf16x3', 'bf16'):
            ops.PRECISION = mode
            wd = ops.polyphase_dgrad_weights_lowres(Weff, C, C, s, kl)
            for pk in (None, k):
                got = ops.conv3_s2d(du, wd, C, B, G, G + 2 * R, -(kl - 1), s, C, poly_k=pk)
                torch.cuda.synchronize()
                assert torch.isfinite(got).all()
                print('s2d G=%d B=%d %s sparse=%d %s' % (G, B, mode, pk is not None, hashlib.sha256(got.cpu().numpy().tobytes()).hexdigest()[:16]))
    ops.PRECISION = 'fp32'


if __name__ == '__main__':
    main()
