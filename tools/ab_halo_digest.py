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


if __name__ == '__main__':
    main()
