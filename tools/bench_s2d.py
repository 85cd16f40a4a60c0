"""The polyphase up-conv's data gradient (tap-list LDS-halo kernel) at the step's size, by split-K factor: time and the
difference to the unsplit launch."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops  # noqa: E402


def main():
    dev = 'cuda:0'
    torch.manual_seed(0)
    B, G, k, s, C = int(os.environ.get('S2D_B', 16)), 20, 5, 5, 64
    Lh, R = ops.polyphase_tables(k, s)
    kl = 2 * R + 1
    Weff = torch.randn(kl ** 3 * C, s ** 3 * C, device=dev) * 0.05
    du = torch.randn(B, G * s, G * s, G * s, C, device=dev)
    ops.PRECISION = 'bf16x3'
    wd = ops.polyphase_dgrad_weights_lowres(Weff, C, C, s, kl)
    ref = None
    for ks in (1, 2, 3, 4, 5, 6, 8, 12, 16):
        ops.S2D_KSPLIT = ks
        got = ops.conv3_s2d(du, wd, C, B, G, G + 2 * R, -(kl - 1), s, C, poly_k=k)
        torch.cuda.synchronize()
        if ref is None:
            ref = got
        err = float((got - ref).abs().max() / ref.abs().max())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ops.conv3_s2d(du, wd, C, B, G, G + 2 * R, -(kl - 1), s, C, poly_k=k)
        b.record()
        torch.cuda.synchronize()
        print('ksplit=%2d  %.3f ms  max |d| / max |ref| = %.2e' % (ks, a.elapsed_time(b) / 5, err))


if __name__ == '__main__':
    main()
