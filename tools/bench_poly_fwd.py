"""The polyphase up-conv forward (conv_poly_wide_x3_kernel: 64 ch at 20^3 -> 125 phases x 64 ch on the 100^3 grid) at the step's size;
VOXACTB_WIDE_DBG=64 = the epilogue of rounds 4 - 5 (stores straight out of the accumulators).   python tools/bench_poly_fwd.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops  # noqa: E402
from tools.bench_halo import timeit  # noqa: E402


def main():
    B, G, k, s, C, dev = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 20, 5, 5, 64, 'cuda:0'
    ops.PRECISION = 'bf16x3'
    torch.manual_seed(0)
    st = ops.polyphase_structure(k, s, dev)
    kl = st['kl']
    z = torch.randn(B, G, G, G, C, device=dev)
    Weff = (torch.randn(kl ** 3 * C, s ** 3 * C, device=dev) * 0.05) * st['weight_mask'].to(dev) if 'weight_mask' in st else torch.randn(kl ** 3 * C, s ** 3 * C, device=dev) * 0.05
    bias = torch.randn(s ** 3 * C, device=dev)
    outs = []
    for it in range(2):
        ops.new_step()
        t = timeit(lambda: ops.conv3_polyphase_fwd(z, Weff, C, B, G, k, s, bias, act=ops.ACT_LRELU), n=5)
        print('B=%d polyphase forward  %.3f ms' % (B, t))
    out = ops.conv3_polyphase_fwd(z, Weff, C, B, G, k, s, bias, act=ops.ACT_LRELU)
    print('checksum %.9e' % float(out.double().sum()), 'abs %.9e' % float(out.double().abs().sum()))


if __name__ == '__main__':
    main()
