"""Micro-benchmark of the fused-attention forward kernels at the model's three shapes (B = 16)."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import flash  # noqa: E402
from tools.bench_halo import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    B = 16
    for name, H, Nq, Nk in (('self', 8, 2048, 2048), ('cross', 1, 2048, 8077), ('decoder', 1, 8077, 2048)):
        q = torch.randn(B * Nq, H * 64, device=dev)
        kv = torch.randn(B * Nk, 2 * H * 64, device=dev)
        fl = 4.0 * B * H * Nq * Nk * 64
        for x3 in (False, True):
            for p in (0.0, 0.1):
                t0 = timeit(lambda: flash.flash_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, p=p, seed=3, x3=x3), n=5)
                t1 = timeit(lambda: flash.flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, 0.125, p=p, seed=3, x3=x3), n=5)
                print('%-8s x3=%d p=%.1f  staged %.3f ms %.1f TF/s | direct-to-LDS (incl. split) %.3f ms %.1f TF/s' % (
                    name, x3, p, t0, fl / t0 * 1e-9, t1, fl / t1 * 1e-9))


if __name__ == '__main__':
    main()
