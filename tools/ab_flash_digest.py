"""Digest + time of the fused attention kernels at the step's self-attention size (B = 16, 8 heads, 2048 x 2048, dropout 0.1) -- run
with two builds (VOXACTB_HIP_LIB) and compare: scheduling changes must leave every output bit-identical."""
import hashlib
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import flash  # noqa: E402


def dg(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12]


def main():
    dev = 'cuda:0'
    torch.manual_seed(0)
    for (B, H, Nq, Nk) in ((2, 2, 200, 333), (16, 8, 2048, 2048)):
        q = torch.randn(B * Nq, H * 64, device=dev)
        kv = torch.randn(B * Nk, 2 * H * 64, device=dev)
        do = torch.randn(B * Nq, H * 64, device=dev)
        for p in (0.1, 0.0):
            o, lse, kvp = flash.flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, 0.125, p, 7, x3=True, return_planes=True)
            dq, dkv = flash.flash_attn_bwd_dl(q, kv, o, do, lse, B, H, Nq, Nk, 0.125, p, 7, x3=True, kv_planes=kvp)
            torch.cuda.synchronize()
            print('B%d H%d %dx%d p=%.1f  o %s lse %s dq %s dkv %s' % (B, H, Nq, Nk, p, dg(o), dg(lse), dg(dq), dg(dkv)))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(5):
        o, lse, kvp = flash.flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, 0.125, 0.1, 7, x3=True, return_planes=True)
    ev[1].record()
    for _ in range(5):
        flash.flash_attn_bwd_dl(q, kv, o, do, lse, B, H, Nq, Nk, 0.125, 0.1, 7, x3=True, kv_planes=kvp)
    ev[2].record()
    torch.cuda.synchronize()
    print('fwd (incl. kv split) %.3f ms   bwd (incl. q / dO split, row dots) %.3f ms' % (ev[0].elapsed_time(ev[1]) / 5, ev[1].elapsed_time(ev[2]) / 5))


if __name__ == '__main__':
    main()
