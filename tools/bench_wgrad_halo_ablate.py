"""Timing experiments of the fp16 LDS-halo weight gradient (wgrad_halo.hip, the `final` conv's shape): as shipped, with the staging /
prefetch of every tile but the first removed (WRONG results: the ceiling of its MFMA loop), with the MFMA loop removed (staging only)."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops, _lib  # noqa: E402
from tools.bench_halo import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    B, S = int(os.environ.get('WH_B', 8)), 100
    x0 = torch.randn(B, S, S, S, 64, device=dev)
    x1 = torch.randn(B, S, S, S, 64, device=dev)
    dy = torch.randn(B, S, S, S, 64, device=dev) * 1e-3
    fl = 2.0 * B * S ** 3 * 64 * 27 * 128
    L = _lib.lib()
    ops.WGRAD_PRECISION = 'fp16'
    more = ((2 | 16, 'no MFMA, no loads after the prologue'), (2 | 4, 'no MFMA, no x conversion'), (2 | 8, 'no MFMA, no dY conversion'),
            (2 | 4 | 8, 'no MFMA, no conversion'), (2 | 4 | 8 | 16, 'no MFMA, no conversion, no loads'), (16, 'no loads after the prologue'),
            (4 | 8, 'no conversion'), (4 | 8 | 16, 'MFMA loop + barriers only')) if os.environ.get('WH_MORE') else ()
    for nch in (2, 1):
        L.vxb_debug_set_wgrad_halo_chunks(nch)
        for bits, what in ((0, 'as shipped'), (1, 'first tile staged only'), (2, 'no MFMA loop')) + (more if nch == 2 else ()):
            L.vxb_debug_set_wgrad_halo_experiment(bits)
            ops.new_step()
            t = timeit(lambda: ops.conv3d_wgrad(x0, dy, 64, B, S, S, 3, -1, src1=x1, force_bf16='bf16x3'), n=5)
            print('fp16 wgrad 128->64 S100 B=%d  chunks/workgroup %d  %-40s %.3f ms  %.1f TF/s' % (B, nch, what, t, fl / t * 1e-9), flush=True)
    L.vxb_debug_set_wgrad_halo_experiment(0)


if __name__ == '__main__':
    main()
