"""Timing experiments of the wide linear-layer GEMM (gemm_wide.hip): which of its streams the main loop waits for.
Results of the ablated runs are WRONG by construction (vxb_debug_set_gemm_wide_experiment).  Needs a library built with
VXB_EXTRA_FLAGS=-DVXB_GW_ABLATE (the bits are compiled out of the shipped kernel: a branch around a load splits the loop's basic block
and costs ~10 % by itself); bit 32 (grid order) works in every build."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops, _lib  # noqa: E402
from tools.bench_halo import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    M = 32768
    ops.set_wide_min_rows(1024)
    L = _lib.lib()
    for waves in (8, 4):
        L.vxb_debug_set_gemm_wide_waves(waves)
        for N, K in [(4096, 512), (512, 4096), (512, 512)]:
            x = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev) * 0.05
            wb = ops.split_bf16(W, True)
            out = torch.empty(M, N, device=dev)
            fl = 2.0 * M * N * K
            res = []
            for bits in (0, 1, 2, 4, 8, 3, 7, 15, 32):
                L.vxb_debug_set_gemm_wide_experiment(bits)
                ops.new_step()
                t = timeit(lambda: ops.gemm_bf16w(x, wb, out=out), n=10)
                res.append('%d: %.3f' % (bits, t))
            L.vxb_debug_set_gemm_wide_experiment(0)
            print('waves %d  %5d x %5d   floor %.3f ms   %s' % (waves, N, K, fl * 3 / 2.5e12, '   '.join(res)), flush=True)


if __name__ == '__main__':
    main()
