"""Micro-benchmark: the linear-layer GEMMs of the Perceiver stack (M = B * latents = 32768) in bf16x3 / bf16 --
register-staged kernel vs direct-to-LDS kernel (with and without weight fragments from global memory), the 256^2 kernel and, for
N = 512, the wide 128 x 512 kernel (|d| = largest difference from the register-staged result: 0 = bit-identical)."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops  # noqa: E402
from tools.bench_halo import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    M = 32768
    ops.set_wide_min_rows(1024)
    for x3 in (True,) if 'x3only' in sys.argv else (True, False):
        for N, K in [(4096, 512), (512, 4096), (2048, 512), (512, 2048), (512, 512), (1024, 512), (512, 1024)]:
            x = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev) * 0.05
            wb = ops.split_bf16(W, x3)
            out = torch.empty(M, N, device=dev)
            fl = 2.0 * M * N * K
            res = []
            ref = None
            for name, g256, dl, bd in (('staged', False, False, False), ('dl', False, 'force', False), ('dl+bfrag', False, 'force', True),
                                       ('gemm256', 'force', True, True), ('wide', False, False, True), ('wide4', False, False, True)):
                if name.startswith("wide") and not (x3 and N % 512 == 0 and K >= 256):
                    continue
                ops.GEMM256, ops.DL_GEMM, ops.GEMM_BD, ops.WIDE_GEMM = g256, dl, bd, name.startswith('wide')
                from voxactb_amd import _lib
                _lib.lib().vxb_debug_set_gemm_wide_waves(4 if name == 'wide4' else 8)

                ops.new_step()
                t = timeit(lambda: ops.gemm_bf16w(x, wb, out=out), n=10)
                if ref is None:
                    ref = out.clone()
                err = float((out - ref).abs().max())
                res.append('%s %.3f ms %5.0f TF/s (|d| %.1e)' % (name, t, fl / t * 1e-9, err))
            ops.GEMM256, ops.DL_GEMM, ops.GEMM_BD, ops.WIDE_GEMM = True, True, True, True
            print('%s  %5d x %5d x %5d   %s' % ('bf16x3' if x3 else 'bf16  ', M, N, K, '   '.join(res)))


if __name__ == '__main__':
    main()
