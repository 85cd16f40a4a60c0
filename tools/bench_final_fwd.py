"""`final`'s forward (conv3_ss3d_fwd: 3x3x3 128 -> 64 at 100^3 with the SpatialSoftmax3D statistics in the epilogue), direct vs the
Winograd-along-depth variant (VOXACTB_FINAL_WINOGRAD).   python tools/bench_final_fwd.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops  # noqa: E402


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    B, S, dev = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 100, 'cuda:0'
    if os.environ.get('HALO_DBG'):      # timing experiments (results are wrong): vxb_debug_set_halo_experiment bits
        from voxactb_amd import _lib
        _lib.lib().vxb_debug_set_halo_experiment(int(os.environ['HALO_DBG']))
    d0 = torch.randn(B, S, S, S, 64, device=dev)
    u0 = torch.randn(B, S, S, S, 64, device=dev)
    wt = (torch.randn(27 * 128, 64, device=dev) * 0.05).contiguous()
    wt._vxb_keep = True
    bias = torch.randn(64, device=dev)
    ops.PRECISION = 'bf16x3'
    fl = 2.0 * B * S ** 3 * 64 * 27 * 128
    for wino in (False, True, False, True):
        ops.FINAL_WINOGRAD = wino
        t = timeit(lambda: ops.conv3_ss3d_fwd(d0, u0, wt, bias, B, S))
        print('B=%d winograd=%d  %.3f ms  %.1f TF/s (direct-conv flops)' % (B, wino, t, fl / t * 1e-9))


if __name__ == '__main__':
    main()
