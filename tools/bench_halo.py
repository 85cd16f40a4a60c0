"""Micro-benchmark: LDS-halo conv vs the generic implicit-GEMM kernel at the `final` conv's real size, bf16 and bf16x3."""
import sys
import os
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
    dev = 'cuda:0'
    if os.environ.get('HALO_WD') == '0':
        ops.HALO_WD = False
    if os.environ.get('HALO_WAVES'):
        from voxactb_amd import _lib
        _lib.lib().vxb_debug_set_halo_waves(int(os.environ['HALO_WAVES']))
    if os.environ.get('HALO_DBG'):      # timing experiments (results are wrong): 1 = stage the first chunk only, 2 = B fragments of 3 taps only
        from voxactb_amd import _lib
        _lib.lib().vxb_debug_set_halo_experiment(int(os.environ['HALO_DBG']))
    if os.environ.get('HALO_WN'):
        from voxactb_amd import _lib
        _lib.lib().vxb_debug_set_halo_wn(int(os.environ['HALO_WN']))
    B, S = int(os.environ.get('HALO_B', 4)), 100
    x0 = torch.randn(B, S, S, S, 64, device=dev)
    x1 = torch.randn(B, S, S, S, 64, device=dev)
    wf32 = torch.randn(64, 27 * 128, device=dev) * 0.05
    wd32 = torch.randn(128, 27 * 64, device=dev) * 0.05
    bias = torch.randn(64, device=dev)
    out_f = torch.empty(B, S, S, S, 64, device=dev)
    out_d = torch.empty(B, S + 2, S + 2, S + 2, 128, device=dev)
    fl_f = 2.0 * B * S ** 3 * 64 * 27 * 128
    fl_d = 2.0 * B * (S + 2) ** 3 * 128 * 27 * 64
    for x3 in (False, True):
        wf, wd = ops.split_bf16(wf32, x3), ops.split_bf16(wd32, x3)
        for halo in (False, True):
            ops.HALO_CONV = halo
            t = timeit(lambda: ops.conv3d_bf16w(x0, wf, 64, B, S, S, 3, -1, bias=bias, act=ops.ACT_LRELU, src1=x1, out=out_f))
            print('x3=%d fwd  128->64  halo=%d  %.3f ms  %.1f TF/s' % (x3, halo, t, fl_f / t * 1e-9))
            t = timeit(lambda: ops.conv3d_bf16w(x0, wd, 128, B, S, S + 2, 3, -2, replicate=False, out=out_d))
            print('x3=%d dgrad 64->128 halo=%d  %.3f ms  %.1f TF/s' % (x3, halo, t, fl_d / t * 1e-9))


if __name__ == '__main__':
    main()
