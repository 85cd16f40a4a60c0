"""Micro-benchmark: weight gradients of the linear layers (plain-GEMM form of the generic weight-gradient kernel, single fp16
products) at the step's shapes, pipelined kernel vs the generic one (WGRAD_LIN=0), and the two 5^3 conv weight gradients.

    python tools/bench_wgrad_lin.py            # env WGRAD_LIN=0: generic kernel
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops, _lib  # noqa: E402


def timeit(fn, n=10):
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
    ops.PRECISION = 'bf16x3'
    ops.WGRAD_PRECISION = 'fp16'
    ops.GENERIC_WGRAD_F16 = True
    if os.environ.get('WGRAD_LIN') is not None:
        _lib.lib().vxb_debug_set_wgrad_lin(int(os.environ['WGRAD_LIN']))
    M = 32768
    tot = 0.0
    for N, K in ((4096, 512), (512, 2048), (512, 512), (1024, 512), (64, 512), (512, 64)):
        x = torch.randn(M, K, device=dev)
        dy = torch.randn(M, N, device=dev) * 1e-3
        W = torch.randn(N, K, device=dev)
        dW = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        ops._GRAD_SCALE = {}
        t = timeit(lambda: (ops.begin_backward(), ops.linear_bwd(x, W, dy, dW, db)))
        # reference on the first call's operands
        dW.zero_(); db.zero_()
        ops.begin_backward()
        ops.linear_bwd(x, W, dy, dW, db)
        ref = dy.double().t() @ x.double()
        err = float((dW.double() - ref).abs().max() / ref.abs().max())
        print('linear_bwd (dW + db) M=%d N=%d K=%d  %.3f ms  %.1f TF/s  rel err %.2e' % (M, N, K, t, 2.0 * M * N * K / t * 1e-9, err))
        tot += t
    print('sum %.3f ms' % tot)
    # the two 5^3 conv weight gradients of the step (B = 16): up0.conv_up.0 (20^3, 128 -> 64) and patchify (100^3 -> 20^3, stride 5)
    B = int(os.environ.get('B', 16))
    for (Cin, Cout, S_in, S_out, k, stride, off) in ((128, 64, 20, 20, 5, 1, -2), (64, 64, 100, 20, 5, 5, -2)):
        x = torch.randn(B, S_in, S_in, S_in, Cin, device=dev)
        dy = torch.randn(B, S_out, S_out, S_out, Cout, device=dev) * 1e-3
        ops._GRAD_SCALE = {}
        t = timeit(lambda: (ops.begin_backward(), ops.conv3d_wgrad(x, dy, Cout, B, S_in, S_out, k, off, stride=stride, grad_key=('conv', k * 1000 + stride))), n=5)
        print('conv3d_wgrad k%d s%d %d->%d S%d  %.3f ms  %.1f TF/s' % (k, stride, Cin, Cout, S_out, t,
                                                                      2.0 * B * S_out ** 3 * Cout * k ** 3 * Cin / t * 1e-9))


if __name__ == '__main__':
    main()
