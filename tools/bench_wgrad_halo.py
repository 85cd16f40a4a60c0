"""Micro-benchmark: LDS-halo 3x3x3 weight gradient vs the generic transposed-read kernel at the `final` conv's size."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import ops  # noqa: E402
from tools.bench_halo import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    B, S = 4, 100
    x0 = torch.randn(B, S, S, S, 64, device=dev)
    x1 = torch.randn(B, S, S, S, 64, device=dev)
    dy = torch.randn(B, S, S, S, 64, device=dev)
    fl = 2.0 * B * S ** 3 * 64 * 27 * 128
    from voxactb_amd import _lib
    if os.environ.get('WH_NCH'):       # 16-channel chunks per workgroup: 1 or 2 (default: 2 where the channel count allows)
        _lib.lib().vxb_debug_set_wgrad_halo_chunks(int(os.environ['WH_NCH']))
    if os.environ.get('WH_DBG'):       # timing experiments (wrong results): 1 = stage / prefetch the first tile only, 2 = no MFMA loop
        _lib.lib().vxb_debug_set_wgrad_halo_experiment(int(os.environ['WH_DBG']))
    for mode in ('bf16', 'bf16x3'):
        for halo, shape in ((False, -1), (True, 0), (True, 1)):
            ops.HALO_CONV = halo
            _lib.lib().vxb_debug_set_wgrad_halo_shape(shape)
            t = timeit(lambda: ops.conv3d_wgrad(x0, dy, 64, B, S, S, 3, -1, src1=x1, force_bf16=mode), n=3)
            print('%-7s wgrad 128->64 S100  halo=%d tile=%s  %.3f ms  %.1f TF/s' % (mode, halo, ('2x8x8', '4x4x8', 'auto')[shape], t, fl / t * 1e-9))
    # polyphase up-conv weight gradient: low-res 20^3 x 64 -> 125 * 64 phase channels, dY on the 100^3 fine grid
    G, s = 20, 5
    z = torch.randn(B, G, G, G, 64, device=dev)
    dyf = torch.randn(B, G * s, G * s, G * s, 64, device=dev)
    fl = 2.0 * B * G ** 3 * 8000 * 27 * 64
    st = ops.polyphase_structure(5, 5, dev)
    for mode in ('bf16', 'bf16x3'):
        for halo, shape, sparse in ((False, -1, 0), (True, 0, 0), (True, 1, 0), (True, 0, 1), (True, 1, 1)):
            ops.HALO_CONV = halo
            _lib.lib().vxb_debug_set_wgrad_halo_shape(shape)
            t = timeit(lambda: ops.conv3d_wgrad(z, dyf, 8000, B, G, G, 3, -1, d2s=(s, 64), force_bf16=mode,
                                                phase_mask=st['phase_mask_t'] if sparse else None), n=3)
            print('%-7s wgrad 64->8000 S20 d2s  halo=%d tile=%s sparse=%d  %.3f ms  %.1f dense-equivalent TF/s' % (
                mode, halo, ('2x8x8', '4x4x8', 'auto')[shape], sparse, t, fl / t * 1e-9))
    _lib.lib().vxb_debug_set_wgrad_halo_shape(-1)


if __name__ == '__main__':
    main()
