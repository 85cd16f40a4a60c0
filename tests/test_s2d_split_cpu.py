"""CPU: host logic of the split-K launch of the up-conv's data gradient (ops.s2d_kparts: chunk boundaries that give every part about the
same number of listed taps) and of the fragment orders of the two-product kernels (ops.halo_wfrag_x2)."""
import torch

from voxactb_amd import ops


def test_chunk_boundaries_balance_the_listed_taps():
    k, s, cpc, cpp = 5, 5, 16, 4
    dev = torch.device('cpu')
    tt, ncls, total, rows = ops.s2d_taptab(k, s, dev, cpc, cpp)
    taps = ops._POLY[('ttc', k, s, str(dev), cpc, cpp)]
    assert len(taps) == s ** 3 * cpp and sum(taps) == total == rows.numel()
    for ks in (1, 2, 3, 6, 8, 16):
        kp = ops.s2d_kparts(k, s, dev, cpc, cpp, ks).tolist()
        assert len(kp) == ks + 1 and kp[0] == 0 and kp[-1] == len(taps)
        assert all(a < b for a, b in zip(kp, kp[1:]))
        parts = [sum(taps[a:b]) for a, b in zip(kp, kp[1:])]
        assert sum(parts) == total
        assert max(parts) - min(parts) <= 2 * max(taps)          # within two chunks of each other


def test_single_plane_fragment_order():
    """halo_wfrag_x2: element (n, tap, channel) of the fp16 [N][27 Ct] matrix sits at [n / 64][channel / 16][tap][(n % 64) / 32]
    [lane = ((channel % 16) / 8) * 32 + n % 32][channel % 8]."""
    N, Ct = 128, 32
    w = torch.arange(N * 27 * Ct, dtype=torch.float32).view(N, 27 * Ct).half()
    f = ops.halo_wfrag_x2(w, Ct)
    assert tuple(f.shape) == (N // 64, Ct // 16, 27, 2, 2, 32, 8)
    for n, tap, c in ((0, 0, 0), (5, 3, 9), (70, 26, 31), (127, 13, 16)):
        got = f[n // 64, c // 16, tap, (n % 64) // 32, (c % 16) // 8, n % 32, c % 8]
        assert got == w[n, tap * Ct + c]
