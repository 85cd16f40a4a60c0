"""GPU: direct-to-LDS GEMM / conv kernels (gemm_dl.hip: both operands as bf16 planes, global_load_lds, XOR-swizzled
tiles) against the register-staged kernels on the same operands -- identical products, different fp32 summation order
(3e-5) -- and, in bf16x3, against float64 references at the bound of the exact-fp32 kernels (2e-5)."""
import pytest
import torch
import torch.nn.functional as F

from voxactb_amd import ops
from .test_ops_gpu import rnd, close, cl, ref_conv, DEV

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _wide_kernels_at_test_sizes():
    """the wide 128 x 512-tile kernels are dispatched from 16384 rows on (ops.WIDE_MIN_M: below that they leave most CUs idle); these
    tests exercise them at a few thousand rows"""
    ops.set_wide_min_rows(1024)
    yield
    ops.set_wide_min_rows(16384)


def _both(fn):
    ops.DL_GEMM = False
    try:
        ref = fn()
    finally:
        ops.DL_GEMM = 'force'
    try:
        return ref, fn()
    finally:
        ops.DL_GEMM = True


@pytest.mark.parametrize('x3', [False, True])
@pytest.mark.parametrize('M,N,K', [(256, 128, 64), (300, 72, 96), (1000, 512, 2048), (131, 64, 512), (4096, 1000, 32)])
def test_gemm_dl_matches_register_staged(M, N, K, x3):
    x, W, b, r = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    wb = ops.split_bf16(W.to(DEV), x3)
    ref, got = _both(lambda: ops.gemm_bf16w(x.to(DEV), wb, bias=b.to(DEV), act=ops.ACT_LRELU, residual=r.to(DEV)))
    close(got, ref, 3e-5, 'dl gemm vs staged')
    if x3:
        close(got, F.leaky_relu(x.double() @ W.double().t() + b.double(), 0.02).float() + r, 2e-5, 'dl gemm x3 vs fp64')
    # accumulate into an existing output, strided input rows
    xx = rnd(M, K + 32, seed=5).to(DEV)
    base = rnd(M, N, seed=6).to(DEV)
    o1, o2 = base.clone(), base.clone()
    ops.DL_GEMM = False
    ops.gemm_bf16w(xx[:, :K], wb, out=o1, accumulate=True)
    ops.DL_GEMM = 'force'
    ops.gemm_bf16w(xx[:, :K], wb, out=o2, accumulate=True)
    ops.DL_GEMM = True
    close(o2, o1, 3e-5, 'dl gemm accumulate / strided rows')


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('Cin,Cout,k,s,S', [(64, 64, 5, 1, 6), (64, 64, 5, 5, 10), (32, 128, 3, 1, 5), (64, 64, 3, 2, 8)])
def test_conv_dl_matches_register_staged(mode, Cin, Cout, k, s, S):
    B = 2
    x = cl(rnd(B, Cin, S, S, S)).to(DEV)
    W = rnd(Cout, Cin, k, k, k, seed=1, scale=0.1).to(DEV)
    b = rnd(Cout, seed=2).to(DEV)
    G = (S + 2 * (k // 2) - k) // s + 1
    ops.PRECISION = mode
    ops.HALO_CONV = False           # isolate the generic paths
    try:
        ref, got = _both(lambda: ops.conv3d(x, ops.conv_weight_fwd(W), Cout, B, S, G, k, -(k // 2), stride=s, bias=b, act=ops.ACT_LRELU))
        close(got, ref, 3e-5, 'dl conv fwd ' + mode)
        if s == 1:      # zero-padded data-gradient geometry: taps outside the cube fetch the zero page
            dy = cl(rnd(B, Cout, G, G, G, seed=3)).to(DEV)
            p = k // 2
            ref, got = _both(lambda: ops.conv3d(dy, ops.conv_weight_dgrad(W), Cin, B, G, S + 2 * p, k, -(k - 1), replicate=False))
            close(got, ref, 3e-5, 'dl conv dgrad ' + mode)
    finally:
        ops.PRECISION = 'fp32'
        ops.HALO_CONV = True


def test_conv_dl_depth_to_space_polyphase_forward():
    B, C, G, s = 2, 64, 6, 5
    z = cl(rnd(B, C, G, G, G)).to(DEV)
    Weff = rnd(27 * C, s ** 3 * 64, seed=1, scale=0.05).to(DEV)
    bias = rnd(s ** 3 * 64, seed=2).to(DEV)
    ops.PRECISION = 'bf16x3'
    try:
        ref, got = _both(lambda: ops.conv3d(z, Weff, s ** 3 * 64, B, G, G, 3, -1, bias=bias, act=ops.ACT_LRELU, d2s=(s, 64)))
    finally:
        ops.PRECISION = 'fp32'
    assert got.shape == (B, G * s, G * s, G * s, 64)
    close(got, ref, 3e-5, 'dl conv d2s')
    exact = ops.conv3d(z, Weff, s ** 3 * 64, B, G, G, 3, -1, bias=bias, act=ops.ACT_LRELU, d2s=(s, 64))     # fp32 matrix cores
    close(got, exact, 2e-5, 'dl conv d2s x3 vs exact fp32')


@pytest.mark.parametrize('x3', [False, True])
@pytest.mark.parametrize('M,N,K', [(300, 72, 96), (1000, 512, 2048), (4096, 1000, 32), (256, 128, 64)])
def test_weight_fragments_from_global_are_bit_identical(M, N, K, x3):
    """B fragments read straight from global memory (ops.gemm_wfrag order, one k-tile ahead in registers) or through the
    LDS tile: same products, same order."""
    x, W, b = rnd(M, K).to(DEV), rnd(N, K, seed=1).to(DEV), rnd(N, seed=2).to(DEV)
    wb = ops.split_bf16(W, x3)
    outs = []
    ops.DL_GEMM = 'force'
    try:
        for bd in (False, True):
            ops.GEMM_BD = bd
            ops.new_step()
            outs.append(ops.gemm_bf16w(x, wb, bias=b, act=ops.ACT_LRELU))
    finally:
        ops.DL_GEMM, ops.GEMM_BD = True, True
    assert torch.equal(outs[0], outs[1])
    f = ops.gemm_wfrag(wb)
    Np = (N + 127) // 128 * 128
    assert f.shape == (Np // 32, K // 16, 2 if x3 else 1, 2, 32, 8)
    w3 = wb if x3 else wb.unsqueeze(0)
    assert torch.equal(f[1, 0, 0, 1, 5], w3[0, 32 + 5, 8:16])


@pytest.mark.parametrize('M,N,K', [(2048, 256, 64), (4096, 4096, 512), (2300, 512, 96), (2048, 768, 32)])
def test_gemm256_matches_float64_and_the_128_tile_kernels(M, N, K):
    """gemm256.hip (256 x 256 tiles, 8 waves) in bf16x3: ragged M, every epilogue option, against float64 (2e-5, the bound of
    the exact-fp32 kernels) and against the register-staged kernel (same products, same k order)."""
    x, W, b, r = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    wb = ops.split_bf16(W.to(DEV), True)
    keep = ops.GEMM256
    try:
        ops.GEMM256 = False
        ref = ops.gemm_bf16w(x.to(DEV), wb, bias=b.to(DEV), act=ops.ACT_LRELU, residual=r.to(DEV))
        ops.GEMM256 = 'force'
        got = ops.gemm_bf16w(x.to(DEV), wb, bias=b.to(DEV), act=ops.ACT_LRELU, residual=r.to(DEV))
        base = rnd(M, N, seed=6).to(DEV)
        acc = base.clone()
        ops.gemm_bf16w(x.to(DEV), wb, out=acc, accumulate=True)
    finally:
        ops.GEMM256 = keep
    close(got, ref, 3e-5, 'gemm256 vs 128-tile kernel')
    close(got, F.leaky_relu(x.double() @ W.double().t() + b.double(), 0.02).float() + r, 2e-5, 'gemm256 vs fp64')
    close(acc, (base.cpu().double() + x.double() @ W.double().t()).float(), 2e-5, 'gemm256 accumulate')


@pytest.mark.parametrize('precision', ['bf16x3', 'bf16'])
def test_batched_weight_split_is_bit_identical(precision):
    """vxb_split_bf16_batch_f32 (all linear weights of a step, plain + transposed, one launch) == the per-weight split."""
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(n, k, generator=g).to(DEV) * 0.05 for n, k in ((512, 512), (1024, 512), (64, 128), (72, 200), (4096, 512))]
    old = ops.PRECISION
    ops.PRECISION = precision
    try:
        ops.new_step()
        ref = [(ops.split_bf16(w.contiguous()), ops.split_bf16(w.t().contiguous())) for w in ws]
        ops.prepare_linear_weights(ws)
        for w, (r0, r1) in zip(ws, ref):
            a0, a1 = ops._bf16_weight(w, False), ops._bf16_weight(w, True)
            assert a0.shape == r0.shape and a1.shape == r1.shape
            assert torch.equal(a0.view(torch.int16), r0.view(torch.int16))
            assert torch.equal(a1.view(torch.int16), r1.view(torch.int16))
        # second step with changed weights: same buffers, new contents
        for w in ws:
            w.mul_(1.5)
        ops.new_step()
        ops.prepare_linear_weights(ws)
        assert torch.equal(ops._bf16_weight(ws[3], True).view(torch.int16), ops.split_bf16(ws[3].t().contiguous()).view(torch.int16))
    finally:
        ops.PRECISION = old
        ops.new_step()


def test_batched_weight_split_emits_fragment_order_for_the_wide_gemm():
    """the same launch writes the MFMA fragment order of every (weight, orientation) the wide GEMM takes (output width % 512 == 0,
    K % 32 == 0, K >= 256): bit-identical to the shuffling copy ops.gemm_wfrag would make, found by gemm_wfrag through the cache."""
    g = torch.Generator().manual_seed(6)
    ws = [torch.randn(n, k, generator=g).to(DEV) * 0.05 for n, k in ((512, 512), (1024, 512), (64, 128), (4096, 512), (512, 2048))]
    old = ops.PRECISION
    ops.PRECISION = 'bf16x3'
    try:
        ops.new_step()
        ops.prepare_linear_weights(ws)
        hits = 0
        for w in ws:
            for tr in (False, True):
                wb = ops._bf16_weight(w, tr)
                n_o, k_o = wb.shape[-2:]
                cached = (wb.data_ptr(), tuple(wb.shape)) in ops._FCACHE
                assert cached == (n_o % 512 == 0 and k_o % 32 == 0 and k_o >= 256), (n_o, k_o)
                f = ops.gemm_wfrag(wb)
                want = wb.view(2, n_o // 32, 32, k_o // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous() if n_o % 128 == 0 else None
                if want is not None:
                    assert f.shape == want.shape and torch.equal(f.view(torch.int16), want.view(torch.int16))
                hits += int(cached)
        assert hits == 8            # both orientations of 512x512, 1024x512, 4096x512 and 512x2048; none of 64x128
    finally:
        ops.PRECISION = old
        ops.new_step()


@pytest.mark.parametrize('waves,grid_bits', [(8, 0), (8, 32), (4, 0), (8, 64)])
@pytest.mark.parametrize('M,N,K', [(2048, 512, 256), (4096, 1024, 512), (2300, 512, 2048), (1024, 4096, 512)])
def test_gemm_wide_is_bit_identical_to_the_register_staged_kernel(M, N, K, waves, grid_bits):
    """gemm_wide.hip (128 x 512 workgroup tiles, A three k-tiles ahead, weight fragments from global memory) in bf16x3: ragged M, every
    epilogue option; same products in the same order as the kernels it replaces -> equal bits; and against float64.  waves = 4: the
    128 x 256 workgroups of four waves (two per CU); grid_bits = 32: round 3's grid order (row blocks fastest) instead of the column
    groups of a row block side by side on one XCD; 64: the epilogue of rounds 3 - 5 (stores straight out of the accumulators) instead of
    round 6's row-contiguous one through LDS."""
    from voxactb_amd import _lib
    x, W, b, r = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    wb = ops.split_bf16(W.to(DEV), True)
    keep = ops.WIDE_GEMM, ops.GEMM256, ops.DL_GEMM
    _lib.lib().vxb_debug_set_gemm_wide_waves(waves)
    _lib.lib().vxb_debug_set_gemm_wide_experiment(grid_bits)
    try:
        ops.WIDE_GEMM, ops.GEMM256, ops.DL_GEMM = False, False, False
        ref = ops.gemm_bf16w(x.to(DEV), wb, bias=b.to(DEV), act=ops.ACT_LRELU, residual=r.to(DEV))
        ops.WIDE_GEMM = True
        got = ops.gemm_bf16w(x.to(DEV), wb, bias=b.to(DEV), act=ops.ACT_LRELU, residual=r.to(DEV))
        base = rnd(M, N, seed=6).to(DEV)
        acc = base.clone()
        ops.gemm_bf16w(x.to(DEV), wb, out=acc, accumulate=True)
    finally:
        ops.WIDE_GEMM, ops.GEMM256, ops.DL_GEMM = keep
        _lib.lib().vxb_debug_set_gemm_wide_waves(8)
        _lib.lib().vxb_debug_set_gemm_wide_experiment(0)
    assert torch.equal(got, ref)
    close(got, F.leaky_relu(x.double() @ W.double().t() + b.double(), 0.02).float() + r, 2e-5, 'gemm_wide vs fp64')
    close(acc, (base.cpu().double() + x.double() @ W.double().t()).float(), 2e-5, 'gemm_wide accumulate')


@pytest.fixture(params=[8, 4])
def _wide_waves(request):
    from voxactb_amd import _lib
    _lib.lib().vxb_debug_set_gemm_wide_waves(request.param)
    yield request.param
    _lib.lib().vxb_debug_set_gemm_wide_waves(8)


@pytest.mark.parametrize('M,F,K', [(2048, 256, 256), (2300, 2048, 512)])
def test_geglu_fused_into_the_wide_gemm_is_bit_identical(M, F, K, _wide_waves):
    """FeedForward's up-projection + GEGLU from one launch (value / gate rows interleaved in the fragment order), and the data
    gradient of the down-projection + GEGLU's backward: the same bits as the separate passes; the interleaved fragments written by
    the batched weight split equal the gather + shuffle of ops.gemm_wfrag_geglu."""
    x, W1, b1 = rnd(M, K).to(DEV), (rnd(2 * F, K, seed=1) * 0.05).to(DEV), rnd(2 * F, seed=2).to(DEV)
    old = ops.PRECISION
    ops.PRECISION = 'bf16x3'
    keep = ops.FUSE_GEGLU
    try:
        ops.new_step()
        ops.FUSE_GEGLU = False
        h0, g0 = ops.linear_geglu(x, W1, b1)
        ops.FUSE_GEGLU = True
        h1, g1 = ops.linear_geglu(x, W1, b1)                 # fragments by gather + shuffle
        assert torch.equal(h0, h1) and torch.equal(g0, g1)
        ref = ops.gemm_wfrag_geglu(ops._bf16_weight(W1, False)).clone()
        ops.new_step()
        ops.prepare_linear_weights([W1], geglu=[W1])         # fragments by the batched split (flag bit 2)
        got = ops.gemm_wfrag_geglu(ops._bf16_weight(W1, False))
        if (2 * F) % 512 == 0 and K % 32 == 0 and K >= 256:
            assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
        h2, g2 = ops.linear_geglu(x, W1, b1)
        assert torch.equal(h0, h2) and torch.equal(g0, g2)
        if F % 512 == 0:
            # backward: dy [M][Kout] @ W2 [Kout][F] -> d(gg) -> dh
            Ko = 512
            W2, dy = (rnd(Ko, F, seed=3) * 0.05).to(DEV), rnd(M, Ko, seed=4).to(DEV)
            ops.FUSE_GEGLU_BWD = True                        # (off by default: slower in the step, see ops.py)
            try:
                dh = ops.linear_dgrad_geglu_bwd(dy, W2, h0)
            finally:
                ops.FUSE_GEGLU_BWD = False
            assert dh is not None
            dgg = torch.empty(M, F, device=DEV)
            ops.linear_bwd(g0, W2, dy, torch.zeros(Ko, F, device=DEV), None, dgg)
            want = ops.geglu_bwd(h0, dgg)
            assert torch.equal(dh, want)
    finally:
        ops.PRECISION = old
        ops.FUSE_GEGLU = keep
        ops.new_step()


@pytest.mark.parametrize('M,N,K', [(2048, 1024, 512), (2300, 512, 512)])
def test_wide_gemm_fp16_plane_of_its_result(M, N, K):
    """vxb_gemm_wide_bf16x3_f16out_f32 (ops.linear(..., f16_out=)): the product as the plain entry writes it, and its fp16 plane -- the k | v
    operand of the pipelined attention kernels, perceiver_lang_io.py:112-113 -- with the bits of vxb_split_f16_f32 applied to that result;
    values beyond the largest half saturate.  Ragged M; both epilogues (vxb_debug_set_gemm_wide_experiment(64))."""
    from voxactb_amd import _lib, flash
    x, W, b = rnd(M, K).to(DEV), rnd(N, K, seed=1).to(DEV), rnd(N, seed=2).to(DEV)
    x[5, :] *= 3e3                       # a row whose outputs overflow half
    old = ops.PRECISION
    ops.PRECISION = 'bf16x3'
    try:
        ops.new_step()
        ref = ops.linear(x, W, b)
        planes = flash.kv_planes(ref, 'f16')
        assert float(ref.abs().max()) > 65504.0
        for bits in (0, 64):
            _lib.lib().vxb_debug_set_gemm_wide_experiment(bits)
            f16 = torch.zeros((M, N), dtype=torch.float16, device=DEV)
            got, filled = ops.linear(x, W, b, f16_out=f16)
            assert filled and torch.equal(got, ref)
            assert torch.equal(f16.view(torch.int16), planes[0].view(torch.int16)), bits
    finally:
        _lib.lib().vxb_debug_set_gemm_wide_experiment(0)
        ops.PRECISION = old
        ops.new_step()
