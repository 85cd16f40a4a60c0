"""GPU: bf16 matrix-core weight gradient (ds_read_b64_tr_b16 transposed operands) == fp32 kernel on bf16-rounded operands."""
import pytest
import torch

from voxactb_amd import ops
from tests.test_ops_gpu import rnd, close, cl, bf, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('Cin,Cout,k,s,S,B', [(64, 64, 3, 1, 6, 2), (128, 64, 5, 1, 5, 2), (64, 64, 5, 5, 10, 3), (64, 128, 3, 1, 7, 1),
                                               (64, 64, 3, 2, 8, 2)])
def test_wgrad_bf16_matches_rounded_fp32(Cin, Cout, k, s, S, B):
    p = k // 2
    G = (S + 2 * p - k) // s + 1
    x = cl(rnd(B, Cin, S, S, S)).to(DEV)
    dy = cl(rnd(B, Cout, G, G, G, seed=3)).to(DEV)
    ref = ops.conv3d_wgrad(bf(x), bf(dy), Cout, B, S, G, k, -p, stride=s, nsplit=2)
    got = ops.conv3d_wgrad(x, dy, Cout, B, S, G, k, -p, stride=s, nsplit=3, force_bf16=True)
    close(got, ref, 3e-5, 'bf16 wgrad')


def test_wgrad_bf16_two_sources_and_d2s():
    B, S = 2, 5
    a, c = cl(rnd(B, 64, S, S, S)).to(DEV), cl(rnd(B, 64, S, S, S, seed=5)).to(DEV)
    dy = cl(rnd(B, 64, S, S, S, seed=7)).to(DEV)
    ref = ops.conv3d_wgrad(bf(a), bf(dy), 64, B, S, S, 3, -1, src1=bf(c), nsplit=1)
    got = ops.conv3d_wgrad(a, dy, 64, B, S, S, 3, -1, src1=c, nsplit=2, force_bf16=True)
    close(got, ref, 3e-5, 'two-source')
    # polyphase: dY is the fine grid of a depth-to-space output
    k, s, G = 5, 5, 3
    L, R = ops.polyphase_tables(k, s)
    kl = 2 * R + 1
    z1 = cl(rnd(B, 64, G, G, G, seed=9)).to(DEV)
    dyf = cl(rnd(B, 64, G * s, G * s, G * s, seed=11)).to(DEV)
    ref = ops.conv3d_wgrad(bf(z1), bf(dyf), s ** 3 * 64, B, G, G, kl, -R, d2s=(s, 64), nsplit=1)
    got = ops.conv3d_wgrad(z1, dyf, s ** 3 * 64, B, G, G, kl, -R, d2s=(s, 64), nsplit=2, force_bf16=True)
    close(got, ref, 3e-5, 'polyphase d2s wgrad')


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('M,N,K', [(4096, 512, 512), (2048 + 96, 256, 128), (8192, 1024, 64)])
def test_linear_bias_gradient_comes_out_of_the_weight_gradient_launch(mode, M, N, K):
    """linear_bwd in the matrix-core modes: db = column sums of dy are taken inside the weight-gradient kernel (`possum`) -- exact
    fp32 sums in a fixed order, equal to vxb_colsum_f32 up to the summation order; dW and dx unchanged."""
    from .test_ops_gpu import rnd, DEV
    x, W, dy = rnd(M, K).to(DEV), rnd(N, K, seed=1).to(DEV), rnd(M, N, seed=2).to(DEV)
    ops.PRECISION = mode
    try:
        dW, db, dx = torch.zeros(N, K, device=DEV), torch.full((N,), 0.5, device=DEV), torch.empty(M, K, device=DEV)
        ops.linear_bwd(x, W, dy, dW, db, dx)
        ref = torch.full((N,), 0.5, device=DEV)
        ops.colsum(dy, ref, accumulate=True)
    finally:
        ops.PRECISION = 'fp32'
    want = dy.double().sum(0).cpu() + 0.5
    assert float((db.double().cpu() - want).abs().max()) < 1e-4 * float(want.abs().max())
    assert float((db - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert float((dW.double().cpu() - (dy.double().t() @ x.double()).cpu()).abs().max()) < (3e-2 if mode == 'bf16' else 2e-4) * float(dW.abs().max())


def test_generic_weight_gradients_on_single_fp16_products_with_delayed_scaling():
    """VOXACTB_GENERIC_WGRAD_F16: a linear layer's weight gradient (plain-GEMM form, gradient operand = src0) and a 5^3 conv's
    (gradient operand = dy) with ONE fp16 product per term.  First call at a site: explicit absmax scale; later calls use the
    maximum the previous launch reported (with 5 bits of headroom) -- checked by growing / shrinking the gradient 30x between calls
    (inside the margin: same accuracy) and 3000x (the stale scale saturates THAT call, the next one is right again)."""
    from .test_ops_gpu import rnd, DEV, cl
    M, N, K = 4096, 256, 512
    x, W = rnd(M, K).to(DEV), rnd(N, K, seed=1).to(DEV)
    ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops._GRAD_SCALE = 'bf16x3', 'fp16', True, {}
    try:
        errs = []
        for gain in (1e-5, 3e-4, 1e-5, 3e-2, 3e-2):
            dy = rnd(M, N, seed=2).to(DEV) * gain
            dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
            ops.begin_backward()                  # (every call of this loop stands for one training step)
            ops.linear_bwd(x, W, dy, dW, db, None)
            ref = (dy.double().t() @ x.double())
            errs.append(float((dW.double() - ref).abs().max() / ref.abs().max()))
            assert float((db.double() - dy.double().sum(0)).abs().max()) < 1e-4 * float(dy.double().sum(0).abs().max())     # bias: exact fp32 sums
        assert max(errs[:3]) < 1e-3 and errs[4] < 1e-3 and errs[3] > 0.1, errs        # (errs[3]: the 3000x jump meets a stale scale)
        assert (('lin', W.data_ptr()), 0) in ops._GRAD_SCALE
        # a second use of the same weight INSIDE one backward pass (transformer_iterations > 1, tied layers) keeps its own delayed scale
        ops.begin_backward()
        for use, gain in enumerate((3e-2, 1e-6)):
            dy = rnd(M, N, seed=2).to(DEV) * gain
            dW = torch.zeros(N, K, device=DEV)
            ops.linear_bwd(x, W, dy, dW, None, None)
            ref = (dy.double().t() @ x.double())
            assert float((dW.double() - ref).abs().max() / ref.abs().max()) < 1e-3, (use, gain)
        assert (('lin', W.data_ptr()), 1) in ops._GRAD_SCALE
        # 5^3 conv (the decoder's first up-conv): gradient operand = dy
        B, S, Ci, Co = 2, 12, 128, 64
        a = cl(rnd(B, Ci, S, S, S)).to(DEV)
        g = cl(rnd(B, Co, S, S, S, seed=3) * 1e-6).to(DEV)
        ops.GENERIC_WGRAD_F16 = False
        want = ops.conv3d_wgrad(a, g, Co, B, S, S, 5, -2, grad_key=('conv', 7))
        ops.GENERIC_WGRAD_F16 = True
        ops.begin_backward()
        got = ops.conv3d_wgrad(a, g, Co, B, S, S, 5, -2, grad_key=('conv', 7))
        ops.begin_backward()
        got2 = ops.conv3d_wgrad(a, g, Co, B, S, S, 5, -2, grad_key=('conv', 7))       # second call: delayed scale == the exact one
        assert float((got - want).abs().max()) < 1e-3 * float(want.abs().max()) and torch.equal(got, got2)
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops._GRAD_SCALE = 'fp32', '', False, {}


@pytest.fixture
def _wide_at_test_sizes():
    ops.set_wide_min_rows(1024)          # (dispatched from 16384 rows on by default: ops.WIDE_MIN_M)
    yield
    ops.set_wide_min_rows(16384)


@pytest.mark.parametrize('M,N,K', [(4096, 512, 1024), (2048, 512, 512), (4096, 1024, 512), (2048 + 8, 512, 1024), (4096, 384, 128)])
def test_linear_weight_gradient_fp16_wide_and_pipelined_kernels(_wide_at_test_sizes, M, N, K):
    """The plain-GEMM forms of the fp16 weight gradient (wgrad_bf16.hip): wide 128 x 512 tiles when one side has 512 channels --
    with the wide operand as `dy` argument (transposed store, bias sums and gradient maximum taken from the narrow operand: N = 512,
    K = 1024) or as src0 (N = 1024, K = 512) --, the pipelined 128 x 128 kernel otherwise (M % 16 != 0, no 512 side); each against
    float64 and against the generic kernel (vxb_debug_set_wgrad_lin(0)), bias gradients exact, second call on the delayed scale."""
    from .test_ops_gpu import rnd, DEV
    from voxactb_amd import _lib
    x, W = rnd(M, K).to(DEV), rnd(N, K, seed=1).to(DEV)
    dy = rnd(M, N, seed=2).to(DEV) * 3e-4
    ref = dy.double().t() @ x.double()
    ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops._GRAD_SCALE = 'bf16x3', 'fp16', True, {}
    try:
        outs = []
        for mode in (2, 2, 0):                  # first call: explicit scale; second: the scale the first launch reported; then the generic kernel
            _lib.lib().vxb_debug_set_wgrad_lin(mode)
            dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
            ops.begin_backward()
            ops.linear_bwd(x, W, dy, dW, db, None)
            assert float((dW.double() - ref).abs().max() / ref.abs().max()) < 1e-3
            assert float((db.double() - dy.double().sum(0)).abs().max()) < 1e-4 * float(dy.double().sum(0).abs().max())
            outs.append(dW)
        assert float((outs[0] - outs[2]).abs().max()) < 1e-3 * float(ref.abs().max())
    finally:
        _lib.lib().vxb_debug_set_wgrad_lin(2)
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops._GRAD_SCALE = 'fp32', '', False, {}


@pytest.mark.parametrize('M,N,K', [(32768, 4096, 512), (32768, 512, 2048), (4096 + 16 * 5, 512, 512)])
def test_wide_weight_gradient_main_loop_against_its_guarded_form(M, N, K):
    """wgrad_wide_f16_kernel (round 6): the main loop -- unconditional loads and stores, flags at compile time, the conversion of the next tile
    and the loads issued between the MFMAs, workgroups in XCD-aware order -- accumulates in the order of the guarded steps of rounds 3 - 5:
    dW, db and the reported operand scale are bit-identical with every step through the guarded form (vxb_debug_set_wgrad_lin(2 + 64)) and
    in the plain workgroup order (2 + 16); a NaN / an inf anywhere in dy reaches the reported scale in both (slices of 5 .. 128 tiles: the
    main loop's rounds, the guarded remainder and slices shorter than a round are all taken)."""
    from .test_ops_gpu import rnd, DEV
    from voxactb_amd import _lib
    x, W = rnd(M, K).to(DEV), rnd(N, K, seed=1).to(DEV)
    dy = rnd(M, N, seed=2).to(DEV) * 3e-4
    ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16 = 'bf16x3', 'fp16', True
    ops.set_wide_min_rows(1024)
    try:
        def run(mode, *dys):
            _lib.lib().vxb_debug_set_wgrad_lin(mode)
            ops._GRAD_SCALE = {}
            for dy_ in dys:                      # (the first call takes its scale from a pass over dy, the next ones from the launch before)
                dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
                ops.begin_backward()
                ops.linear_bwd(x, W, dy_, dW, db, None)
            (st,) = ops._GRAD_SCALE.values()
            return dW, db, st[0].clone()         # st[0]: the scale this launch reported for the next one
        a = run(2, dy)
        ref = dy.double().t() @ x.double()
        assert float((a[0].double() - ref).abs().max() / ref.abs().max()) < 1e-3
        for mode in (2 + 64, 2 + 16):
            b = run(mode, dy)
            for u, v in zip(a, b):
                assert torch.equal(u, v), mode
        for bad in (float('nan'), float('inf')):
            dyb = dy.clone()
            dyb[M - 7, N // 2 + 3] = bad
            for mode in (2, 2 + 64):
                assert bool(torch.isfinite(run(mode, dy, dy)[2]).all())
                sc = run(mode, dy, dyb)[2]
                assert not bool(torch.isfinite(sc).all()), (bad, mode, sc)
    finally:
        _lib.lib().vxb_debug_set_wgrad_lin(2)
        ops.set_wide_min_rows(16384)
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops._GRAD_SCALE = 'fp32', '', False, {}
