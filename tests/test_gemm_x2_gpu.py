"""GPU: data gradient of a linear layer on two fp16 products (vxb_gemm_wide_f16x2_f32 through ops.linear_bwd: dY * 2^k as an fp16
hi + lo pair, the weight as one fp16 value, its single-plane fragments made by the batched weight split) against the bf16x3 GEMM:
equal to fp32 rounding when the weights ARE fp16 values, 2^-12 per weight otherwise; gradients of ordinary, tiny and huge magnitude
(first call: exact operand scale from an absmax pass; later calls: the delayed scale of the weight-gradient launch)."""
import pytest
import torch

from voxactb_amd import ops
from .test_ops_gpu import rnd, DEV

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _wide_kernels_at_test_sizes():
    """the wide 128 x 512-tile kernels are dispatched from 16384 rows on (ops.WIDE_MIN_M: below that they leave most CUs idle); these
    tests exercise them at a few thousand rows"""
    ops.set_wide_min_rows(1024)
    yield
    ops.set_wide_min_rows(16384)


@pytest.mark.parametrize('N,K', [(512, 512), (256, 1024), (2048, 512)])
@pytest.mark.parametrize('gain', [1.0, 2e-9, 5e5])
def test_linear_dgrad_on_two_fp16_products(N, K, gain):
    M = 2048
    x = rnd(M, K, seed=1).to(DEV)
    dy = (rnd(M, N, seed=2) * gain).to(DEV)
    dy[3, :5] *= 30.0
    res = {}
    state = (ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops.DGRAD_PRECISION)
    try:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16 = 'bf16x3', 'fp16', True
        for exact_w in (True, False):
            W = (rnd(N, K, seed=3, scale=0.05)).to(DEV)
            if exact_w:
                W = W.half().float()
            out = {}
            for mode in ('bf16x3', 'fp16x2'):
                ops.DGRAD_PRECISION = mode
                ops.new_step()
                ops._GRAD_SCALE.clear()
                ops.prepare_linear_weights([W], f16_dgrad=True)
                dxs = []
                for it in range(2):                      # second call: the delayed scale reported by the first weight-gradient launch
                    dW, db, dx = torch.zeros_like(W), torch.zeros(N, device=DEV), torch.full((M, K), float('nan'), device=DEV)
                    ops.begin_backward()
                    ops.linear_bwd(x, W, dy, dW, db, dx)
                    dxs.append(dx)
                assert torch.isfinite(dxs[0]).all() and torch.isfinite(dxs[1]).all()
                out[mode] = dxs
            ref = out['bf16x3'][0]
            assert torch.equal(out['bf16x3'][0], out['bf16x3'][1])
            scale = float(ref.abs().max())
            res[exact_w] = [float((d - ref).abs().max()) / scale for d in out['fp16x2']]
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops.DGRAD_PRECISION = state
        ops.new_step()
        ops._GRAD_SCALE.clear()
    assert max(res[True]) < 2e-5, res                 # exact weights, dY in 22 bits: what is left is bf16x3's own dropped lo * lo terms
    assert 0 < max(res[False]) < 4e-4, res            # ... the 2^-12 weight rounding over K terms


def test_linear_dgrad_accumulates_into_dx():
    M, N, K = 1024, 512, 512
    x, dy, W = rnd(M, K, seed=1).to(DEV), rnd(M, N, seed=2).to(DEV), rnd(N, K, seed=3, scale=0.05).to(DEV).half().float()
    base = rnd(M, K, seed=4).to(DEV)
    state = (ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops.DGRAD_PRECISION)
    try:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops.DGRAD_PRECISION = 'bf16x3', 'fp16', True, 'fp16x2'
        ops.new_step()
        ops._GRAD_SCALE.clear()
        ops.prepare_linear_weights([W], f16_dgrad=True)
        dx0 = torch.empty(M, K, device=DEV)
        ops.begin_backward()
        ops.linear_bwd(x, W, dy, torch.zeros_like(W), None, dx0)
        dx1 = base.clone()
        ops.begin_backward()
        ops.linear_bwd(x, W, dy, torch.zeros_like(W), None, dx1, dx_accumulate=True)
    finally:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops.DGRAD_PRECISION = state
        ops.new_step()
        ops._GRAD_SCALE.clear()
    assert float((dx1 - (base + dx0)).abs().max()) < 2e-5 * float(dx0.abs().max())


@pytest.mark.parametrize('M,F,K', [(2048, 512, 512), (2300, 2048, 512)])
def test_geglu_backward_fused_into_the_two_product_data_gradient(M, F, K):
    """FeedForward's backward (perceiver_lang_io.py:74-78 / :100-106 through autograd): d(gg) = dy @ W2 on two fp16 products with GEGLU's
    backward in the wide kernel's row-contiguous epilogue (vxb_gemm_wide_geglu_bwd_f16x2_f32, ops.linear_dgrad_geglu_bwd) against the
    two launches it replaces (vxb_gemm_wide_f16x2_f32 + vxb_geglu_bwd_f32): equal bits, first on the exact operand scale, then on the
    delayed one.  Ragged M.  Also: the epilogue that stores straight out of the accumulators (vxb_debug_set_gemm_wide_experiment(64))."""
    from voxactb_amd import _lib
    gg = rnd(M, F, seed=1).to(DEV)
    h = rnd(M, 2 * F, seed=5).to(DEV)
    dy = (rnd(M, K, seed=2) * 3e-3).to(DEV)
    W2 = rnd(K, F, seed=3, scale=0.05).to(DEV)
    state = (ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops.DGRAD_PRECISION, ops.FUSE_GEGLU_BWD)
    try:
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops.DGRAD_PRECISION, ops.FUSE_GEGLU_BWD = 'bf16x3', 'fp16', True, 'fp16x2', 'x2'
        ops.new_step()
        ops._GRAD_SCALE.clear()
        ops.prepare_linear_weights([W2], f16_dgrad=True)
        assert ops.geglu_bwd_fusable(dy, W2, h)
        for it in range(2):
            dW, db = torch.zeros_like(W2), torch.zeros(K, device=DEV)
            ops.begin_backward()
            ops._LAST_LIN_DY_SCALE[0] = None
            ops.linear_bwd(gg, W2, dy, dW, db, None)
            sc = ops._LAST_LIN_DY_SCALE[0]
            assert sc is not None
            dgg = torch.full((M, F), float('nan'), device=DEV)
            ops.linear_dgrad(dy, W2, dgg, False, sc)
            ref = ops.geglu_bwd(h, dgg)
            for bits in (0, 64):
                _lib.lib().vxb_debug_set_gemm_wide_experiment(bits)
                got = ops.linear_dgrad_geglu_bwd(dy, W2, h, sc)
                assert got is not None and torch.equal(got, ref), (it, bits)
    finally:
        _lib.lib().vxb_debug_set_gemm_wide_experiment(0)
        ops.PRECISION, ops.WGRAD_PRECISION, ops.GENERIC_WGRAD_F16, ops.DGRAD_PRECISION, ops.FUSE_GEGLU_BWD = state
        ops.new_step()
        ops._GRAD_SCALE.clear()
