"""Gradient gates against the FLOAT64 reference (fixtures f5n_noise_*: tests/golden/make_golden.py, section F5n).

The reference encoder was run in fp32 AND in float64 on three seeded batches per headline shape (configs[1] geometry, B = 1; the
configs[2] twin-agent shape, B = 2).  The fixture holds, per parameter tensor, ||g64||, ||g32 - g64|| (the reference's own rounding
noise) and 16 random +-1 projections of g64, from which this test estimates ||g - g64|| of the product's gradient (E[(s.(a - b))^2] =
||a - b||^2; 16 projections: +-18 %) without the 133 MB tensor.

Gate, stated once:   ||g - g64|| <= max(REL * ||g64||, K * ||g32 - g64||) + ABS      with REL = 5e-3, K = 3, ABS = 1e-7 * max ||g64||
i.e. every parameter gradient within 0.5 % (relative L2 -- a stronger statement than the 3e-3 gate on gradient NORMS of the digests in
test_c2_reference_gpu.py, which cannot see an error orthogonal to the gradient) of the float64 truth, or within three times the
reference's own fp32 error where that is larger (conv bias gradients: fp32 sums over 10^6 voxels, 0.6 - 1.3 %).  Measured in the default
precision over the regular fixtures: median 3e-4 .. 9e-4 per fixture, worst tensor 3.9e-3 (the up-conv's weight gradient, single fp16
products over 8000 positions per sample; 5e-4 with VOXACTB_WGRAD_PRECISION=bf16x3); exact-fp32 mode: median 1e-4 .. 3e-4, worst 5e-4.  Q-values: within 1e-4 of the float64 forward
(BASELINE.json north_star).  Every precision the engine ships is held to the same gate on every seed.

Root cause of the 'forward-sensitive' batches of round 4 (c2_s3, v50b_s1: gradients of up0 and everything upstream 3 - 8 % off float64
in the default precision, 2e-4 in exact fp32), found in round 5 by swapping saved activations between an fp32 and a bf16x3 forward one
group at a time (tools/experiments/fwd_sensitivity_gpu.py, profiles/r05_fwd_sensitivity.log): the WHOLE difference enters through u0, and
through 95 of its 64 000 000 elements (21 of 8 M at V = 50) -- those whose pre-activation is within 1e-6 of zero and comes out with the
other sign.  LeakyReLU' jumps from 0.02 to 1 there (network_utils.py:12, :128-170); parameter gradients are cancelling sums over 10^6
voxels with heavy-tailed terms (SpatialSoftmax3D's 1 / 0.01), so one such element can weigh percents.  The loss is piecewise smooth and
the product evaluates a DIFFERENT, equally valid subgradient -- the same situation as a max-pool tie.  The float64 run's choices at every
pre-activation within 3e-5 of zero are therefore part of the fixtures (f5n_kinks_*.npz: 65 KB per batch) and the backward is evaluated at
them, as it already was at the float64 run's pool arg-maxima; with that, every batch holds the regular 0.5 % gate in every precision and
the 10 % carve-out of round 4 is gone.  test_unforced_kink_choices_* keeps the un-forced numbers on record."""
import os

import numpy as np
import pytest
import torch

from oracle import weights as ow
from tests.test_c2_reference_gpu import DEV, T, _setup, force_kinks
from voxactb_amd import ops

pytestmark = pytest.mark.gpu
REL, K_NOISE = 5e-3, 3.0
FIXTURES = ['f5n_noise_c2_s1', 'f5n_noise_c2_s2', 'f5n_noise_c2_s3', 'f5n_noise_c3_s1', 'f5n_noise_c3_s2', 'f5n_noise_c3_s3',
            'f5n_noise_v50a_s1', 'f5n_noise_v50b_s1']
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _kinks(fixture_name):
    """the float64 run's LeakyReLU choices near zero for this fixture's batch (f5n_kinks_*.npz, make_golden.py: grad_noise_kinks), or None"""
    path = os.path.join(GOLDEN, fixture_name.replace('f5n_noise_', 'f5n_kinks_') + '.npz')
    return np.load(path, allow_pickle=False) if os.path.exists(path) else None


def _grads(g, precision, attn_kernel, attn_gx, force_pools=True, kinks=None):
    enc, rs, grid, arm, V, B = _setup(g)
    eng = enc.engine()
    eng.precision = precision
    eng.attn_kernel, _, eng.attn_bwd_kernel = attn_kernel.partition('/')
    eng.attn_bwd_gx = attn_gx
    outs, cache = eng.forward(grid, rs['low_dim_state'].to(DEV), rs['lang_token_embs'].to(DEV), training=False, save=True)
    at = rs['trans_action_indicies'].long()
    lab = ((at[:, 0] * V + at[:, 1]) * V + at[:, 2]).int().to(DEV)
    dq = torch.empty((B, V ** 3), device=DEV)
    l_t, _, _ = ops.ce_big(outs[0].view(B, -1), lab, dq, 1.0 / B)
    labs = torch.cat([rs['rot_grip_action_indicies'].int(), rs['ignore_collisions'].int()[:, :1]], 1).to(DEV).contiguous()
    d_o = torch.empty_like(cache['o'])
    l_h, _ = ops.ce_rows(cache['o'], [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)], labs, d_o, 1.0 / B)
    total = l_t + l_h.sum(1)
    d_arm = None
    if arm:
        d_arm = torch.empty_like(outs[3])
        la, _ = ops.ce_rows(outs[3], [(0, 2)], rs['label'].int()[:, :1].to(DEV).contiguous(), d_arm, 1.0 / B)
        total = total + la[:, 0]
    loss = float(total.mean())
    for p in enc.parameters():
        p.grad = None
    flips = None
    if force_pools and 'pool_argmax64_0' in g.files:
        # the backward at the float64 run's max-pool choices (the loss is piecewise smooth: see make_golden.py, grad_noise_fixture)
        flips = []
        for i, key in enumerate(('ss0', 'ss1', 'ss2')):
            ss, mx, st, am = cache[key]
            ref = T(g['pool_argmax64_%d' % i]).to(DEV).int().reshape(am.shape).contiguous()
            flips.append(int((am != ref).sum()))
            cache[key] = (ss, mx, st, ref)
    if kinks is not None:
        flips = (flips or []) + ['LeakyReLU: %d' % force_kinks(cache, kinks)]
    eng.backward(cache, dq, d_o, d_arm)
    return enc, outs, loss, arm, flips


def _measure(g, precision, attn_kernel, attn_gx, tag, kinks=None):
    enc, outs, loss, arm, flips = _grads(g, precision, attn_kernel, attn_gx, kinks=kinks)
    B = outs[0].shape[0]
    flat = outs[0].reshape(B, -1).double().cpu()
    sidx = T(g['q_trans_sample_idx']).long()
    eq = max(float((flat[:, sidx] - T(g['q_trans_sample'])).abs().max()),
             float((torch.gather(flat, 1, T(g['q_trans_top_idx']).long()) - T(g['q_trans_top_vals'])).abs().max()),
             float((outs[1].double().cpu() - T(g['rot_grip'])).abs().max()), float((outs[2].double().cpu() - T(g['collision'])).abs().max()))
    if arm:
        eq = max(eq, float((outs[3].double().cpu() - T(g['arm_out'])).abs().max()))
    names = [str(n) for n in g['grad_names']]
    n64, e32, p64 = T(g['grad_norm64']), T(g['grad_err32']), T(g['grad_proj64'])
    nproj = int(g['nproj'])
    P = dict(enc.named_parameters())
    rows = []
    absfloor = 1e-7 * float(n64.max())
    for i, n in enumerate(names):
        est = float(ow.projection_error(ow.project(P[n].grad, n, nproj), p64[i]))
        lim = max(REL * float(n64[i]), K_NOISE * float(e32[i])) + absfloor
        rows.append((est / lim, n, est, float(n64[i]), float(e32[i])))
    rows.sort(reverse=True)
    print('%s: loss %.6f (f64 %.6f, reference fp32 %.6f) | max |Q - Q64| %.2e (reference fp32: %.2e) | max-pool choices differing from float64: %s '
          '(reference fp32: %s)' % (tag, loss, float(g['loss']), float(g['loss32']), eq, float(T(g['q_spread32']).max()), flips,
                                    g['pool_flips32'].tolist() if 'pool_flips32' in g.files else None))
    for r in rows[:6]:
        print('   %-46s ||g-g64||/||g64|| %.2e   reference fp32 %.2e   (x gate %.2f)' % (r[1], r[2] / (r[3] + 1e-300), r[4] / (r[3] + 1e-300), r[0]))
    med = float(np.median([r[2] / (r[3] + 1e-300) for r in rows if r[3] > 1e-6]))
    print('   median relative L2 error over the tensors %.2e' % med)
    return eq, rows, loss


def _available():
    return [f for f in FIXTURES if os.path.exists(os.path.join(GOLDEN, f + '.npz'))]


@pytest.mark.parametrize('fixture', FIXTURES)
@pytest.mark.parametrize('mode,wide_dispatch', [('fp32', False), ('bf16x3+r3/f16', False), ('bf16x3+r3/f16-gx0', False), ('bf16x3+r3/f16', True),
                                                ('bf16x3+auto/f16', False)],
                         indirect=['wide_dispatch'], ids=['fp32', 'bf16x3+r3/f16', 'bf16x3+r3/f16-gx0', 'bf16x3+r3/f16-wide', 'bf16x3+auto/f16'])
def test_gradients_against_the_float64_reference(golden, fixture, mode, wide_dispatch):
    """'fp32': the exact-fp32 kernels; 'bf16x3+r3/f16': the shipped default (bf16x3 forward incl. round 3's attention forward, fp16 single /
    double products in the backward, pipelined fp16 attention backward with hi + lo gradient operands); '-wide': through the kernels the
    B = 16 headline dispatches; 'bf16x3+auto/f16': the named mode VOXACTB_ATTN_KERNEL=auto (the attention FORWARD on single fp16 products at
    these sizes; bench.py's `bf16x3+attn_f16`) -- it holds this gate too since the kink effect is separated out (round 4 had rejected it on
    this very test), and stays a named mode for the F5c3 element gates it misses (perceiver_lang_io.py: attn_kernel).  Backward evaluated at
    the float64 run's max-pool and LeakyReLU choices (module docstring)."""
    if fixture not in _available():
        pytest.skip('fixture not generated')
    g = golden(fixture)
    kinks = _kinks(fixture)
    assert kinks is not None, 'tests/golden/%s.npz is missing (make_golden.py --only f5k_*)' % fixture.replace('f5n_noise_', 'f5n_kinks_')
    assert abs(float(kinks['q_trans_lse'][0]) - float(g['q_trans_lse'][0])) < 1e-9              # (the same float64 forward)
    precision, _, attn = mode.partition('+')
    eq, rows, loss = _measure(g, precision, attn.replace('-gx0', '') or 'r3', not attn.endswith('-gx0'), '%s/%s' % (fixture[10:], mode), kinks=kinks)
    assert eq < 1e-4
    assert abs(loss - float(g['loss'])) < 1e-4
    bad = [(r[1], 'x gate %.2f' % r[0], '%.2e' % (r[2] / (r[3] + 1e-300))) for r in rows if r[0] > 1.0]
    assert not bad, bad


@pytest.mark.parametrize('fixture', ['f5n_noise_c2_s3', 'f5n_noise_v50b_s1', 'f5n_noise_c2_s1'])
def test_unforced_kink_choices_are_what_moved_the_forward_sensitive_batches(golden, fixture):
    """On record: the default precision WITHOUT the float64 run's LeakyReLU choices.  c2_s3 / v50b_s1 then sit 3 - 8 % / ~1 % off float64
    upstream of u0 (round 4's carve-out); c2_s1 is unaffected.  Asserted: forcing the choices -- a few dozen elements of 64 M -- is the
    whole difference (the forced run holds the regular gate, tested above; here: the un-forced worst tensor is at least 5 x the forced one
    on the two sensitive batches), and nothing is worse than round 4's 10 %."""
    if fixture not in _available() or _kinks(fixture) is None:
        pytest.skip('fixture not generated')
    g = golden(fixture)

    def worst(rows):
        top = max(q[3] for q in rows)
        return max(r[2] / (r[3] + 1e-300) for r in rows if r[3] > 1e-6 * top)
    _, rows_u, _ = _measure(g, 'bf16x3', 'r3/f16', True, '%s/unforced' % fixture[10:])
    _, rows_f, _ = _measure(g, 'bf16x3', 'r3/f16', True, '%s/forced' % fixture[10:], kinks=_kinks(fixture))
    wu, wf = worst(rows_u), worst(rows_f)
    print('%s: worst tensor, relative L2 vs float64: un-forced %.2e, at the float64 choices %.2e' % (fixture[10:], wu, wf))
    assert wu < 0.10
    if fixture != 'f5n_noise_c2_s1':
        assert wu > 5 * wf, (wu, wf)


@pytest.mark.parametrize('fixture', ['f5n_noise_c2_s1', 'f5n_noise_c2_s3', 'f5n_noise_c3_s2', 'f5n_noise_v50b_s1'])
@pytest.mark.parametrize('variant', ['f16-gx1', 'f16-gx0'])
def test_pipelined_attention_backward_equals_the_bf16x3_backward_on_the_same_forward(golden, fixture, variant):
    """The backward arithmetic in isolation: the SAME forward (default precision), then the attention backward on round 3's bf16x3 kernels
    and on the pipelined fp16 kernels (gradient operands hi + lo, or single) -- every parameter gradient within 1e-3 (hi + lo) / 3e-3
    (single) relative L2 of each other, also on the forward-sensitive batches (same forward, same masks)."""
    if fixture not in _available():
        pytest.skip('fixture not generated')
    g = golden(fixture)
    enc_a, _, _, _, _ = _grads(g, 'bf16x3', 'r3/', True, force_pools=False)
    ga = {n: p.grad.detach().clone() for n, p in enc_a.named_parameters()}
    del enc_a
    torch.cuda.empty_cache()
    enc_b, _, _, _, _ = _grads(g, 'bf16x3', 'r3/f16', variant.endswith('gx1'), force_pools=False)
    worst, wn = 0.0, ''
    gmax = max(float(v.norm()) for v in ga.values())
    for n, p in enc_b.named_parameters():
        nr = float(ga[n].norm())
        if nr < 1e-6 * gmax:
            continue
        e = float((p.grad - ga[n]).norm()) / nr
        if e > worst:
            worst, wn = e, n
    print('%s %s: worst relative L2 difference %.2e (%s)' % (fixture[10:], variant, worst, wn))
    assert worst < (1e-3 if variant.endswith('gx1') else 3e-3), (wn, worst)
