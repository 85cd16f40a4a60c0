"""Precision experiment (GPU): which matrix products tolerate ONE 16-bit operand pair instead of the bf16x3 triple?

The exact-fp32 kernels (v_mfma_f32_32x32x2_f32) fed with operands that were first rounded to fp16 / bf16 compute what a
single fp16 / bf16 MFMA with fp32 accumulation computes (the product of two 11-bit or 8-bit mantissas is exact in fp32), so
the effect of a cheaper product type on the REFERENCE digests (F5g / F5c3: loss, Q-values, every parameter-gradient norm,
the small gradients in full) can be measured before any kernel is written.

    python tools/experiments/emu_precision.py [f5g|f5c3] ...

Each row = one configuration {forward linears, forward convs, forward attention core, backward products, loss scale}.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from voxactb_amd import ops                                    # noqa: E402
import tests.test_c2_reference_gpu as T                        # noqa: E402

DEV = 'cuda:0'
KINDS = ('lin_fwd', 'conv_fwd', 'attn_fwd', 'conv_dgrad', 'conv_wgrad', 'lin_dgrad', 'lin_wgrad', 'attn_bwd', 'final_dgrad', 'final_wgrad', 'up_wgrad')
EMU = {k: None for k in KINDS}
PHASE = ['fwd']


def rnd(t, kind, role='a'):
    """kind 'fp16w' / 'fp16a' (round 5): only the weight operand (role 'w') / only the data operand (role 'a') is rounded to fp16 --
    what a TWO-product scheme computes that carries the other operand as an fp16 hi + lo pair (22 bits: exact for this purpose)."""
    if t is None or kind is None:
        return t
    if kind in ('fp16w', 'fp16a'):
        return t.half().float() if kind[-1] == role else t
    if kind == 'fp16':
        return t.half().float()
    if kind == 'bf16':
        return t.bfloat16().float()
    raise ValueError(kind)


def rnd_(t, kind):
    if t is not None and kind is not None:
        t.copy_(rnd(t, kind))
    return t


_linear, _linear_bwd, _conv3d, _conv3d_wgrad, _gemm = ops.linear, ops.linear_bwd, ops.conv3d, ops.conv3d_wgrad, ops.gemm


def linear(x, W, bias=None, act=ops.ACT_NONE, residual=None, out=None):
    k = EMU['lin_fwd'] if PHASE[0] == 'fwd' else None
    return _linear(rnd(x, k), rnd(W, k), bias, act, residual, out)


HALF = [None]
BIG_ONLY = [False]      # round only the linears with >= 1024 rows (the ones that reach the matrix-core weight-gradient kernel)


def linear_bwd(x, W, dy, dW, db=None, dx=None, dx_accumulate=False, ws=None):
    kw, kd = EMU['lin_wgrad'], EMU['lin_dgrad']
    if BIG_ONLY[0] and x.shape[0] < 1024:
        kw = kd = None
    if kw == kd:
        _linear_bwd(rnd(x, kw), rnd(W, kw), rnd(dy, kw), dW, None, dx, dx_accumulate, ws)
    else:
        _linear_bwd(rnd(x, kw), W, rnd(dy, kw), dW, None, None, False, ws)                    # weight gradient only
        if dx is not None:
            _linear_bwd(x, rnd(W, kd, 'w'), rnd(dy, kd), torch.zeros_like(dW), None, dx, dx_accumulate, ws)   # data gradient only
    if db is not None:
        ops.colsum(dy, db, accumulate=True)             # bias gradients are fp32 column sums of the unrounded dy


def conv3d(src0, wt, *a, **kw):
    k = EMU['conv_fwd'] if PHASE[0] == 'fwd' else EMU['conv_dgrad']
    if PHASE[0] == 'bwd' and len(a) >= 5 and a[0] == 128 and a[4] == 3:          # (N, B, S_in, S_out, kext): final's data gradient
        k = EMU['final_dgrad'] or k
        if HALF[0] == 'both':
            # round 5: d(d0) half as shipped (single fp16 product: both operands rounded), d(u0) half on the kind under test
            full = _conv3d(rnd(src0, 'fp16'), rnd(wt, 'fp16'), *a, **kw)
            other = _conv3d(rnd(src0, k), rnd(wt, k, 'w'), *a, **kw)
            full[..., 64:128] = other[..., 64:128]
            return full
        if HALF[0] is not None and k is not None:
            # only one 64-column half of the data gradient (0: d(d0), 1: d(u0)) on the cheap product, the other exact
            full = _conv3d(src0, wt, *a, **kw)
            cheap = _conv3d(rnd(src0, k), rnd(wt, k), *a, **kw)
            lo, hi = (0, 64) if HALF[0] == 0 else (64, 128)
            full[..., lo:hi] = cheap[..., lo:hi]
            return full
    if kw.get('src1') is not None:
        kw['src1'] = rnd(kw['src1'], k)
    return _conv3d(rnd(src0, k), rnd(wt, k, 'w'), *a, **kw)


def conv3d_wgrad(src0, dy, *a, **kw):
    k = EMU['conv_wgrad']
    if kw.get('src1') is not None:
        k = EMU['final_wgrad'] or k
    elif kw.get('d2s', (0, 0))[0] > 0:
        k = EMU['up_wgrad'] or k
    if kw.get('src1') is not None:
        kw['src1'] = rnd(kw['src1'], k)
    return _conv3d_wgrad(rnd(src0, k), rnd(dy, k), *a, **kw)


def gemm(A, B, C, *a, **kw):
    if kw.get('label') == 'attn_core':
        k = EMU['attn_fwd'] if PHASE[0] == 'fwd' else EMU['attn_bwd']
        rnd_(A, k)
        rnd_(B, k)
    return _gemm(A, B, C, *a, **kw)


ops.linear, ops.linear_bwd, ops.conv3d, ops.conv3d_wgrad, ops.gemm = linear, linear_bwd, conv3d, conv3d_wgrad, gemm


def run(g, fwd_precision, bwd_precision, emu, S, tag, attn_bwd=''):
    EMU.update({k: None for k in KINDS})
    if 'bwd' in emu:
        emu = dict(emu, **{k: emu['bwd'] for k in KINDS[3:]})
        emu.pop('bwd')
    EMU.update(emu)
    enc, rs, grid, arm, V, B = T._setup(g)
    eng = enc.engine()
    eng.precision = fwd_precision
    eng.bwd_precision = bwd_precision
    eng.attn_bwd_precision = attn_bwd
    PHASE[0] = 'fwd'
    outs, cache = eng.forward(grid, rs['low_dim_state'].to(DEV), rs['lang_token_embs'].to(DEV), training=False, save=True)
    flat = outs[0].reshape(B, -1).float().cpu()
    sidx = T.T(g['q_trans_sample_idx']).long()
    e_q = float((flat[:, sidx] - T.T(g['q_trans_sample'])).abs().max())
    e_r = float((outs[1].float().cpu() - T.T(g['rot_grip'])).abs().max())
    e_c = float((outs[2].float().cpu() - T.T(g['collision'])).abs().max())
    at = rs['trans_action_indicies'].long()
    lab = ((at[:, 0] * V + at[:, 1]) * V + at[:, 2]).int().to(DEV)
    dq = torch.empty((B, V ** 3), device=DEV)
    l_t, _, _ = ops.ce_big(outs[0].view(B, -1), lab, dq, S / B)
    labs = torch.cat([rs['rot_grip_action_indicies'].int(), rs['ignore_collisions'].int()[:, :1]], 1).to(DEV).contiguous()
    d_o = torch.empty_like(cache['o'])
    l_h, _ = ops.ce_rows(cache['o'], [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)], labs, d_o, S / B)
    total = l_t + l_h.sum(1)
    d_arm = None
    if arm:
        d_arm = torch.empty_like(outs[3])
        la, _ = ops.ce_rows(outs[3], [(0, 2)], rs['label'].int()[:, :1].to(DEV).contiguous(), d_arm, S / B)
        total = total + la[:, 0]
    loss = float(total.mean())
    for p in enc.parameters():
        p.grad = None
    PHASE[0] = 'bwd'
    eng.backward(cache, dq, d_o, d_arm)
    PHASE[0] = 'fwd'
    P = dict(enc.named_parameters())
    worst_n, worst_e, worst_b, nbad, names, wname = 0.0, 0.0, 0.0, 0, [], '-'
    nonfinite = 0
    for n, rn in zip([str(n) for n in g['grad_names']], T.T(g['grad_norms'])):
        gr = P[n].grad / S
        if not bool(torch.isfinite(gr).all()):
            nonfinite += 1
            continue
        gn, rn = float(gr.norm()), float(rn)
        key64 = 'dysum64__' + n
        if key64 in g.files:
            ref = T.T(g[key64])
            e = float((gr.double().cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
            worst_b = max(worst_b, e)
            continue
        if rn > 1e-4:
            rel = abs(gn - rn) / rn
            if rel > worst_n:
                worst_n, wname = rel, n
            if abs(gn - rn) > 3e-3 * rn + 1e-5:
                nbad += 1
                names.append(n)
        key = 'grad__' + n
        if key in g.files:
            ref = T.T(g[key])
            e = float((gr.float().cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
            worst_e = max(worst_e, e)
    print('%-58s q %.2e rot %.2e col %.2e | loss err %.2e | grad norm %.2e (%s) elem %.2e bias64 %.2e | gate fails %d nonfinite %d %s'
          % (tag, e_q, e_r, e_c, abs(loss - float(g['loss'])), worst_n, wname, worst_e, worst_b, nbad, nonfinite,
             names[:4]), flush=True)
    del cache, outs
    torch.cuda.empty_cache()


if __name__ == '__main__':
    fixtures = [a for a in sys.argv[1:] if not a.startswith('-')] or ['f5g']
    names = {'f5g': 'f5g_encoder_c2_grads', 'f5c3': 'f5c3_encoder_c3_digest'}
    quick = '--quick' in sys.argv
    for f in fixtures:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', names[f] + '.npz'), allow_pickle=False)
        print('== %s' % f)
        if '--round1' in sys.argv:
            run(g, 'fp32', '', {}, 1.0, 'fp32 fwd / fp32 bwd')
            run(g, 'bf16x3', '', {}, 1.0, 'bf16x3 fwd / bf16x3 bwd (shipped)')
            run(g, 'fp32', 'bf16x3', {}, 1.0, 'fp32 fwd / bf16x3 bwd (convs, linears; attention fp32)')
            run(g, 'bf16x3', 'bf16', {}, 1.0, 'bf16x3 fwd / bf16 bwd (real kernels)')
            run(g, 'fp32', '', dict(bwd='bf16'), 1.0, 'fp32 fwd / EMU bf16 bwd')
            for S in [256.0, 4096.0]:
                run(g, 'fp32', '', dict(bwd='fp16'), S, 'fp32 fwd / EMU fp16 bwd, loss scale %g' % S)
            run(g, 'fp32', '', dict(attn_fwd='fp16'), 1.0, 'EMU fp16 attention core fwd, rest fp32')
            run(g, 'fp32', '', dict(lin_fwd='fp16'), 1.0, 'EMU fp16 linears fwd, rest fp32')
            run(g, 'fp32', '', dict(conv_fwd='fp16'), 1.0, 'EMU fp16 convs fwd, rest fp32')
            continue
        S = 4096.0
        if '--round5' in sys.argv:
            # two-product candidates for the data gradients that PROPAGATE (second session): the gradient operand as an fp16 hi + lo
            # pair, the weights as one fp16 value ('fp16w'); the leaves as shipped (conv / big-linear weight gradients and the d(d0)
            # half of final's data gradient on single fp16 products)
            LEAF = dict(conv_wgrad='fp16', lin_wgrad='fp16')
            BIG_ONLY[0] = True
            HALF[0] = 'both'
            run(g, 'bf16x3', 'fp32', dict(LEAF, final_dgrad=None), S, 'shipped leaves, propagating dgrads exact', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(LEAF, final_dgrad='fp16w'), S, '+ d(u0): weights fp16, dY exact', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(LEAF, final_dgrad='fp16a'), S, '+ d(u0): dY fp16, weights exact', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(LEAF, final_dgrad='fp16'), S, '+ d(u0): single fp16 product', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(LEAF, final_dgrad='fp16w', conv_dgrad='fp16w'), S, '+ all conv dgrads: weights fp16, dY exact', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(LEAF, final_dgrad='fp16w', conv_dgrad='fp16w', lin_dgrad='fp16w'), S, '+ conv and linear dgrads: weights fp16', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(LEAF, lin_dgrad='fp16w'), S, '+ linear dgrads only: weights fp16', attn_bwd='bf16x3')
            HALF[0] = None
            BIG_ONLY[0] = False
            continue
        if '--round4' in sys.argv:
            for half in (0, 1):
                HALF[0] = half
                run(g, 'bf16x3', 'fp32', dict(final_dgrad='fp16'), S, 'x3 fwd | fp16 final dgrad, %s half only' % ('d(d0)' if half == 0 else 'd(u0)'), attn_bwd='bf16x3')
                run(g, 'fp32', '', dict(final_dgrad='fp16'), S, 'fp32 | fp16 final dgrad, %s half only' % ('d(d0)' if half == 0 else 'd(u0)'))
            HALF[0] = None
            run(g, 'fp32', '', dict(final_dgrad='fp16'), S, 'fp32 | fp16 final dgrad (both halves)')
            continue
        if '--round3' in sys.argv:
            run(g, 'fp32', '', {}, 1.0, 'fp32 fwd / fp32 bwd')
            run(g, 'fp32', '', dict(attn_bwd='fp16'), S, 'fp32 | fp16 attention bwd only')
            run(g, 'fp32', '', dict(attn_fwd='fp16', attn_bwd='fp16'), S, 'fp16 attention fwd + bwd, rest fp32')
            run(g, 'fp32', '', dict(lin_dgrad='fp16'), S, 'fp32 | fp16 linear dgrads only')
            run(g, 'fp32', '', dict(lin_wgrad='fp16'), S, 'fp32 | fp16 linear wgrads only')
            run(g, 'fp32', '', dict(attn_bwd='fp16', lin_wgrad='fp16', conv_wgrad='fp16'), S, 'fp32 | fp16 attention bwd + all wgrads')
            BIG_ONLY[0] = True
            run(g, 'fp32', '', dict(lin_wgrad='fp16'), S, 'fp32 | fp16 linear wgrads, M >= 1024 only')
            run(g, 'fp32', '', dict(lin_wgrad='fp16', conv_wgrad='fp16'), S, 'fp32 | fp16 linear (M >= 1024) + conv wgrads')
            run(g, 'fp32', '', dict(lin_wgrad='fp16', conv_wgrad='fp16', attn_bwd='fp16', attn_fwd='fp16'), S, 'fp16 attention fwd+bwd, fp16 big-linear + conv wgrads')
            run(g, 'fp32', '', dict(lin_dgrad='fp16'), S, 'fp32 | fp16 linear dgrads, M >= 1024 only')
            BIG_ONLY[0] = False
            continue
        if '--round2' not in sys.argv:
            CW = dict(final_wgrad='fp16', up_wgrad='fp16')
            run(g, 'bf16x3', 'fp32', {}, 1.0, 'x3 fwd | fp32 conv+linear bwd, x3 attention bwd', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', CW, S, 'x3 fwd | fp16 final + up-conv weight gradients', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(conv_wgrad='fp16'), S, 'x3 fwd | fp16 all conv weight gradients', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(CW, final_dgrad='fp16'), S, 'x3 fwd | fp16 final + up-conv wgrads + final dgrad', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(final_dgrad='fp16'), S, 'x3 fwd | fp16 final dgrad only', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(conv_dgrad='fp16'), S, 'x3 fwd | fp16 all conv dgrads only', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(lin_wgrad='fp16'), S, 'x3 fwd | fp16 linear wgrads only', attn_bwd='bf16x3')
            run(g, 'bf16x3', 'fp32', dict(CW, final_dgrad='fp16'), 256.0, 'x3 fwd | fp16 final + up wgrads + final dgrad, scale 2^8', attn_bwd='bf16x3')
            run(g, 'fp32', '', dict(CW, final_dgrad='fp16'), S, 'fp32 fwd | fp16 final + up wgrads + final dgrad')
            continue
        W16 = dict(conv_wgrad='fp16', lin_wgrad='fp16')
        run(g, 'fp32', '', {}, 1.0, 'fp32 fwd / fp32 bwd')
        run(g, 'fp32', '', W16, S, 'fp32 | fp16 weight gradients (convs + linears)')
        run(g, 'fp32', '', dict(W16, conv_dgrad='fp16'), S, 'fp32 | fp16 weight gradients + conv data gradients')
        run(g, 'fp32', '', dict(W16, conv_dgrad='fp16', attn_bwd='fp16'), S, 'fp32 | fp16 wgrads + conv dgrads + attention bwd')
        run(g, 'fp32', '', dict(W16, conv_dgrad='fp16', lin_dgrad='fp16'), S, 'fp32 | fp16 wgrads + conv dgrads + linear dgrads')
        run(g, 'fp32', '', dict(W16, lin_dgrad='fp16'), S, 'fp32 | fp16 wgrads + linear dgrads')
        run(g, 'fp32', '', dict(conv_wgrad='bf16', lin_wgrad='bf16'), 1.0, 'fp32 | bf16 weight gradients')
        run(g, 'fp32', '', dict(conv_wgrad='bf16', lin_wgrad='bf16', conv_dgrad='bf16'), 1.0, 'fp32 | bf16 wgrads + conv dgrads')
        # the forward as shipped (bf16x3 kernels, fused attention); backward convs / linears on the emulation, attention bwd bf16x3
        run(g, 'bf16x3', 'fp32', {}, 1.0, 'bf16x3 fwd | fp32 conv+linear bwd, x3 attention bwd', attn_bwd='bf16x3')
        run(g, 'bf16x3', 'fp32', W16, S, 'bf16x3 fwd | fp16 wgrads, x3 attention bwd', attn_bwd='bf16x3')
        run(g, 'bf16x3', 'fp32', dict(W16, conv_dgrad='fp16'), S, 'bf16x3 fwd | fp16 wgrads + conv dgrads, x3 attention bwd', attn_bwd='bf16x3')
        run(g, 'bf16x3', 'fp32', dict(W16, conv_dgrad='fp16'), 16384.0, 'bf16x3 fwd | same, loss scale 2^14', attn_bwd='bf16x3')
