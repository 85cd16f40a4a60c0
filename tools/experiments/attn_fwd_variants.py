"""Precision experiment (GPU, round 6): which operand of the attention FORWARD's two products needs more than one fp16 value?

Round 5 left the single-fp16 pipelined forward as a named mode because the F5c3 digest's element gates (0.3 % of a small tensor's maximum)
are missed by 2 x.  Before writing a kernel with an arithmetic between one and three products per term, this script measures the
candidates on the REFERENCE digests: everything else runs as shipped (default precision, fp16 pipelined backward, backward evaluated at the
reference run's LeakyReLU choices) and only the forward of the attention core is replaced by an emulation built from exact fp32 matrix
products on operands rounded the way the candidate rounds them (the product of two 11-bit mantissas is exact in fp32).

    python tools/experiments/attn_fwd_variants.py [f5c3] [f5g] [--variants a,b,c]

A variant names how each of q, k (scores) and p, v (output) enters its product:
    x = exact (what a hi + lo pair carries: 22 bits), h = one fp16 value, c<n> = fp16 hi plus the two cross terms hi * lo with BOTH factors
    of the cross terms cut to n explicit mantissa bits (what a low-precision MFMA at 2 x / 4 x rate would contribute), m = mean over the
    keys removed first (softmax / the weighted mean are invariant), then one fp16 value
written 'qk/pv' as e.g. 'hh/hh' (round 5's single-fp16 forward), 'xh/hh' (q as hi + lo), 'xx/hh', 'c3c3/hh'; a suffix '@self' / '@cross'
applies the variant to the 8-head self-attention layers / the 1-head cross attentions only (the others stay exact).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from voxactb_amd import flash, ops                              # noqa: E402
import tests.test_c2_reference_gpu as T                        # noqa: E402

DEV = 'cuda:0'
VARIANT = ['xx/xx']
_real_fwd_dl = flash.flash_attn_fwd_dl


def f16(t):
    return t.half().float()


def cut(t, mb):
    """round to mb explicit mantissa bits (round to nearest, ties away: good enough for an error estimate)"""
    b = t.contiguous().view(torch.int32)
    sh = 23 - mb
    b = (b + (1 << (sh - 1))) & ~((1 << sh) - 1)
    return b.view(torch.float32)


def product(a, b, ka, kb, keys_dim_b=-2):
    """a [.., M, K] x b [.., N, K]^T with operand treatments ka, kb ('m': the mean of b over its key dimension is removed first)"""
    def parts(t, k):
        if k == 'm':
            t = t - t.mean(dim=keys_dim_b, keepdim=True)
            k = 'h'
        if k == 'x':
            return t, None, None
        hi = f16(t)
        if k == 'h':
            return hi, None, None
        mb = int(k[1:])
        return hi, cut(f16(t - hi), mb), cut(hi, mb)
    ah, al, ahc = parts(a, ka)
    bh, bl, bhc = parts(b, kb)
    out = ah @ bh.transpose(-1, -2)
    if al is not None:
        out = out + al @ (bhc if bhc is not None else bh).transpose(-1, -2)
    if bl is not None:
        out = out + (ahc if ahc is not None else ah) @ bl.transpose(-1, -2)
    return out


def parse(v):
    qk, pv = v.split('/')
    def two(s):
        out, i = [], 0
        while i < len(s):
            if s[i] == 'c':
                j = i + 1
                while j < len(s) and s[j].isdigit():
                    j += 1
                out.append(s[i:j]); i = j
            else:
                out.append(s[i]); i += 1
        assert len(out) == 2, s
        return out
    return two(qk), two(pv)


def emu_fwd_dl(q, kv, B, H, Nq, Nk, scale, p=0.0, seed=0, x3=False, return_planes=False):
    v = VARIANT[0]
    where = ''
    if '@' in v:
        v, where = v.split('@')
    if v == 'real' or (where == 'self' and H == 1) or (where == 'cross' and H != 1):
        return _real_fwd_dl(q, kv, B, H, Nq, Nk, scale, p, seed, x3=x3, return_planes=return_planes)
    if v == 'realf16':                                                    # round 5's pipelined single-fp16 kernel itself
        O, lse = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, scale, p, seed, mode='f16')
        return (O, lse, None) if return_planes else (O, lse)
    assert p == 0.0
    (kq, kk), (kp, kvv) = parse(v)
    inner = H * 64
    qh = q.view(B, Nq, H, 64).permute(0, 2, 1, 3) * scale               # [B, H, Nq, 64]  (the kernel folds scale * log2e into q)
    kh = kv[:, :inner].reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    vh = kv[:, inner:].reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    S = product(qh, kh, kq, kk)                                          # [B, H, Nq, Nk]
    m = S.max(dim=-1, keepdim=True).values
    P = torch.exp(S - m)
    l = P.sum(dim=-1, keepdim=True)
    if kvv == 'm':
        vbar = vh.mean(dim=-2, keepdim=True)
        O = product(P, (vh - vbar).transpose(-1, -2), kp, 'h') / l + vbar
    else:
        O = product(P, vh.transpose(-1, -2), kp, kvv) / l               # b = V^T [64, Nk]: keys are its last dim -> treat per element
    lse = (m + torch.log(l)).reshape(B * H, Nq).contiguous()
    O = O.permute(0, 2, 1, 3).reshape(B * Nq, inner).contiguous()
    if return_planes:
        return O, lse, None
    return O, lse


flash.flash_attn_fwd_dl = emu_fwd_dl


POOLS = {}


def run(g, variant, tag):
    force_pools = variant.endswith('+pools')          # backward at the max-pool arg-maxima of the 'real' run (run 'real' first)
    variant = variant.replace('+pools', '')
    VARIANT[0] = variant
    enc, rs, grid, arm, V, B = T._setup(g)
    eng = enc.engine()
    eng.precision = 'bf16x3'
    outs, cache = eng.forward(grid, rs['low_dim_state'].to(DEV), rs['lang_token_embs'].to(DEV), training=False, save=True,
                              lang_goal_emb=rs['lang_goal_emb'].to(DEV))
    flat = outs[0].reshape(B, -1).float().cpu()
    sidx = T.T(g['q_trans_sample_idx']).long()
    e_q = float((flat[:, sidx] - T.T(g['q_trans_sample'])).abs().max())
    e_r = float((outs[1].float().cpu() - T.T(g['rot_grip'])).abs().max())
    e_c = float((outs[2].float().cpu() - T.T(g['collision'])).abs().max())
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
    flips = T.force_kinks(cache, g) if 'kink_tau' in g.files else -1
    pool_diff = []
    for key in ('ss0', 'ss1', 'ss2'):
        ss, mx, st, am = cache[key]
        if variant == 'real':
            POOLS[(tag, key)] = am.clone()
        elif (tag, key) in POOLS:
            pool_diff.append(int((am != POOLS[(tag, key)]).sum()))
            if force_pools:
                cache[key] = (ss, mx, st, POOLS[(tag, key)])
    eng.backward(cache, dq, d_o, d_arm)
    P = dict(enc.named_parameters())
    rows = []
    for n, rn in zip([str(n) for n in g['grad_names']], T.T(g['grad_norms'])):
        if P[n].grad is None or ('dysum64__' + n) in g.files:
            continue
        gn, rn = float(P[n].grad.norm()), float(rn)
        if rn > 1e-4:
            rows.append((abs(gn - rn) / (3e-3 * rn + 1e-5), 'norm', n))
        key = 'grad__' + n
        if key in g.files:
            ref = T.T(g[key])
            e = float((P[n].grad.float().cpu() - ref).abs().max())
            rows.append((e / (3e-3 * float(ref.abs().max()) + 1e-5), 'elem', n))
    rows.sort(reverse=True)
    print('%-22s %-5s q %.2e rot %.2e col %.2e loss err %.2e flips %3d pools %s | worst x gate: %s' % (
        variant + ('+pools' if force_pools else ''), tag, e_q, e_r, e_c, abs(loss - float(g['loss'])), flips, pool_diff,
        '  '.join('%.2f %s %s' % (r[0], r[1], r[2].replace('cross_attend_blocks', 'cab').replace('decoder_cross_attn', 'dca')) for r in rows[:4])), flush=True)
    del cache, outs
    torch.cuda.empty_cache()


if __name__ == '__main__':
    fixtures = [a for a in sys.argv[1:] if not a.startswith('-')] or ['f5c3']
    names = {'f5g': 'f5g_encoder_c2_grads', 'f5c3': 'f5c3_encoder_c3_digest', 'f5gb8': 'f5gb8_encoder_c2_b8_grads'}
    variants = ['real', 'xx/xx', 'hh/hh', 'xx/hh', 'hh/xx', 'xh/hh', 'hx/hh', 'xh/xh', 'xh/hx', 'hh/xh', 'hh/hx', 'xx/xh', 'xx/hx',
                'hh/hh@self', 'hh/hh@cross', 'c3c3/hh', 'c3c3/c3c3', 'c1c1/c1c1', 'c3c3/hc3', 'xc3/hh', 'hm/hm', 'xm/hm', 'realf16', 'realf16@self', 'realf16@cross', 'xh/hh@self', 'xx/hh@self']
    for a in sys.argv[1:]:
        if a.startswith('--variants='):
            variants = a.split('=', 1)[1].split(',')
    for f in fixtures:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', names[f] + '.npz'), allow_pickle=False)
        print('== %s' % f, flush=True)
        for v in variants:
            try:
                run(g, v, f)
            except Exception as e:                                   # noqa: BLE001
                print('%-22s FAILED %r' % (v, e), flush=True)
