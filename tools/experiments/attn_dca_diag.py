"""Diagnostic (GPU, round 6): why does a single-fp16 q fail the F5c3 gates only together with a single-fp16 p or v, and only in the decoder
cross-attention?  Captures that attention's q, k | v on the F5c3 batch and prints operand statistics and the output error of each variant
(tools/experiments/attn_fwd_variants.py) against the exact product, split into the part along the keys' mean value and the rest."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tools.experiments.attn_fwd_variants as A          # noqa: E402  (patches flash.flash_attn_fwd_dl)
from voxactb_amd import flash                            # noqa: E402
import tests.test_c2_reference_gpu as T                 # noqa: E402

CAP = {}
_emu = flash.flash_attn_fwd_dl


def capture(q, kv, B, H, Nq, Nk, scale, p=0.0, seed=0, x3=False, return_planes=False):
    CAP[(H, Nq, Nk, len(CAP))] = (q.clone(), kv.clone(), B, H, Nq, Nk, scale)
    return _emu(q, kv, B, H, Nq, Nk, scale, p, seed, x3=x3, return_planes=return_planes)


flash.flash_attn_fwd_dl = capture
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'f5c3_encoder_c3_digest.npz'), allow_pickle=False)
A.VARIANT[0] = 'real'
enc, rs, grid, arm, V, B = T._setup(g)
eng = enc.engine()
eng.precision = 'bf16x3'
eng.forward(grid, rs['low_dim_state'].to(A.DEV), rs['lang_token_embs'].to(A.DEV), training=False, save=False, lang_goal_emb=rs['lang_goal_emb'].to(A.DEV))
for key, (q, kv, B, H, Nq, Nk, scale) in CAP.items():
    inner = H * 64
    qh = (q.view(B, Nq, H, 64).permute(0, 2, 1, 3) * scale).double()
    kh = kv[:, :inner].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).double()
    vh = kv[:, inner:].reshape(B, Nk, H, 64).permute(0, 2, 1, 3).double()
    S = qh @ kh.transpose(-1, -2)
    P = torch.softmax(S, -1)
    O = P @ vh
    vbar = vh.mean(-2, keepdim=True)
    kbar = kh.mean(-2, keepdim=True)
    print('H %d Nq %d Nk %d | |q*scale| rms %.3e max %.3e | |k| rms %.3e |kbar| %.3e | |v| rms %.3e |vbar| %.3e | S rms %.3e row spread (max-min) mean %.3e | P*Nk range [%.4f, %.4f] | |O - vbar| / |O| %.3e'
          % (H, Nq, Nk, float(qh.pow(2).mean().sqrt()), float(qh.abs().max()), float(kh.pow(2).mean().sqrt()), float(kbar.pow(2).mean().sqrt()),
             float(vh.pow(2).mean().sqrt()), float(vbar.pow(2).mean().sqrt()), float(S.pow(2).mean().sqrt()), float((S.max(-1).values - S.min(-1).values).mean()),
             float((P * Nk).min()), float((P * Nk).max()), float((O - vbar).norm() / O.norm())), flush=True)
    if not (H == 1 and Nq > Nk):
        continue
    for v in ('xx/xx', 'hh/hh', 'xh/hh', 'hh/xx', 'hh/xh', 'hh/hx', 'xx/hh', 'hx/hh'):
        (kq, kk), (kp, kvv) = A.parse(v)
        qf, kf, vf = qh.float(), kh.float(), vh.float()
        Sv = A.product(qf, kf, kq, kk)
        m = Sv.max(-1, keepdim=True).values
        Pv = torch.exp(Sv - m)
        l = Pv.sum(-1, keepdim=True)
        Ov = (A.product(Pv, vf.transpose(-1, -2), kp, kvv) / l).double()
        E = Ov - O
        dev = O - vbar                                                # the query-dependent part of the output
        # component of the error along the (query-dependent) deviation and its size relative to the deviation
        print('   %-6s |O_v - O| / |O| %.3e   / |O - vbar| %.3e   row-mean of the error / |vbar| %.3e   |S_v - S| rms %.3e  (row-constant part removed %.3e)'
              % (v, float(E.norm() / O.norm()), float(E.norm() / dev.norm()), float(E.mean(-2).norm() / vbar.norm()),
                 float((Sv.double() - S).pow(2).mean().sqrt()), float(((Sv.double() - S) - (Sv.double() - S).mean(-1, keepdim=True)).pow(2).mean().sqrt())), flush=True)
