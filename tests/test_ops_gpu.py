"""GPU parity of every fp32 kernel (called through the C ABI via voxactb_amd.ops) against a plain PyTorch
CPU fp32 reference of the same op.  Tolerances are written next to each check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from voxactb_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


def close(a, b, tol, what=''):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float((a - b).abs().max())
    ref = float(b.abs().max()) + 1e-12
    assert err <= tol * max(ref, 1.0), '%s: max abs err %.3e (ref max %.3e)' % (what, err, ref)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 72, 100), (77, 130, 36), (300, 64, 512), (5, 220, 64)])
def test_gemm_nt_bias_act_residual(M, N, K):
    x, W, b, r = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(x.to(DEV), W.to(DEV), out, M, N, K, K, 1, 1, K, N, bias=b.to(DEV), residual=r.to(DEV), act=ops.ACT_LRELU)
    close(out, F.leaky_relu(x @ W.t() + b, 0.02) + r, 2e-5, 'nt')
    out2 = out.clone()
    ops.gemm(x.to(DEV), W.to(DEV), out2, M, N, K, K, 1, 1, K, N, alpha=0.5, accumulate=True)
    close(out2, out.cpu() + 0.5 * (x @ W.t()), 2e-5, 'nt-acc')


def test_gemm_nn_tn_and_linear_bwd():
    M, N, K = 260, 96, 132
    x, W, dy = rnd(M, K), rnd(N, K, seed=1), rnd(M, N, seed=2)
    dW = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    dx = torch.empty(M, K, device=DEV)
    ops.linear_bwd(x.to(DEV), W.to(DEV), dy.to(DEV), dW, db, dx)
    close(dx, dy @ W, 2e-5, 'dx')
    close(dW, dy.t() @ x, 2e-5, 'dW')
    close(db, dy.sum(0), 2e-5, 'db')
    # tiny / unaligned shapes go through the naive kernel
    x, W, dy = rnd(16, 7), rnd(64, 7, seed=1), rnd(16, 64, seed=2)
    y = ops.linear(x.to(DEV), W.to(DEV), None, ops.ACT_LRELU)
    close(y, F.leaky_relu(x @ W.t(), 0.02), 1e-5, 'naive fwd')
    dW = torch.zeros(64, 7, device=DEV)
    dx = torch.empty(16, 7, device=DEV)
    ops.linear_bwd(x.to(DEV), W.to(DEV), dy.to(DEV), dW, None, dx)
    close(dW, dy.t() @ x, 1e-5, 'naive dW')
    close(dx, dy @ W, 1e-5, 'naive dx')


def test_gemm_batched_heads_padded_k():
    # attention-shaped: S = scale * Q K^T per (b, h); O = P V with K = 77 (row stride padded to 80, zero padding)
    B, H, L, J, d = 2, 3, 70, 77, 64
    q, kv = rnd(B, L, H * d), rnd(B, J, 2 * H * d, seed=1)
    ld = 80
    S = torch.zeros(B * H, L, ld, device=DEV)
    qd, kvd = q.to(DEV), kv.to(DEV)
    ops.gemm(qd, kvd, S, L, J, d, H * d, 1, 1, 2 * H * d, ld, batch=B * H, H=H, bA=(L * H * d, d), bB=(J * 2 * H * d, d),
             bC=(H * L * ld, L * ld), alpha=0.125)
    k = kv[..., :H * d].view(B, J, H, d).permute(0, 2, 1, 3)
    v = kv[..., H * d:].view(B, J, H, d).permute(0, 2, 1, 3)
    qh = q.view(B, L, H, d).permute(0, 2, 1, 3)
    ref = torch.einsum('bhid,bhjd->bhij', qh, k) * 0.125
    close(S.view(B, H, L, ld)[..., :J], ref, 2e-5, 'qk')
    P = ops.softmax_rows(S, B * H * L, J, ld)
    close(P.view(B, H, L, ld)[..., :J], ref.softmax(-1), 1e-5, 'softmax')
    assert float(P.view(B, H, L, ld)[..., J:].abs().max()) == 0.0
    O = torch.empty(B, L, H * d, device=DEV)
    ops.gemm(P, kvd[..., H * d:], O, L, d, J, ld, 1, 2 * H * d, 1, H * d, batch=B * H, H=H, bA=(H * L * ld, L * ld),
             bB=(J * 2 * H * d, d), bC=(L * H * d, d))
    refO = torch.einsum('bhij,bhjd->bhid', ref.softmax(-1), v).permute(0, 2, 1, 3).reshape(B, L, H * d)
    close(O, refO, 2e-5, 'pv')


# ------------------------------------------------------------------------------------------------ conv family
def cl(x):   # [B,C,D,H,W] -> channels-last [B,D,H,W,C] contiguous
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ref_conv(x, W, b, stride=1):
    p = W.shape[-1] // 2
    return F.conv3d(F.pad(x, (p,) * 6, mode='replicate'), W, b, stride=stride)


@pytest.mark.parametrize('Cin,Cout,k,s,S', [(64, 64, 3, 1, 6), (128, 64, 5, 1, 5), (64, 64, 5, 5, 10), (64, 64, 5, 4, 8),
                                             (64, 64, 3, 2, 8), (16, 128, 3, 1, 4)])
def test_conv3d_fwd_dgrad_wgrad(Cin, Cout, k, s, S):
    B = 2
    x = rnd(B, Cin, S, S, S).requires_grad_(True)
    W = rnd(Cout, Cin, k, k, k, seed=1, scale=0.1).requires_grad_(True)
    b = rnd(Cout, seed=2)
    y_ref = F.leaky_relu(ref_conv(x, W, b, s), 0.02)
    G = y_ref.shape[-1]
    p = k // 2
    xd, Wd = cl(x.detach()).to(DEV), W.detach().to(DEV)
    y = ops.conv3d(xd, ops.conv_weight_fwd(Wd), Cout, B, S, G, k, -p, stride=s, bias=b.to(DEV), act=ops.ACT_LRELU)
    close(y, cl(y_ref), 2e-5, 'fwd')
    dy = rnd(B, Cout, G, G, G, seed=3)
    (y_ref * dy).sum().backward()
    dyd = cl(dy).to(DEV)
    dpre = ops.lrelu_bwd_(dyd.clone(), y)
    # weight gradient
    dWt = ops.conv3d_wgrad(xd, dpre, Cout, B, S, G, k, -p, stride=s, nsplit=3)
    dW = dWt.view(k ** 3, Cin, Cout).permute(2, 1, 0).reshape(Cout, Cin, k, k, k)
    close(dW, W.grad, 3e-5, 'wgrad')
    # data gradient
    dx = torch.empty(B, S, S, S, Cin, device=DEV)
    if s == 1:
        Sp = S + 2 * p
        dxp = ops.conv3d(dpre, ops.conv_weight_dgrad(Wd), Cin, B, G, Sp, k, -(k - 1), replicate=False)
        ops.fold_pad(dxp, Sp, Cin, 0, dx, B, S, Cin, p)
    else:
        wt, U = ops.strided_dgrad_weights(Wd, s)
        Gp = (S + 2 * p + s - 1) // s
        dxp = ops.conv3d(dpre, wt, s ** 3 * Cin, B, G, Gp, U, -(U - 1), replicate=False, d2s=(s, Cin))
        ops.fold_pad(dxp, Gp * s, Cin, 0, dx, B, S, Cin, p)
    close(dx, cl(x.grad), 3e-5, 'dgrad')


def test_conv3d_two_sources():
    B, S = 2, 5
    a, c = rnd(B, 64, S, S, S), rnd(B, 64, S, S, S, seed=5)
    W, b = rnd(64, 128, 3, 3, 3, seed=1, scale=0.1), rnd(64, seed=2)
    ref = F.leaky_relu(ref_conv(torch.cat([a, c], 1), W, b), 0.02)
    y = ops.conv3d(cl(a).to(DEV), ops.conv_weight_fwd(W.to(DEV)), 64, B, S, S, 3, -1, bias=b.to(DEV), act=ops.ACT_LRELU,
                   src1=cl(c).to(DEV))
    close(y, cl(ref), 2e-5, 'two-source fwd')
    dy = cl(rnd(B, 64, S, S, S, seed=7)).to(DEV)
    dWt = ops.conv3d_wgrad(cl(a).to(DEV), dy, 64, B, S, S, 3, -1, src1=cl(c).to(DEV), nsplit=2)
    xx = torch.cat([a, c], 1).requires_grad_(True)
    Wr = W.clone().requires_grad_(True)
    (ref_conv(xx, Wr, b) * dy.cpu().permute(0, 4, 1, 2, 3)).sum().backward()
    close(dWt.view(27, 128, 64).permute(2, 1, 0).reshape(64, 128, 3, 3, 3), Wr.grad, 3e-5, 'two-source wgrad')
    dxp = ops.conv3d(dy, ops.conv_weight_dgrad(W.to(DEV)), 128, B, S, S + 2, 3, -2, replicate=False)
    d0 = torch.empty(B, S, S, S, 64, device=DEV)
    d1 = torch.empty(B, S, S, S, 64, device=DEV)
    ops.fold_pad(dxp, S + 2, 128, 0, d0, B, S, 64, 1)
    ops.fold_pad(dxp, S + 2, 128, 64, d1, B, S, 64, 1)
    close(d0, cl(xx.grad[:, :64]), 3e-5, 'two-source dgrad a')
    close(d1, cl(xx.grad[:, 64:]), 3e-5, 'two-source dgrad b')


@pytest.mark.parametrize('k,s,G', [(5, 5, 4), (5, 4, 3), (3, 2, 4)])
def test_polyphase_upconv(k, s, G):
    """upsample(x s, trilinear) o conv(k) == conv(kl, replicate) with s^3*Cout phase channels + depth-to-space."""
    B, C = 2, 64
    z1 = rnd(B, C, G, G, G).requires_grad_(True)
    W = rnd(C, C, k, k, k, seed=1, scale=0.1).requires_grad_(True)
    b = rnd(C, seed=2)
    up = F.interpolate(z1, scale_factor=s, mode='trilinear', align_corners=False)
    ref = F.leaky_relu(ref_conv(up, W, b), 0.02)
    L, R = ops.polyphase_tables(k, s)
    kl = 2 * R + 1
    Ld = torch.from_numpy(L).to(DEV)
    Wdev = W.detach().to(DEV)
    Weff = ops.polyphase_weights(Wdev, Ld, s, kl)
    z1d = cl(z1.detach()).to(DEV)
    bias_e = b.to(DEV).repeat(s ** 3)
    u0 = ops.conv3d(z1d, Weff, s ** 3 * C, B, G, G, kl, -R, bias=bias_e, act=ops.ACT_LRELU, d2s=(s, C))
    close(u0, cl(ref), 3e-5, 'polyphase fwd')
    V = G * s
    dy = rnd(B, C, V, V, V, seed=3)
    (ref * dy).sum().backward()
    dpre = ops.lrelu_bwd_(cl(dy).to(DEV), u0)
    dWeff = ops.conv3d_wgrad(z1d, dpre, s ** 3 * C, B, G, G, kl, -R, d2s=(s, C), nsplit=2)
    dW = torch.zeros_like(Wdev)
    ops.polyphase_weights_bwd(dWeff, Ld, dW, s, kl)
    close(dW, W.grad, 5e-5, 'polyphase wgrad')
    wd = ops.polyphase_dgrad_weights(Weff, C, C, s, kl)
    Sp = G + 2 * R
    dzp = ops.conv3d(dpre, wd, C, B, V, Sp, s * kl, -s * (kl - 1), stride=s, replicate=False)
    dz = torch.empty(B, G, G, G, C, device=DEV)
    ops.fold_pad(dzp, Sp, C, 0, dz, B, G, C, R)
    close(dz, cl(z1.grad), 5e-5, 'polyphase dgrad')


def test_pointwise_and_c1():
    B, S = 2, 6
    x = rnd(B, 10, S, S, S)
    W, b = rnd(64, 10, 1, 1, 1, seed=1).requires_grad_(True), rnd(64, seed=2).requires_grad_(True)
    ref = F.leaky_relu(F.conv3d(x, W, b), 0.02)
    xd = cl(x).to(DEV)
    y = ops.pointwise_fwd(xd, W.detach().view(64, 10).to(DEV), b.detach().to(DEV))
    close(y, cl(ref), 1e-5, 'pointwise')
    dy = rnd(B, 64, S, S, S, seed=3)
    (ref * dy).sum().backward()
    dW, db = torch.zeros(64, 10, device=DEV), torch.zeros(64, device=DEV)
    ops.pointwise_wgrad(xd, y, cl(dy).to(DEV), dW, db)
    close(dW, W.grad.view(64, 10), 3e-5, 'pointwise dW')
    close(db, b.grad, 3e-5, 'pointwise db')
    # one-output-channel 3x3x3 conv
    u = rnd(B, 64, S, S, S, seed=4).requires_grad_(True)
    w1, b1 = rnd(1, 64, 3, 3, 3, seed=5, scale=0.1).requires_grad_(True), rnd(1, seed=6).requires_grad_(True)
    q_ref = ref_conv(u, w1, b1)
    ud = cl(u.detach()).to(DEV)
    q = ops.conv3_c1_fwd(ud, w1.detach().to(DEV), b1.detach().to(DEV), B, S)
    close(q, q_ref[:, 0], 2e-5, 'c1 fwd')
    dq = rnd(B, S, S, S, seed=7)
    (q_ref[:, 0] * dq).sum().backward()
    du = torch.zeros(B, S, S, S, 64, device=DEV)
    ops.conv3_c1_dgrad(dq.to(DEV), w1.detach().to(DEV), ud, du, B, S, accumulate=True, mask=False)
    close(du, cl(u.grad), 2e-5, 'c1 dgrad')
    dw, dbb = torch.zeros(1, 64, 3, 3, 3, device=DEV), torch.zeros(1, device=DEV)
    ops.conv3_c1_wgrad(ud, dq.to(DEV), dw, dbb, B, S)
    close(dw, w1.grad, 3e-5, 'c1 wgrad')
    close(dbb, b1.grad, 3e-5, 'c1 db')


def ref_ss3d(x):
    from oracle import perceiver as operc
    return operc.spatial_softmax3d(x), x.amax(dim=(2, 3, 4))


@pytest.mark.parametrize('C,S,tokens0', [(64, 7, 0), (128, 4, 5)])
def test_ss3d_max(C, S, tokens0):
    B = 3
    x = (rnd(B, C, S, S, S) * 0.05).requires_grad_(True)
    ss_ref, mx_ref = ref_ss3d(x)
    # the product reads channels-last with a batch stride (decoder output has `tokens0` language rows in front)
    buf = torch.zeros(B, tokens0 + S ** 3, C)
    buf[:, tokens0:] = cl(x.detach()).view(B, -1, C)
    bd = buf.to(DEV)
    view = bd[:, tokens0:]
    ss, mx, stats, arg = ops.ss3d_max_fwd(view, bd.stride(0), B, S, C)
    close(ss, ss_ref, 2e-5, 'ss3d')
    close(mx, mx_ref, 0, 'max')
    g_ss, g_mx = rnd(B, 3 * C, seed=1), rnd(B, C, seed=2)
    ((ss_ref * g_ss).sum() + (mx_ref * g_mx).sum()).backward()
    dx = torch.zeros(B, S ** 3, C, device=DEV)
    ops.ss3d_max_bwd(view, bd.stride(0), B, S, C, stats, ss, arg, g_ss.to(DEV), g_mx.to(DEV), dx, S ** 3 * C)
    close(dx, cl(x.grad).view(B, -1, C), 1e-4, 'ss3d bwd')


def test_layernorm_geglu_softmax_bwd():
    rows, D = 300, 512
    x = rnd(rows, D).requires_grad_(True)
    g, b = (1 + 0.1 * rnd(D, seed=1)).requires_grad_(True), (0.1 * rnd(D, seed=2)).requires_grad_(True)
    ref = F.layer_norm(x, (D,), g, b, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x.detach().to(DEV), g.detach().to(DEV), b.detach().to(DEV))
    close(y, ref, 1e-5, 'ln fwd')
    dy = rnd(rows, D, seed=3)
    (ref * dy).sum().backward()
    dg, dbt = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dx = ops.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), g.detach().to(DEV), mean, rstd, dg, dbt)
    close(dx, x.grad, 2e-5, 'ln dx')
    close(dg, g.grad, 3e-5, 'ln dgamma')
    close(dbt, b.grad, 3e-5, 'ln dbeta')
    # D = 128 variant with accumulate
    x2 = rnd(70, 128)
    g2, b2 = 1 + 0.1 * rnd(128, seed=1), 0.1 * rnd(128, seed=2)
    y2, m2, r2 = ops.layernorm_fwd(x2.to(DEV), g2.to(DEV), b2.to(DEV))
    close(y2, F.layer_norm(x2, (128,), g2, b2, 1e-5), 1e-5, 'ln128')
    # GEGLU
    h = rnd(100, 256).requires_grad_(True)
    a, gt = h.chunk(2, -1)
    refg = a * F.gelu(gt)
    og = ops.geglu_fwd(h.detach().to(DEV))
    close(og, refg, 1e-5, 'geglu fwd')
    dgo = rnd(100, 128, seed=5)
    (refg * dgo).sum().backward()
    close(ops.geglu_bwd(h.detach().to(DEV), dgo.to(DEV)), h.grad, 2e-5, 'geglu bwd')
    # softmax backward (no dropout) and dropout bookkeeping
    S = rnd(64, 100).requires_grad_(True)
    P_ref = (S * 0.5).softmax(-1)
    dP = rnd(64, 100, seed=9)
    (P_ref * dP).sum().backward()
    Sd = (S.detach() * 0.5).to(DEV).contiguous()
    P = ops.softmax_rows(Sd, 64, 100, 100)
    dS = ops.softmax_bwd_rows(P, dP.to(DEV).clone(), 64, 100, 100, 0.5)
    close(dS, S.grad, 2e-5, 'softmax bwd')
    Sd2 = (S.detach() * 0.5).to(DEV).contiguous()
    Pd = ops.softmax_rows(Sd2, 64, 100, 100, p=0.25, seed=123)
    keep = (Pd != 0)
    frac = float(keep.float().mean())
    assert 0.70 < frac < 0.80, frac
    close(Pd, torch.where(keep.cpu(), P.cpu() / 0.75, torch.zeros(())), 1e-6, 'dropout scaling')
    dS2 = ops.softmax_bwd_rows(Sd2, dP.to(DEV).clone(), 64, 100, 100, 0.5, p=0.25, seed=123)
    Pc = P.cpu()
    gk = torch.where(keep.cpu(), dP / 0.75, torch.zeros(()))
    close(dS2, 0.5 * Pc * (gk - (gk * Pc).sum(-1, keepdim=True)), 2e-5, 'softmax+dropout bwd')


def test_ctx_and_ce():
    B, T0, T1, C = 2, 5, 27, 64
    lang, patch, pp, pos = rnd(B * T0, 2 * C), rnd(B * T1, C, seed=1), rnd(B, C, seed=2), rnd(T0 + T1, 2 * C, seed=3)
    ctx = ops.ctx_build(lang.to(DEV), patch.to(DEV), pp.to(DEV), pos.to(DEV), B, T0, T1, C)
    tok = torch.cat([patch.view(B, T1, C), pp.view(B, 1, C).expand(B, T1, C)], -1)
    ref = torch.cat([lang.view(B, T0, 2 * C), tok], 1) + pos
    close(ctx, ref, 1e-6, 'ctx')
    dctx = rnd(B, T0 + T1, 2 * C, seed=4)
    dpos = torch.zeros(T0 + T1, 2 * C, device=DEV)
    dl, dpa, dpp = ops.ctx_bwd(dctx.to(DEV), dpos, B, T0, T1, C)
    close(dl, dctx[:, :T0].reshape(B * T0, 2 * C), 1e-6, 'dlang')
    close(dpa, dctx[:, T0:, :C].reshape(B * T1, C), 1e-6, 'dpatch')
    close(dpp, dctx[:, T0:, C:].sum(1), 1e-5, 'dpp')
    close(dpos, dctx.sum(0), 1e-5, 'dpos')
    # big CE
    P = 200000
    x = rnd(3, P).requires_grad_(True)
    lab = torch.tensor([5, 199999, 70000])
    ref_l = F.cross_entropy(x, lab, reduction='none')
    (ref_l.sum() / 3).backward()
    dx = torch.empty(3, P, device=DEV)
    loss, lse, arg = ops.ce_big(x.detach().to(DEV), lab.int().to(DEV), dx, 1.0 / 3)
    close(loss, ref_l, 1e-5, 'ce big loss')
    assert arg.cpu().tolist() == x.argmax(1).tolist()
    close(dx, x.grad, 1e-6, 'ce big grad')
    # small heads
    lg = rnd(4, 220).requires_grad_(True)
    labs = torch.tensor([[3, 71, 0, 1, 0], [10, 20, 30, 0, 1], [0, 0, 0, 0, 0], [71, 5, 9, 1, 1]])
    segs = [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)]
    ref = torch.stack([F.cross_entropy(lg[:, c0:c0 + n], labs[:, i], reduction='none') for i, (c0, n) in enumerate(segs)], 1)
    (ref.sum() / 4).backward()
    dl = torch.zeros(4, 220, device=DEV)
    loss, pred = ops.ce_rows(lg.detach().to(DEV), segs, labs.int().to(DEV), dl, 0.25)
    close(loss, ref, 1e-5, 'ce rows')
    close(dl, lg.grad, 1e-6, 'ce rows grad')
    assert pred.cpu().tolist() == [[int(lg[r, c0:c0 + n].argmax()) for (c0, n) in segs] for r in range(4)]


# ------------------------------------------------------------------------------------------------ bf16 matrix-core mode
def bf(x):
    return x.bfloat16().float()


@pytest.mark.parametrize('M,N,K', [(256, 128, 64), (300, 72, 96), (1000, 512, 2048), (77, 64, 512)])
def test_gemm_bf16w(M, N, K):
    x, W, b, r = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    out = ops.gemm_bf16w(x.to(DEV), W.to(DEV).to(torch.bfloat16), bias=b.to(DEV), act=ops.ACT_LRELU, residual=r.to(DEV))
    # the kernel must equal an fp32 GEMM of the bf16-ROUNDED operands (fp32 accumulate) ...
    close(out, F.leaky_relu(bf(x) @ bf(W).t() + b, 0.02) + r, 2e-5, 'bf16 gemm (rounded-operand reference)')
    # ... and stay within bf16 rounding of the true fp32 result
    close(out, F.leaky_relu(x @ W.t() + b, 0.02) + r, 2e-2, 'bf16 gemm vs fp32')


@pytest.mark.parametrize('Cin,Cout,k,s,S', [(64, 64, 3, 1, 6), (128, 64, 5, 1, 5), (64, 64, 5, 5, 10), (64, 128, 3, 1, 4)])
def test_conv3d_bf16w(Cin, Cout, k, s, S):
    B = 2
    x = rnd(B, Cin, S, S, S)
    W = rnd(Cout, Cin, k, k, k, seed=1, scale=0.1)
    b = rnd(Cout, seed=2)
    ref = F.leaky_relu(ref_conv(bf(x), bf(W), b, s), 0.02)
    G = ref.shape[-1]
    wb = ops.to_bf16_nk(ops.conv_weight_fwd(W.to(DEV)))
    y = ops.conv3d_bf16w(cl(x).to(DEV), wb, Cout, B, S, G, k, -(k // 2), stride=s, bias=b.to(DEV), act=ops.ACT_LRELU)
    close(y, cl(ref), 3e-5, 'bf16 conv fwd')
    # data gradient through the same kernel (zero padding, flipped weights) + two-source + depth-to-space variants
    if s == 1:
        dy = rnd(B, Cout, G, G, G, seed=3)
        wd = ops.to_bf16_nk(ops.conv_weight_dgrad(W.to(DEV)))
        p = k // 2
        dxp = ops.conv3d_bf16w(cl(dy).to(DEV), wd, Cin, B, G, S + 2 * p, k, -(k - 1), replicate=False)
        dxp_ref = ops.conv3d(cl(bf(dy)).to(DEV), ops.conv_weight_dgrad(bf(W).to(DEV)), Cin, B, G, S + 2 * p, k, -(k - 1), replicate=False)
        close(dxp, dxp_ref, 3e-5, 'bf16 conv dgrad')


def test_conv3d_bf16w_two_sources_and_d2s():
    B, S = 2, 5
    a, c = rnd(B, 64, S, S, S), rnd(B, 64, S, S, S, seed=5)
    W, b = rnd(64, 128, 3, 3, 3, seed=1, scale=0.1), rnd(64, seed=2)
    ref = F.leaky_relu(ref_conv(torch.cat([bf(a), bf(c)], 1), bf(W), b), 0.02)
    y = ops.conv3d_bf16w(cl(a).to(DEV), ops.to_bf16_nk(ops.conv_weight_fwd(W.to(DEV))), 64, B, S, S, 3, -1, bias=b.to(DEV),
                         act=ops.ACT_LRELU, src1=cl(c).to(DEV))
    close(y, cl(ref), 3e-5, 'bf16 two-source')
    # polyphase forward in bf16 == fp32 kernel on rounded operands
    k, s, G = 5, 5, 3
    z1 = rnd(B, 64, G, G, G, seed=7)
    W2 = rnd(64, 64, k, k, k, seed=8, scale=0.1)
    L, R = ops.polyphase_tables(k, s)
    kl = 2 * R + 1
    Weff = ops.polyphase_weights(W2.to(DEV), torch.from_numpy(L).to(DEV), s, kl)
    u_ref = ops.conv3d(cl(bf(z1)).to(DEV), bf(Weff), s ** 3 * 64, B, G, G, kl, -R, d2s=(s, 64))
    u = ops.conv3d_bf16w(cl(z1).to(DEV), ops.to_bf16_nk(Weff), s ** 3 * 64, B, G, G, kl, -R, d2s=(s, 64))
    close(u, u_ref, 3e-5, 'bf16 polyphase d2s')


@pytest.mark.parametrize('rows,N,ld', [(300, 64, 64), (1000, 4096, 4096), (777, 1280, 1284), (513, 258, 258), (4100, 512, 512)])
def test_colsum_variants(rows, N, ld):
    """bias gradients: column sums of [rows, N] (flat, scalar and float4 partial kernels) against a float64 sum."""
    buf = rnd(rows, ld, seed=rows)
    x = buf.to(DEV)[:, :N]
    out = torch.full((N,), 0.25, device=DEV)
    ops.colsum(x, out, accumulate=True)
    ref = buf[:, :N].double().sum(0) + 0.25
    assert float((out.cpu().double() - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-5
