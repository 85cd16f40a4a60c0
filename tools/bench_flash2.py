"""Round-4 attention kernels (csrc/flash2_*.hip): correctness against an fp64 reference and the round-3 kernels, then timing of
every (mode, waves, dropout) variant at the step's shapes.  GPU box: python tools/bench_flash2.py [--quick]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxactb_amd import flash  # noqa: E402


def ref_attn(q, kv, B, H, Nq, Nk, scale):
    """fp64 softmax attention (no dropout) -> o [B*Nq, H*64], lse [B*H, Nq]"""
    q4 = q.double().view(B, Nq, H, 64).permute(0, 2, 1, 3)
    k4 = kv.double()[:, :H * 64].reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    v4 = kv.double()[:, H * 64:].reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    s = torch.einsum('bhid,bhjd->bhij', q4, k4) * scale
    lse = torch.logsumexp(s, -1)
    o = torch.einsum('bhij,bhjd->bhid', torch.softmax(s, -1), v4)
    return o.permute(0, 2, 1, 3).reshape(B * Nq, H * 64), lse.reshape(B * H, Nq)


def ref_bwd(q, kv, d_o, B, H, Nq, Nk, scale):
    """fp64 autograd of the softmax attention (no dropout) -> dq, dkv"""
    q64 = q.double().requires_grad_(True)
    kv64 = kv.double().requires_grad_(True)
    q4 = q64.view(B, Nq, H, 64).permute(0, 2, 1, 3)
    k4 = kv64[:, :H * 64].reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    v4 = kv64[:, H * 64:].reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    s = torch.einsum('bhid,bhjd->bhij', q4, k4) * scale
    o = torch.einsum('bhij,bhjd->bhid', torch.softmax(s, -1), v4).permute(0, 2, 1, 3).reshape(B * Nq, H * 64)
    (o * d_o.double()).sum().backward()
    return q64.grad, kv64.grad


def check_bwd(which=3):
    dev = 'cuda:0'
    torch.manual_seed(1)
    ok = True
    for (B, H, Nq, Nk) in ((2, 2, 200, 333), (1, 1, 64, 64), (2, 8, 512, 640), (1, 1, 300, 8077), (1, 2, 8077, 130)):
        q = torch.randn(B * Nq, H * 64, device=dev)
        kv = torch.randn(B * Nk, 2 * H * 64, device=dev)
        d_o = torch.randn(B * Nq, H * 64, device=dev) * 3e-4
        d_o[3] *= 40.0
        dq_ref, dkv_ref = ref_bwd(q, kv, d_o, B, H, Nq, Nk, 0.125)
        for mode in ('bf16', 'f16'):
            pl = flash.kv_planes(kv, mode)
            o, lse = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, 0.0, 5, mode=mode, planes=pl)
            for gx in (False, True):
                dq, dkv = flash.flash2_attn_bwd(q, kv, o, d_o, lse, pl, B, H, Nq, Nk, 0.125, 0.0, 5, mode=mode, gx=gx, which=which)
                tol = {'bf16': 3e-2, 'f16': 4e-3}[mode]
                msg = ''
                for nm, a, r in (('dq', dq, dq_ref), ('dkv', dkv, dkv_ref)):
                    if a is None:
                        continue
                    e = (a.double() - r).abs().max().item() / r.abs().max().item()
                    en = ((a.double() - r).norm() / r.norm()).item()
                    good = e < tol and bool(torch.isfinite(a).all())
                    ok &= good
                    msg += ' %s max %.2e rms %.2e %s' % (nm, e, en, 'ok' if good else 'FAIL')
                print('%-18s bwd %-5s gx=%d %s' % ((B, H, Nq, Nk), mode, gx, msg))
        # dropout: against the round-3 kernels (same mask)
        o3, lse3, kvp3 = flash.flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, 0.125, 0.1, 7, x3=True, return_planes=True)
        dq3, dkv3 = flash.flash_attn_bwd_dl(q, kv, o3, d_o, lse3, B, H, Nq, Nk, 0.125, 0.1, 7, x3=True, kv_planes=kvp3)
        pl = flash.kv_planes(kv, 'f16')
        o, lse = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, 0.1, 7, mode='f16', planes=pl)
        for gx in (False, True):
            dq, dkv = flash.flash2_attn_bwd(q, kv, o, d_o, lse, pl, B, H, Nq, Nk, 0.125, 0.1, 7, mode='f16', gx=gx, which=which)
            msg = ''
            for nm, a, r in (('dq', dq, dq3), ('dkv', dkv, dkv3)):
                if a is None:
                    continue
                e = (a - r).abs().max().item() / r.abs().max().item()
                good = e < 8e-3
                ok &= good
                msg += ' %s %.2e %s' % (nm, e, 'ok' if good else 'FAIL')
            print('%-18s bwd dropout f16 gx=%d vs round-3 bf16x3:%s' % ((B, H, Nq, Nk), gx, msg))
    print('CHECK BWD', 'PASSED' if ok else 'FAILED')
    return ok


def bench_bwd(which=3):
    dev = 'cuda:0'
    B = 16
    for name, H, Nq, Nk in (('self', 8, 2048, 2048), ('cross', 1, 2048, 8077), ('decoder', 1, 8077, 2048)):
        q = torch.randn(B * Nq, H * 64, device=dev)
        kv = torch.randn(B * Nk, 2 * H * 64, device=dev)
        d_o = torch.randn(B * Nq, H * 64, device=dev) * 1e-3
        fl = 10.0 * B * H * Nq * Nk * 64
        for p in (0.0, 0.1):
            o3, lse3, kvp3 = flash.flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, 0.125, p, 7, x3=True, return_planes=True)
            t = timeit(lambda: flash.flash_attn_bwd_dl(q, kv, o3, d_o, lse3, B, H, Nq, Nk, 0.125, p, 7, x3=True, kv_planes=kvp3), n=5)
            print('%-8s p=%.1f  round-3 bf16x3 bwd (with its split passes) %.3f ms %7.1f TF/s' % (name, p, t, fl / t * 1e-9))
            for mode in ('bf16', 'f16'):
                pl = flash.kv_planes(kv, mode)
                o, lse = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, p, 7, mode=mode, planes=pl)
                for gx in (False, True):
                    for w in sorted({1, 2, 3} & ({which, 1, 2} if which == 3 else {which})):
                        t = timeit(lambda: flash.flash2_attn_bwd(q, kv, o, d_o, lse, pl, B, H, Nq, Nk, 0.125, p, 7, mode=mode, gx=gx, which=w), n=5)
                        f = fl * (0.4 if w == 1 else 0.6 if w == 2 else 1.0)
                        print('%-8s p=%.1f  flash2 bwd %-5s gx=%d which=%d (with prep) %.3f ms %7.1f TF/s' % (name, p, mode, gx, w, t, f / t * 1e-9))


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def check():
    dev = 'cuda:0'
    torch.manual_seed(0)
    ok = True
    for (B, H, Nq, Nk) in ((2, 2, 200, 333), (1, 1, 64, 64), (2, 8, 512, 640), (1, 1, 300, 8077), (1, 2, 8077, 130)):
        q = torch.randn(B * Nq, H * 64, device=dev)
        kv = torch.randn(B * Nk, 2 * H * 64, device=dev)
        for spike in (0, 1):
            if spike:      # one key far above the first tile's maximum in a late tile (log2 score ~ +35 for query 5, +-4 for the others):
                kv = kv.clone()          # the raise-m path, at operand magnitudes that keep the 16-bit rounding small
                for hh in range(H):
                    kv[Nk - 7, hh * 64:(hh + 1) * 64] = 3.0 * q[5, hh * 64:(hh + 1) * 64]
                    kv[B * Nk - 70 if B * Nk > 140 else 3, hh * 64:(hh + 1) * 64] = 2.0 * q[min(40, B * Nq - 1), hh * 64:(hh + 1) * 64]
            o_ref, lse_ref = ref_attn(q, kv, B, H, Nq, Nk, 0.125)
            for mode in ('bf16', 'f16', 'bf16x3'):
                for waves in (4, 8):
                    o, lse = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, 0.0, 0, mode=mode, waves=waves)
                    eo = (o.double() - o_ref).abs().max().item() / o_ref.abs().max().item()
                    el = (lse.double() - lse_ref).abs().max().item()
                    tol = {'bf16': 2e-2, 'f16': 3e-3, 'bf16x3': 2e-5}[mode] * (8 if spike else 1)
                    good = eo < tol and el < tol and bool(torch.isfinite(o).all())
                    ok &= good
                    print('%-18s spike=%d %-6s w%d  o %.2e  lse %.2e  %s' % ((B, H, Nq, Nk), spike, mode, waves, eo, el, 'ok' if good else 'FAIL'))
        # dropout: same mask as the round-3 kernel
        o3, lse3 = flash.flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, 0.125, 0.1, 7, x3=True)
        for mode in ('bf16', 'f16', 'bf16x3'):
            o, lse = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, 0.1, 7, mode=mode)
            eo = (o - o3).abs().max().item() / o3.abs().max().item()
            tol = {'bf16': 3e-2, 'f16': 6e-3, 'bf16x3': 1e-5}[mode] * 8
            good = eo < tol
            ok &= good
            print('%-18s dropout %-6s vs round-3 bf16x3: o %.2e  lse %.2e %s' % ((B, H, Nq, Nk), mode, eo, (lse - lse3).abs().max().item(), 'ok' if good else 'FAIL'))
    print('CHECK', 'PASSED' if ok else 'FAILED')
    return ok


def bench():
    dev = 'cuda:0'
    B = 16
    shapes = (('self', 8, 2048, 2048), ('cross', 1, 2048, 8077), ('decoder', 1, 8077, 2048))
    if '--self-only' in sys.argv:
        shapes = shapes[:1]
    for name, H, Nq, Nk in shapes:
        q = torch.randn(B * Nq, H * 64, device=dev)
        kv = torch.randn(B * Nk, 2 * H * 64, device=dev)
        fl = 4.0 * B * H * Nq * Nk * 64
        for p in (0.0, 0.1):
            pl3 = flash._planes(kv, 2)
            t = timeit(lambda: flash.call('vxb_flash_attn_fwd_dl', q, pl3, 2, torch.empty_like(q), torch.empty(B * H * Nq, device=dev), B, H, Nq, Nk, 64, 0.125, p, 3))
            print('%-8s p=%.1f  round-3 bf16x3        %.3f ms %7.1f TF/s' % (name, p, t, fl / t * 1e-9))
            pl1 = flash._planes(kv, 1)
            t = timeit(lambda: flash.call('vxb_flash_attn_fwd_dl', q, pl1, 1, torch.empty_like(q), torch.empty(B * H * Nq, device=dev), B, H, Nq, Nk, 64, 0.125, p, 3))
            print('%-8s p=%.1f  round-3 bf16          %.3f ms %7.1f TF/s' % (name, p, t, fl / t * 1e-9))
            for mode in ('bf16', 'f16', 'bf16x3'):
                pl = flash.kv_planes(kv, mode)
                for waves in (4, 8):
                    t = timeit(lambda: flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, p, 3, mode=mode, waves=waves, planes=pl))
                    print('%-8s p=%.1f  flash2 %-6s waves %d  %.3f ms %7.1f TF/s' % (name, p, mode, waves, t, fl / t * 1e-9))


def one(spec):
    """--one mode,waves,p[,shape]: that variant alone, 10 launches (for rocprofv3 --pmc passes)"""
    f = spec.split(',')
    mode, waves, p = f[0], int(f[1]), float(f[2])
    name, H, Nq, Nk = {'self': ('self', 8, 2048, 2048), 'cross': ('cross', 1, 2048, 8077), 'decoder': ('decoder', 1, 8077, 2048)}[f[3] if len(f) > 3 else 'self']
    B = 16
    q = torch.randn(B * Nq, H * 64, device='cuda:0')
    kv = torch.randn(B * Nk, 2 * H * 64, device='cuda:0')
    if mode.startswith('r3'):
        npl = 2 if mode == 'r3x3' else 1
        pl = flash._planes(kv, npl)
        fn = lambda: flash.call('vxb_flash_attn_fwd_dl', q, pl, npl, torch.empty_like(q), torch.empty(B * H * Nq, device='cuda:0'), B, H, Nq, Nk, 64, 0.125, p, 3)
    else:
        pl = flash.kv_planes(kv, mode)
        fn = lambda: flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, p, 3, mode=mode, waves=waves, planes=pl)
    t = timeit(fn)
    print('%s %s: %.3f ms %.1f TF/s' % (name, spec, t, 4.0 * B * H * Nq * Nk * 64 / t * 1e-9))


if __name__ == '__main__':
    if '--bwd' in sys.argv:
        w = int(sys.argv[sys.argv.index('--bwd') + 1])
        good = check_bwd(w)
        if '--quick' not in sys.argv:
            bench_bwd(w)
        sys.exit(0 if good else 1)
    if '--one-bwd' in sys.argv:
        f = sys.argv[sys.argv.index('--one-bwd') + 1].split(',')
        mode, gx, p = f[0], int(f[1]), float(f[2])
        name, H, Nq, Nk = {'self': ('self', 8, 2048, 2048), 'cross': ('cross', 1, 2048, 8077), 'decoder': ('decoder', 1, 8077, 2048)}[f[3] if len(f) > 3 else 'self']
        B = 16
        q = torch.randn(B * Nq, H * 64, device='cuda:0'); kv = torch.randn(B * Nk, 2 * H * 64, device='cuda:0')
        d_o = torch.randn(B * Nq, H * 64, device='cuda:0') * 1e-3
        pl = flash.kv_planes(kv, mode)
        o, lse = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, p, 7, mode=mode, planes=pl)
        t = timeit(lambda: flash.flash2_attn_bwd(q, kv, o, d_o, lse, pl, B, H, Nq, Nk, 0.125, p, 7, mode=mode, gx=bool(gx)))
        print('bwd %s: %.3f ms' % (f, t))
        sys.exit(0)
    if '--one' in sys.argv:
        one(sys.argv[sys.argv.index('--one') + 1])
        sys.exit(0)
    good = True if '--bench-only' in sys.argv else check()
    if '--quick' not in sys.argv:
        bench()
    sys.exit(0 if good else 1)
