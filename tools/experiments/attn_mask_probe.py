import sys, torch, json
sys.path.insert(0, '/root/repo')
from voxactb_amd import flash
dev = 'cuda:0'
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = {}
for (B, H, Nq, Nk) in [(16, 8, 2048, 2048), (16, 1, 2048, 8077), (16, 1, 8077, 2048)]:
    g = torch.Generator(device=dev); g.manual_seed(5)
    q = torch.randn(B * Nq, H * 64, device=dev, generator=g)
    kv = torch.randn(B * Nk, 2 * H * 64, device=dev, generator=g)
    d_o = torch.randn(B * Nq, H * 64, device=dev, generator=g) * 1e-3
    pl = flash.kv_planes(kv, 'f16')
    ffl, bfl = 4.0 * B * H * Nq * Nk * 64, 10.0 * B * H * Nq * Nk * 64
    o, lse, mask = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, 0.1, 3, mode='f16', planes=pl, return_mask=True)
    r = {}
    r['fwd_p0'] = t(lambda: flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, 0.0, 3, mode='f16', planes=pl))
    r['fwd_p0.1_hash'] = t(lambda: flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, 0.1, 3, mode='f16', planes=pl))
    r['fwd_p0.1_store_mask'] = t(lambda: flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, 0.125, 0.1, 3, mode='f16', planes=pl, return_mask=True))
    for which, nm in ((1, 'dq'), (2, 'dkv'), (3, 'both')):
        r['bwd_%s_p0' % nm] = t(lambda: flash.flash2_attn_bwd(q, kv, o, d_o, lse, pl, B, H, Nq, Nk, 0.125, 0.0, 3, mode='f16', gx=False, which=which))
        r['bwd_%s_hash' % nm] = t(lambda: flash.flash2_attn_bwd(q, kv, o, d_o, lse, pl, B, H, Nq, Nk, 0.125, 0.1, 3, mode='f16', gx=False, which=which))
        r['bwd_%s_mask' % nm] = t(lambda: flash.flash2_attn_bwd(q, kv, o, d_o, lse, pl, B, H, Nq, Nk, 0.125, 0.1, 3, mode='f16', gx=False, which=which, drop_mask=mask))
    r['frac_fwd_mask'] = ffl / r['fwd_p0.1_store_mask'] * 1e-9 / 2500
    r['frac_bwd_mask'] = bfl / r['bwd_both_mask'] * 1e-9 / 2500
    r['frac_fwd+bwd_mask'] = (ffl + bfl) / (r['fwd_p0.1_store_mask'] + r['bwd_both_mask']) * 1e-9 / 2500
    r['frac_fwd+bwd_hash'] = (ffl + bfl) / (r['fwd_p0.1_hash'] + r['bwd_both_hash']) * 1e-9 / 2500
    out['B%d_H%d_Nq%d_Nk%d' % (B, H, Nq, Nk)] = {k: round(v, 4) for k, v in r.items()}
print(json.dumps(out, indent=1))
