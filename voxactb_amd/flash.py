"""Tensor-level wrappers of the fused-attention kernels (csrc/flash_attn.hip; 'bf16' and 'bf16x3' precisions)."""
import os

import torch

from ._lib import call, set_meta


def flash_attn_fwd(q, kv, B, H, Nq, Nk, scale, p=0.0, seed=0, x3=False):
    """q [B*Nq, H*64] fp32, kv [B*Nk, 2*H*64] fp32 -> (o [B*Nq, H*64], lse [B*H, Nq])."""
    o = torch.empty_like(q)
    lse = torch.empty((B * H, Nq), dtype=torch.float32, device=q.device)
    set_meta('attn_core', 4.0 * B * H * Nq * Nk * 64)
    call('vxb_flash_attn_fwd_bf16x3' if x3 else 'vxb_flash_attn_fwd_bf16', q, kv, o, lse, B, H, Nq, Nk, 64, float(scale),
         float(p), int(seed) & 0xFFFFFFFF)
    return o, lse


def flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, scale, p=0.0, seed=0, x3=False, return_planes=False):
    """same result as flash_attn_fwd; k | v are first split into bf16 planes (one streaming pass) and then loaded
    global -> LDS directly by the kernel (csrc/flash_fwd_dl.hip)."""
    npl = 2 if x3 else 1
    planes = torch.empty((npl,) + tuple(kv.shape), dtype=torch.bfloat16, device=kv.device)
    set_meta('attn_core', 0.0)
    call('vxb_split_bf16_f32', kv, kv.stride(0), kv.shape[0], kv.shape[1], planes, npl)
    o = torch.empty_like(q)
    lse = torch.empty((B * H, Nq), dtype=torch.float32, device=q.device)
    set_meta('attn_core', 4.0 * B * H * Nq * Nk * 64)
    call('vxb_flash_attn_fwd_dl', q, planes, npl, o, lse, B, H, Nq, Nk, 64, float(scale), float(p), int(seed) & 0xFFFFFFFF)
    if return_planes:
        return o, lse, planes
    return o, lse


def _planes(x, npl):
    out = torch.empty((npl,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    set_meta('attn_core', 0.0)
    call('vxb_split_bf16_f32', x, x.stride(0), x.shape[0], x.shape[1], out, npl)
    return out


def flash_attn_bwd_dl(q, kv, o, d_o, lse, B, H, Nq, Nk, scale, p=0.0, seed=0, x3=False, kv_planes=None):
    """flash_attn_bwd with q, dO and k | v tiles loaded global -> LDS directly from bf16 planes (csrc/flash_bwd_dl.hip);
    kv_planes: the planes made by flash_attn_fwd_dl(..., return_planes=True), recomputed when None."""
    npl = 2 if x3 else 1
    if kv_planes is None:
        kv_planes = _planes(kv, npl)
    qp, dop = _planes(q, npl), _planes(d_o, npl)
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    ws = torch.empty(B * H * Nq, dtype=torch.float32, device=q.device)
    set_meta('attn_core', 10.0 * B * H * Nq * Nk * 64)
    call('vxb_flash_attn_bwd_dl', q, kv, o, d_o, lse, kv_planes, qp, dop, npl, dq, dkv, ws, B, H, Nq, Nk, 64, float(scale),
         float(p), int(seed) & 0xFFFFFFFF)
    return dq, dkv


def flash_attn_bwd(q, kv, o, d_o, lse, B, H, Nq, Nk, scale, p=0.0, seed=0, x3=False):
    """-> (dq [B*Nq, H*64], dkv [B*Nk, 2*H*64]); same (p, seed) as the forward call."""
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    ws = torch.empty(B * H * Nq, dtype=torch.float32, device=q.device)
    set_meta('attn_core', 10.0 * B * H * Nq * Nk * 64)        # S, dP, dV, dK, dQ (S and dP are computed twice: 14 with recompute)
    call('vxb_flash_attn_bwd_bf16x3' if x3 else 'vxb_flash_attn_bwd_bf16', q, kv, o, d_o, lse, dq, dkv, ws, B, H, Nq, Nk, 64,
         float(scale), float(p), int(seed) & 0xFFFFFFFF)
    return dq, dkv


MODES = {'bf16': 0, 'f16': 1, 'bf16x3': 2}


def kv_planes(kv, mode):
    """k | v operand plane of the round-4 kernels: bf16 ('bf16') or fp16 ('f16'), [1][B*Nk][2*H*64]."""
    npl = 2 if mode == 'bf16x3' else 1          # 'bf16x3' (forward only): the hi | lo bf16 planes of vxb_split_bf16_f32
    out = torch.empty((npl,) + tuple(kv.shape), dtype=torch.float16 if mode == 'f16' else torch.bfloat16, device=kv.device)
    set_meta('attn_core', 0.0)
    call('vxb_split_f16_f32' if mode == 'f16' else 'vxb_split_bf16_f32', kv, kv.stride(0), kv.shape[0], kv.shape[1], out, npl)
    return out


# Round 6: with dropout the forward stores the mask as "keep words" and the backward reads them instead of hashing (seed, row, key) again;
# the storing forward draws the mask from a per-row linear congruential sequence (one instruction per score pair instead of a 7-instruction
# hash).  Measured at B = 16 (tools/experiments/attn_mask_probe.py, profiles/r06_attn_mask_probe.log), hash pair -> stored mask:
#   8-head self-attention 2048 x 2048:  forward 0.277 -> 0.232 ms, backward (dQ + dK | dV + preparation) 0.727 -> 0.668
#   decoder cross attention 8077 x 2048 (1 head, 1024 workgroups): forward 0.110 -> 0.103, backward 0.405 -> 0.365
#   latent cross attention 2048 x 8077 (1 head, 256 workgroups): forward 0.129 -> 0.131 (every wave's scalar stores sit on its critical
#   path below one workgroup per CU), backward 0.363 -> 0.333
# Stored from 256 workgroups on (the step's three attention shapes); smaller problems keep the hash pair.  '0' = never, '2' = always (tests).
DROP_MASK = os.environ.get('VOXACTB_ATTN_DROP_MASK', '1')
DROP_MASK_MIN_WORKGROUPS = 256


def drop_mask_words(B, H, Nq, Nk, device):
    """buffer of the dropout keep words of one attention call (include/voxactb_hip.h: vxb_flash2_attn_fwd_mask)"""
    from ._lib import lib
    return torch.empty((lib().vxb_flash2_drop_mask_bytes(B, H, Nq, Nk) + 3) // 4, dtype=torch.int32, device=device)


def flash2_attn_fwd(q, kv, B, H, Nq, Nk, scale, p=0.0, seed=0, mode='f16', waves=0, planes=None, return_planes=False, return_mask=False,
                    store_mask=True):
    """Pipelined forward (csrc/flash2_fwd.hip).  Same contract as flash_attn_fwd_dl.  return_mask (single-product modes, p > 0): the forward
    also stores the dropout keep words; appended to the result (None when there is no dropout) for flash2_attn_bwd(drop_mask=...)."""
    if planes is None:
        planes = kv_planes(kv, mode)
    o = torch.empty_like(q)
    lse = torch.empty((B * H, Nq), dtype=torch.float32, device=q.device)
    mask = None
    set_meta('attn_core', 4.0 * B * H * Nq * Nk * 64)
    big = DROP_MASK == '2' or B * H * ((Nq + 127) // 128) >= DROP_MASK_MIN_WORKGROUPS
    if return_mask and store_mask and DROP_MASK != '0' and big and mode != 'bf16x3' and int(float(p) * 65536.0) > 0:
        mask = drop_mask_words(B, H, Nq, Nk, q.device)
        call('vxb_flash2_attn_fwd_mask', q, planes, MODES[mode], o, lse, mask, B, H, Nq, Nk, 64, float(scale), float(p), int(seed) & 0xFFFFFFFF,
             int(waves))
    else:
        call('vxb_flash2_attn_fwd', q, planes, MODES[mode], o, lse, B, H, Nq, Nk, 64, float(scale), float(p), int(seed) & 0xFFFFFFFF,
             int(waves))
    out = (o, lse) + ((planes,) if return_planes else ()) + ((mask,) if return_mask else ())
    return out


def flash2_attn_bwd(q, kv, o, d_o, lse, planes, B, H, Nq, Nk, scale, p=0.0, seed=0, mode='f16', gx=True, which=3, drop_mask=None):
    """Pipelined backward (csrc/flash2_bwd.hip) -> (dq, dkv); planes: the forward's k | v plane (kv_planes(kv, mode)); drop_mask: the keep
    words the forward call wrote (flash2_attn_fwd(return_mask=True)) -- None: the mask is regenerated from (seed, row, key)."""
    from ._lib import lib
    dq = torch.empty_like(q) if which & 1 else None
    dkv = torch.empty_like(kv) if which & 2 else None
    nws = lib().vxb_flash2_attn_bwd_ws_bytes(B, H, Nq, int(gx))
    ws = torch.empty(nws, dtype=torch.uint8, device=q.device)
    set_meta('attn_core', 10.0 * B * H * Nq * Nk * 64 * (0.4 if which == 1 else 0.6 if which == 2 else 1.0))
    if drop_mask is not None:
        call('vxb_flash2_attn_bwd_mask', q, kv, o, d_o, lse, planes, drop_mask, MODES[mode], int(gx), dq, dkv, ws, B, H, Nq, Nk, 64, float(scale),
             float(p), int(seed) & 0xFFFFFFFF, int(which))
    else:
        call('vxb_flash2_attn_bwd', q, kv, o, d_o, lse, planes, MODES[mode], int(gx), dq, dkv, ws, B, H, Nq, Nk, 64, float(scale), float(p),
             int(seed) & 0xFFFFFFFF, int(which))
    return dq, dkv
