"""Tensor-level wrappers of the fused-attention kernels (csrc/flash_attn.hip; 'bf16' and 'bf16x3' precisions)."""
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


def flash_attn_bwd(q, kv, o, d_o, lse, B, H, Nq, Nk, scale, p=0.0, seed=0, x3=False):
    """-> (dq [B*Nq, H*64], dkv [B*Nk, 2*H*64]); same (p, seed) as the forward call."""
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    ws = torch.empty(B * H * Nq, dtype=torch.float32, device=q.device)
    set_meta('attn_core', 10.0 * B * H * Nq * Nk * 64)        # S, dP, dV, dK, dQ (S and dP are computed twice: 14 with recompute)
    call('vxb_flash_attn_bwd_bf16x3' if x3 else 'vxb_flash_attn_bwd_bf16', q, kv, o, d_o, lse, dq, dkv, ws, B, H, Nq, Nk, 64,
         float(scale), float(p), int(seed) & 0xFFFFFFFF)
    return dq, dkv
