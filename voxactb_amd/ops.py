"""Thin tensor-level wrappers over the C ABI (include/voxactb_hip.h).  Every function launches hand-written
gfx950 kernels on the current torch stream; torch is used for allocation and views only.
"""
import numpy as np
import os

import torch

from . import _lib
from ._lib import call

LRELU_SLOPE = 0.02   # reference: peract/helpers/network_utils.py:12
ACT_NONE, ACT_LRELU = 0, 1
_NAIVE_MACS = 1 << 22

# Compute precision of the matrix-core kernels, set by PerceiverEngine per call (its default is bf16x3):
# "fp32" = exact fp32 MFMA everywhere; "bf16x3" = hi/lo split products on the bf16 matrix cores (same parity bounds);
# "bf16" = plain bf16 operands (throughput mode).  Accumulators and every streaming kernel are fp32 in all modes.
PRECISION = 'fp32'
HALO_CONV = True     # 3x3x3 stride-1 bf16 convs go through the LDS-halo kernel (conv_halo_bf16.hip)
HALO_D2S = False     # ... also the depth-to-space forward of the polyphase up-conv: measured 21.0 ms vs 19.3 ms generic at
                     # 20^3 x 8000 columns (the 4x8x8 tile wastes 44 % on a 20^3 grid and the halo is re-staged per column block)
# Weight gradients of the 3x3x3 LDS-halo convs (`final`, the polyphase up-conv) in the 'bf16x3' precision: 'fp16' = ONE fp16
# product per term with the gradient operand pre-scaled by a power of two from its largest magnitude (absmax_scale); '' = the
# bf16x3 triple.  Set by PerceiverEngine.backward from its `wgrad_precision`.  A weight gradient is a leaf of the backward pass:
# measured against the reference's gradients at configs[1] / [2] size the two are indistinguishable.
WGRAD_PRECISION = ''
# ... and the weight gradients that go through the generic transposed-read kernel (wgrad_bf16.hip): the linear layers with >= 1024
# rows and the 5^3 convs.  Their gradient operand is scaled by the power of two computed from the PREVIOUS call at the same site
# (delayed scaling: the kernel reports the maximum it saw, for free; the first call takes an explicit absmax pass), with 5 bits
# of headroom (a 32-fold jump from one step to the next still fits) and saturation on top -- see DESIGN.md 4a.
GENERIC_WGRAD_F16 = False
_GRAD_SCALE = {}          # call site (weight address / name, use within the step) -> [current scale, next scale] device tensors, used in turn
_GRAD_USE = {}            # call site -> how often this backward pass has used it so far (PerceiverEngine.backward resets it): with
                          # transformer_iterations > 1 or weight_tie_layers one weight is hit several times per step, by gradients of
                          # different magnitude -- every use keeps its own delayed scale (round-4 advisor)
_WCACHE = {}


_FCACHE = {}
_F16CACHE = {}       # transposed weight planes (address, shape) -> the same matrix as ONE fp16 plane in fragment order (fp16x2 data gradients)
_LAST_LIN_DY_SCALE = [None]   # operand scale of dY the last linear_bwd's weight-gradient launch took (None: it took none)
_LAST_GRAD_SCALE = [None]     # operand scale (device {2^k, 2^-k}) the last delayed-scaling weight-gradient launch applied to its gradient
# rows from which the 128 x 512 workgroup tiles of the wide kernels fill the chip (M / 128 workgroups per 512 columns: 128 at M = 16384).
# Below it -- the released VoxAct-B recipe trains with replay.batch_size = 1, M = 2048 -- the 128 x 64 / 128 x 128 tile kernels run
# (16 workgroups of 92 us each per linear layer otherwise: profiles/r04_v50_*).
WIDE_MIN_M = int(os.environ.get('VOXACTB_WIDE_MIN_M', 16384))


def begin_backward():
    """a new backward pass starts: every delayed-scaling call site counts its uses from zero again (_GRAD_USE)"""
    _GRAD_USE.clear()


def set_wide_min_rows(n):
    """tests / experiments: dispatch the wide kernels from `n` rows on (Python side and the weight-gradient entry's own test)"""
    global WIDE_MIN_M
    WIDE_MIN_M = int(n)
    _lib.lib().vxb_debug_set_wide_min_rows(int(n))


WIDE_GEMM = os.environ.get('VOXACTB_WIDE_GEMM', '1') != '0'     # N = 512 linear layers on the wide kernel (gemm_wide.hip); '0': register-staged 128^2 kernel
# the data gradients of the big linear layers on two fp16 products (vxb_gemm_wide_f16x2_f32; DGRAD_PRECISION below selects the arithmetic)
LIN_DGRAD_X2 = os.environ.get('VOXACTB_DGRAD_PRECISION', 'fp16x2') == 'fp16x2' and os.environ.get('VOXACTB_LIN_DGRAD_X2', '1') != '0'
GEMM_BD = True       # direct-to-LDS GEMMs read their weight fragments straight from global memory (no B tile in LDS)


def gemm_wfrag(wb):
    """bf16 weights [N][K] or planes [2][N][K] -> MFMA fragment order [N/32][K/16][planes][half 2][col 32][8] (cached per
    weight tensor until new_step()): the B operand of the direct-to-LDS kernels without a trip through LDS."""
    if not GEMM_BD or wb.shape[-1] % 16:
        return None
    key = (wb.data_ptr(), tuple(wb.shape))
    f = _FCACHE.get(key)
    if f is None:
        w3 = wb if wb.dim() == 3 else wb.unsqueeze(0)
        P, N, K = w3.shape
        if N % 128:                      # the kernel's column tiles are 128 wide: zero rows up to the next multiple
            w3 = torch.cat((w3, w3.new_zeros((P, 128 - N % 128, K))), dim=1)
            N = w3.shape[1]
        f = w3.view(P, N // 32, 32, K // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
        _FCACHE[key] = (f, wb)           # (keeps wb alive: the key is its address)
        return f
    return f[0]


_IDX = {}          # cached index tables of gather_cvt (layout only: they do not depend on the weights' values)


def traced_index(key, shape, layout_fn, dev):
    """int32 table T with  layout_fn(x).reshape(-1) == x.reshape(-1)[T]  for any tensor x of `shape` and any chain of pure layout ops
    (view / permute / flip / index_select / contiguous / cat with -1 padding): the chain is run ONCE on arange(numel) and cached."""
    t = _IDX.get(key)
    if t is None:
        n = 1
        for d in shape:
            n *= int(d)
        assert n < (1 << 30)
        t = layout_fn(torch.arange(n, dtype=torch.int32, device=dev).view(shape)).reshape(-1).contiguous()
        _IDX[key] = t
    return t


def gather_cvt(src, idx, mode, plane_off=0):
    """src fp32 (contiguous, any shape) -> flat fp16 (mode 0) / bf16 hi | lo by plane offset (mode 1) tensor dst[i] = cvt(src[idx[i]])
    in one pass (gemm_dl.hip: vxb_gather_cvt_f32); idx from traced_index."""
    assert src.is_contiguous() and src.dtype == torch.float32
    dst = torch.empty(idx.numel(), dtype=torch.float16 if mode == 0 else torch.bfloat16, device=src.device)
    call('vxb_gather_cvt_f32', src, idx, idx.numel(), dst, int(mode), int(plane_off))
    return dst


def _wfrag_index(I2, plane_off):
    """gemm_wfrag's layout on an index matrix [N][K]: planes hi (index j) and lo (index plane_off + j), -1 in the zero rows that pad N
    to a multiple of 128."""
    N, K = I2.shape
    w3 = torch.stack((I2, I2 + plane_off))
    if N % 128:
        w3 = torch.cat((w3, w3.new_full((2, 128 - N % 128, K), -1)), dim=1)
        N = w3.shape[1]
    return w3.view(2, N // 32, 32, K // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()


WEIGHT_GATHER = os.environ.get('VOXACTB_WEIGHT_GATHER', '1') != '0'     # polyphase weights into their fragment orders by one gather pass each ('0': the ATen chains)


def gemm_wfrag_geglu(wb):
    """fragment order of the GEGLU up-projection's planes [2][2 F][K] with the rows of every 64-row block interleaved as [32 value rows
    | their 32 gate rows] (include/voxactb_hip.h: vxb_gemm_wide_geglu_fwd_f32); made by prepare_linear_weights(..., geglu=...) in the
    step, by a gather + shuffle here otherwise."""
    key = (wb.data_ptr(), tuple(wb.shape), 'glu')
    f = _FCACHE.get(key)
    if f is not None:
        return f[0]
    P, N, K = wb.shape
    F = N // 2
    q = torch.arange(F, device=wb.device).view(F // 32, 32)
    idx = torch.cat((q, q + F), dim=1).reshape(-1)                      # position -> source row
    w3 = wb.index_select(1, idx)
    f = w3.view(P, N // 32, 32, K // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
    _FCACHE[key] = (f, wb)
    return f


CACHE_OWNER = None       # the engine whose prepared weights fill the caches below (PerceiverEngine.forward: inference with frozen weights reuses them)


def new_step():
    """weights change every optimizer step: drop the per-step bf16 weight copies."""
    global CACHE_OWNER
    CACHE_OWNER = None
    _WCACHE.clear()
    _FCACHE.clear()
    _F16CACHE.clear()


_WPREP = {}
BATCH_WEIGHT_SPLIT = os.environ.get('VOXACTB_BATCH_WSPLIT', '1') != '0'      # '0': per-weight splits (A/B runs)


def prepare_linear_weights(weights, geglu=(), f16_dgrad=False):
    """bf16 planes of every linear-layer weight of the step, plain AND transposed (the B operands of the forward / data-gradient
    GEMMs), made by ONE launch (vxb_split_bf16_batch_f32) instead of a split + a transposing copy per weight and use; fills the
    cache _bf16_weight() reads.  `weights`: 2-D fp32 tensors (views into the flat parameter arena, so the descriptor table and
    the output buffer are built once and reused every step).  f16_dgrad: the backward pass of this step will run the data gradients
    of the wide linear layers on two fp16 products (fp16x2): also emit every such transposed matrix as one fp16 plane in fragment order."""
    if not (BATCH_WEIGHT_SPLIT and _mm()):
        return
    ws = [w for w in weights if w.dim() == 2 and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
          and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0 and w.data_ptr() % 16 == 0]
    if not ws:
        return
    npl = 2 if PRECISION == 'bf16x3' else 1
    glu = set(w.data_ptr() for w in geglu)      # GEGLU up-projections: their forward fragments get the value / gate interleave
    f16_dgrad = bool(f16_dgrad and LIN_DGRAD_X2 and DGRAD_PRECISION == 'fp16x2')
    key = (tuple(w.data_ptr() for w in ws), tuple(tuple(w.shape) for w in ws), npl, tuple(sorted(glu)), f16_dgrad)
    ent = _WPREP.get(key)
    if ent is None:
        if len(_WPREP) > 8:
            _WPREP.clear()
        def wants_frag(n_out, k_in):        # the shapes gemm_bf16w() sends to the wide kernel: also emit the MFMA fragment order
            return WIDE_GEMM and GEMM_BD and npl == 2 and n_out % 512 == 0 and k_in % 32 == 0 and k_in >= 256
        def wants_f16(n_out, k_in):         # ... and, for the transposed matrix (the data gradient's weight operand), one fp16 plane
            return wants_frag(n_out, k_in) and f16_dgrad
        total = sum(npl * w.numel() * (2 + int(wants_frag(*w.shape)) + int(wants_frag(w.shape[1], w.shape[0])))
                    + w.numel() * int(wants_f16(w.shape[1], w.shape[0])) for w in ws)
        buf = torch.empty(total, dtype=torch.bfloat16, device=ws[0].device)
        rows, views, frags, f16s, off, tile0 = [], [], [], [], 0, 0
        for w in ws:
            N, K = w.shape
            for tr in (0, 1):
                rows.append([w.data_ptr(), buf.data_ptr() + 2 * off, N, K, tr, tile0])
                shape = ((K, N) if tr else (N, K))
                v = buf[off:off + npl * N * K].view((2,) + shape if npl == 2 else shape)
                views.append(((w.data_ptr(), (N, K), bool(tr)), v))
                off += npl * N * K
                tile0 += ((N + 63) // 64) * ((K + 63) // 64)
                if wants_frag(*shape):
                    il = 4 if (not tr and w.data_ptr() in glu and N % 512 == 0) else 0
                    rows.append([w.data_ptr(), buf.data_ptr() + 2 * off, N, K, tr | 2 | il, tile0])
                    n_o, k_o = shape
                    f = buf[off:off + npl * N * K].view(n_o // 32, k_o // 16, npl, 2, 32, 8)
                    frags.append((v, f, bool(il)))
                    off += npl * N * K
                    tile0 += ((N + 63) // 64) * ((K + 63) // 64)
                if tr and wants_f16(*shape):
                    rows.append([w.data_ptr(), buf.data_ptr() + 2 * off, N, K, tr | 2 | 8, tile0])
                    n_o, k_o = shape
                    f16s.append((v, buf[off:off + N * K].view(n_o // 32, k_o // 16, 2, 32, 8)))
                    off += N * K
                    tile0 += ((N + 63) // 64) * ((K + 63) // 64)
        desc = torch.tensor(rows, dtype=torch.int64).to(ws[0].device)
        ent = _WPREP[key] = (desc, len(rows), tile0, buf, views, ws, frags, f16s)       # (ws keeps the sources alive: keyed by address)
    desc, n, tiles, buf, views, _, frags, f16s = ent
    call('vxb_split_bf16_batch_f32', desc, n, tiles, npl)
    for (ptr, shape, tr), v in views:
        _WCACHE[(ptr, shape, tr, PRECISION)] = v
    for v, f, il in frags:                  # gemm_wfrag(v) / gemm_wfrag_geglu(v) find the fragment-order copy made by the same launch
        _FCACHE[(v.data_ptr(), tuple(v.shape)) + (('glu',) if il else ())] = (f, v)
    for v, f in f16s:
        _F16CACHE[(v.data_ptr(), tuple(v.shape))] = (f, v)


def split_bf16(w, x3=None):
    """fp32 [N][K] -> bf16 [N][K] ('bf16') or the hi/lo planes [2][N][K] of the 'bf16x3' split (lo = bf16(w - hi))."""
    if x3 is None:
        x3 = PRECISION == 'bf16x3'
    if (w.is_cuda and w.dim() == 2 and w.dtype == torch.float32 and w.shape[1] % 4 == 0 and w.stride(1) == 1
            and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0):
        planes = split_planes(w, 2 if x3 else 1)          # one pass (vxb_split_bf16_f32) instead of five torch kernels
        return planes if x3 else planes[0]
    hi = w.to(torch.bfloat16)
    if not x3:
        return hi
    return torch.stack((hi, (w - hi.float()).to(torch.bfloat16)))


def _mm():
    """matrix-core mode of the large GEMMs / convs: False = fp32 MFMA, True = bf16 MFMA (plain or x3 split)."""
    return PRECISION in ('bf16', 'bf16x3')


def _bf16_weight(W, transposed):
    key = (W.data_ptr(), tuple(W.shape), transposed, PRECISION)
    wb = _WCACHE.get(key)
    if wb is None:
        wb = split_bf16(W.t().contiguous() if transposed else W.contiguous())
        _WCACHE[key] = wb
    return wb


def _f32c(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise _lib.VoxactbHipError('tensor on %s: the kernels need a HIP device (no CPU fallback)' % t.device)
            assert t.dtype == torch.float32, t.dtype


def gemm(A, B, C, M, N, K, sAm, sAk, sBk, sBn, ldc, bias=None, residual=None, batch=1, H=1,
         bA=(0, 0), bB=(0, 0), bC=(0, 0), alpha=1.0, act=ACT_NONE, accumulate=False, label=None):
    _f32c(A, B, C, bias, residual)
    _lib.set_meta(label or 'gemm', 2.0 * M * N * K * batch)
    call('vxb_gemm_f32', A, B, C, bias, residual, M, N, K, sAm, sAk, sBk, sBn, ldc, batch, H,
         bA[0], bA[1], bB[0], bB[1], bC[0], bC[1], float(alpha), act, LRELU_SLOPE, int(accumulate))


def naive_gemm(A, B, C, M, N, K, sAm, sAk, sBk, sBn, ldc, bias=None, act=ACT_NONE, accumulate=False):
    _f32c(A, B, C, bias)
    call('vxb_naive_gemm_f32', A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, act, LRELU_SLOPE, int(accumulate))


def _small(M, N, K):
    return M * N * K <= _NAIVE_MACS or (K % 4) or (N % 4)


def linear(x, W, bias=None, act=ACT_NONE, residual=None, out=None, f16_out=None):
    """out[M,N] = act(x[M,K] @ W[N,K]^T + bias) (+ residual).  nn.Linear / DenseBlock forward.  f16_out: see gemm_bf16w -- then the
    return value is (out, filled)."""
    M, K = x.shape
    N = W.shape[0]
    if f16_out is not None:
        if not _small(M, N, K) and _mm() and K % 8 == 0 and W.is_contiguous():
            return gemm_bf16w(x, _bf16_weight(W, False), out, bias, act, residual, label='gemm_fwd %dx%dx%d' % (M, N, K), f16_out=f16_out)
        return linear(x, W, bias, act, residual, out), False
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if _small(M, N, K):
        naive_gemm(x, W, out, M, N, K, x.stride(0), 1, 1, W.stride(0), out.stride(0), bias, act)
        if residual is not None:
            axpy_(out, residual)
    elif _mm() and K % 8 == 0 and W.is_contiguous():
        gemm_bf16w(x, _bf16_weight(W, False), out, bias, act, residual, label='gemm_fwd %dx%dx%d' % (M, N, K))
    else:
        gemm(x, W, out, M, N, K, x.stride(0), 1, 1, W.stride(0), out.stride(0), bias=bias, residual=residual, act=act,
             label='gemm_fwd %dx%dx%d' % (M, N, K))
    return out


def linear_dgrad(dy, W, dx, dx_accumulate=False, sc_dy=None):
    """dx[M,K] (+)= dy[M,N] @ W[N,K] for the non-small shapes of linear_bwd; sc_dy: the operand scale of dy its weight-gradient launch
    reported (None: the bf16x3 / fp32 arithmetic)."""
    M, N = dy.shape
    K = W.shape[1]
    if _mm() and N % 8 == 0 and W.is_contiguous():
        Wt = _bf16_weight(W, True)
        f16 = _F16CACHE.get((Wt.data_ptr(), tuple(Wt.shape))) if (LIN_DGRAD_X2 and DGRAD_PRECISION == 'fp16x2' and sc_dy is not None) else None
        if (f16 is not None and K % 512 == 0 and N % 32 == 0 and N >= 256 and M >= WIDE_MIN_M and dy.stride(1) == 1 and dy.stride(0) % 4 == 0
                and dy.data_ptr() % 16 == 0):
            # dX = dY @ W on two fp16 products: dY * 2^k as an fp16 hi + lo pair, W as one fp16 value (gemm_wide.hip, X2)
            _lib.set_meta('gemm_dgrad %dx%dx%d' % (M, K, N), 2.0 * M * N * K)
            call('vxb_gemm_wide_f16x2_f32', dy, dy.stride(0), f16[0], dx, dx.stride(0), None, M, K, N, int(dx_accumulate), sc_dy)
        else:
            gemm_bf16w(dy, Wt, dx, accumulate=dx_accumulate, label='gemm_dgrad %dx%dx%d' % (M, K, N))
    else:
        gemm(dy, W, dx, M, K, N, dy.stride(0), 1, W.stride(0), 1, dx.stride(0), accumulate=dx_accumulate,
             label='gemm_dgrad %dx%dx%d' % (M, K, N))


def linear_bwd(x, W, dy, dW, db=None, dx=None, dx_accumulate=False, ws=None):
    """dW[N,K] += dy^T x ; db[N] += colsum(dy) ; dx[M,K] (+)= dy @ W   (dy already includes the activation')."""
    M, K = x.shape
    N = W.shape[0]
    if _small(M, N, K) or (dy.stride(0) % 4):
        naive_gemm(dy, x, dW, N, K, M, 1, dy.stride(0), x.stride(0), 1, dW.stride(0), accumulate=True)
        if dx is not None:
            naive_gemm(dy, W, dx, M, K, N, dy.stride(0), 1, W.stride(0), 1, dx.stride(0), accumulate=dx_accumulate)
    else:
        # dW = dy^T x reduces over M rows: few output tiles, so split the reduction across workgroups (batch = split)
        tiles = ((N + 127) // 128) * ((K + 127) // 128 if K > 64 else 1)
        ns = 1
        sc_dy = None                 # the operand scale of dy, when the weight-gradient launch below takes (and reports) one
        if tiles < 512 and dW.is_contiguous():
            cands = [d for d in range(1, 65) if M % d == 0 and (M // d) >= 256]
            ok = [d for d in cands if tiles * d >= 512]
            ns = min(ok) if ok else (max(cands) if cands else 1)
        if _mm() and dW.is_contiguous() and dy.stride(0) == N:
            # both operands are row(position)-major -> the transposed-read bf16 kernel (a 1x1x1 "conv" over M positions)
            nsb = max(1, min(64, (512 + tiles - 1) // tiles, M // 256))
            wt = wide_wgrad_tiles(N, K)
            if wt and WGRAD_PRECISION == 'fp16' and GENERIC_WGRAD_F16 and M >= WIDE_MIN_M:
                # the wide kernel's workgroups own 128 x 512 tiles, one per CU: slices so that tiles x slices fills the 256 CUs once
                nsb = max(1, min(64, int(os.environ.get('VOXACTB_WIDE_WGS', 256)) // wt, M // 1024))
            _LAST_GRAD_SCALE[0] = None
            res = conv3d_wgrad(dy, x, K, M, 1, 1, 1, 0, ldy=x.stride(0), nsplit=nsb, label='gemm_wgrad %dx%dx%d' % (N, K, M),
                               possum_into=db, grad_key=('lin', W.data_ptr()) if M >= 1024 else None, grad_is_src0=True,
                               add_into=dW)
            if res is not None:
                axpy_(dW, res)
            db = None                    # (the bias gradient came out of the same launch)
            sc_dy, _LAST_GRAD_SCALE[0] = _LAST_GRAD_SCALE[0], None
            _LAST_LIN_DY_SCALE[0] = sc_dy     # (for a caller that runs the data gradient itself: linear_dgrad_geglu_bwd)
        elif ns > 1:
            rc = M // ns
            part = torch.empty((ns, N, K), dtype=torch.float32, device=x.device)
            gemm(dy, x, part, N, K, rc, 1, dy.stride(0), x.stride(0), 1, K, batch=ns, H=1, bA=(rc * dy.stride(0), 0),
                 bB=(rc * x.stride(0), 0), bC=(N * K, 0), label='gemm_wgrad %dx%dx%d' % (N, K, M))
            sum_splits(part, ns, N * K, dW, accumulate=True)
        else:
            gemm(dy, x, dW, N, K, M, 1, dy.stride(0), x.stride(0), 1, dW.stride(0), accumulate=True, label='gemm_wgrad %dx%dx%d' % (N, K, M))
        if dx is not None:
            linear_dgrad(dy, W, dx, dx_accumulate, sc_dy)
    if db is not None:
        colsum(dy, db, accumulate=True)


def wide_wgrad_tiles(R, N):
    """row tiles of the wide fp16 weight-gradient kernel (wgrad_bf16.hip: one operand with exactly 512 channels, the other a
    multiple of 128) for part [R][N], 0 where it does not apply -- mirrors the test in wgrad_bf16_impl."""
    if N == 512 and R >= 128 and R % 128 == 0:
        return R // 128
    if R == 512 and N >= 128 and N % 128 == 0:
        return N // 128
    return 0


def colsum(x, out, accumulate=False):
    rows, N = x.shape
    rpb = max(64, rows // 1024)
    ws = torch.empty(((rows + rpb - 1) // rpb) * N, dtype=torch.float32, device=x.device)
    call('vxb_colsum_f32', x, rows, N, x.stride(0), ws, out, int(accumulate))


def sum_splits(part, nsplit, n, dst, accumulate=False, alpha=1.0):
    if isinstance(alpha, torch.Tensor):         # a factor that lives on the device (e.g. 1 / the fp16 operand scale): no host sync
        call('vxb_sum_splits_dev_f32', part, nsplit, n, dst, int(accumulate), alpha)
        return
    call('vxb_sum_splits_f32', part, nsplit, n, dst, int(accumulate), float(alpha))


def absmax_scale(x):
    """-> device tensor [scale, 1 / scale]: the power of two that maps max |x| into [2^14, 2^15) (vxb_absmax_scale_f32)."""
    assert x.is_contiguous() and x.dtype == torch.float32
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    sc = torch.empty(2, dtype=torch.float32, device=x.device)
    call('vxb_absmax_scale_f32', x, x.numel(), ws, sc)
    return sc


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    rows, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call('vxb_layernorm_fwd_f32', x, gamma, beta, y, mean, rstd, rows, D, float(eps))
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dx=None, accumulate_dx=False):
    rows, D = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    nblk = (rows + 31) // 32
    ws = torch.empty(nblk * 2 * D + 2 * D, dtype=torch.float32, device=x.device)
    call('vxb_layernorm_bwd_f32', dy, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, rows, D, int(accumulate_dx))
    return dx


def softmax_rows(S, rows, cols, ld, p=0.0, seed=0):
    """in place S -> P; returns P_drop (== S itself when p == 0)."""
    if p == 0 and cols >= 65536 and rows <= 1024:
        # a handful of very long rows (act(): B x V^3): many workgroups per row instead of one
        ws = torch.empty(rows * ((cols + 8191) // 8192) * 2, dtype=torch.float32, device=S.device)
        call('vxb_softmax_long_rows_f32', S, ws, rows, cols, ld)
        return S
    Pd = torch.empty_like(S) if p > 0 else None
    call('vxb_softmax_rows_f32', S, Pd, rows, cols, ld, float(p), int(seed) & 0xFFFFFFFF)
    return Pd if p > 0 else S


def softmax_bwd_rows(P, dP, rows, cols, ld, scale, p=0.0, seed=0):
    call('vxb_softmax_bwd_rows_f32', P, dP, rows, cols, ld, float(scale), float(p), int(seed) & 0xFFFFFFFF)
    return dP


FUSE_GEGLU = os.environ.get('VOXACTB_FUSE_GEGLU', '1') != '0'     # GEGLU inside the wide GEMM's epilogues (gemm_wide.hip); '0': separate passes


# ... and its backward inside the down-projection's data gradient: OFF.  Measured in the step (same box): the fused launch takes 0.63 ms where
# the wide GEMM (0.235 ms) + vxb_geglu_bwd_f32 (0.264 ms, a streaming pass at 5 TB/s) take 0.50 -- with one workgroup per CU the 1 MB of
# epilogue traffic and the erf / exp arithmetic of a 128 x 512 tile are serial with its main loop instead of running at full HBM bandwidth.
# Round 6: with the row-contiguous epilogue and on the two-fp16-product arithmetic of the layer's plain data gradient the fused launch wins
# (vxb_gemm_wide_geglu_bwd_f16x2_f32; the bf16x3 one -- '1' -- still only ties: a third more MFMAs than the separate fp16x2 launch).
# 'x2' (default): fused where the fp16x2 data gradient applies; '1': fused on bf16x3 elsewhere too; '0': never.
_fgb = os.environ.get('VOXACTB_FUSE_GEGLU_BWD', 'x2')
FUSE_GEGLU_BWD = False if _fgb == '0' else ('x2' if _fgb == 'x2' else True)


def _geglu_wide_ok(x, rows_out, K):
    return (FUSE_GEGLU and WIDE_GEMM and GEMM_BD and PRECISION == 'bf16x3' and rows_out % 512 == 0 and K % 32 == 0 and K >= 256
            and x.shape[0] >= WIDE_MIN_M and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def linear_geglu(x, W, bias):
    """h = x @ W^T + bias [M][2 F] and gg = h[:, :F] * gelu(h[:, F:]) [M][F] (FeedForward's up-projection + GEGLU, perceiver_lang_io.py:
    74-78, :100-106): one launch on the wide kernel where it applies (same bits as linear() + geglu_fwd())."""
    M, K = x.shape
    N = W.shape[0]
    if _geglu_wide_ok(x, N, K) and W.is_contiguous() and bias is not None:
        wf = gemm_wfrag_geglu(_bf16_weight(W, False))
        h = torch.empty((M, N), dtype=torch.float32, device=x.device)
        gg = torch.empty((M, N // 2), dtype=torch.float32, device=x.device)
        _lib.set_meta('gemm_fwd %dx%dx%d' % (M, N, K), 2.0 * M * N * K)
        call('vxb_gemm_wide_geglu_fwd_f32', x, x.stride(0), wf, bias, h, gg, M, N // 2, K)
        return h, gg
    h = linear(x, W, bias)
    return h, geglu_fwd(h)


def geglu_bwd_fusable(dy, W2, h):
    """Will linear_dgrad_geglu_bwd take this layer?  ('x2', the default: only where the layer's data gradient runs on two fp16 products and
    its weight-gradient launch reports the operand scale of dy.)"""
    M, K = dy.shape
    F = W2.shape[1]
    if not (FUSE_GEGLU_BWD and _geglu_wide_ok(dy, F, K) and W2.is_contiguous() and h.is_contiguous()):
        return False
    if FUSE_GEGLU_BWD == 'x2':
        if not (LIN_DGRAD_X2 and DGRAD_PRECISION == 'fp16x2' and WGRAD_PRECISION == 'fp16' and GENERIC_WGRAD_F16 and M >= WIDE_MIN_M):
            return False
        Wt = _bf16_weight(W2, True)
        return _F16CACHE.get((Wt.data_ptr(), tuple(Wt.shape))) is not None
    return True


def linear_dgrad_geglu_bwd(dy, W2, h, sc_dy=None):
    """dh [M][2 F] = GEGLU'(h) applied to d(gg) = dy [M][K] @ W2 [K][F] (data gradient of FeedForward's down-projection + GEGLU's
    backward): one launch on the wide kernel where it applies, else None (the caller takes the two-pass route).  sc_dy: the operand
    scale of dy (the weight-gradient launch of the same dy reports it: _LAST_LIN_DY_SCALE) -> the two-fp16-product arithmetic of the
    layer's plain data gradient (vxb_gemm_wide_f16x2_f32), same bits as that launch followed by geglu_bwd."""
    M, K = dy.shape
    F = W2.shape[1]
    if not (FUSE_GEGLU_BWD and _geglu_wide_ok(dy, F, K) and W2.is_contiguous() and h.is_contiguous()):
        return None
    if sc_dy is not None and LIN_DGRAD_X2 and DGRAD_PRECISION == 'fp16x2':
        Wt = _bf16_weight(W2, True)
        f16 = _F16CACHE.get((Wt.data_ptr(), tuple(Wt.shape)))
        if f16 is not None:
            dh = torch.empty_like(h)
            _lib.set_meta('gemm_dgrad %dx%dx%d' % (M, F, K), 2.0 * M * F * K)
            call('vxb_gemm_wide_geglu_bwd_f16x2_f32', dy, dy.stride(0), f16[0], h, dh, M, F, K, sc_dy)
            return dh
    wf = gemm_wfrag(_bf16_weight(W2, True))
    if wf is None:
        return None
    dh = torch.empty_like(h)
    _lib.set_meta('gemm_dgrad %dx%dx%d' % (M, F, K), 2.0 * M * F * K)
    call('vxb_gemm_wide_geglu_bwd_f32', dy, dy.stride(0), wf, h, dh, M, F, K)
    return dh


def geglu_fwd(h):
    rows, F2 = h.shape
    out = torch.empty((rows, F2 // 2), dtype=torch.float32, device=h.device)
    call('vxb_geglu_fwd_f32', h, out, rows, F2 // 2)
    return out


def geglu_bwd(h, dout):
    dh = torch.empty_like(h)
    call('vxb_geglu_bwd_f32', h, dout, dh, h.shape[0], h.shape[1] // 2)
    return dh


def lrelu_bwd_(dy, y):
    call('vxb_lrelu_bwd_f32', dy, y, dy, dy.numel(), LRELU_SLOPE)
    return dy


def axpy_(dst, src, alpha=1.0):
    call('vxb_axpy_f32', dst, src, dst.numel(), float(alpha))
    return dst


# --------------------------------------------------------------------------------------------- conv family
def conv_weight_fwd(W):
    """[Co,Ci,k,k,k] -> [(tap, ci)][co]  (tiny re-layout, done with torch views + one copy).  Memoised per weight tensor until new_step():
    a second call on the same weights -- the evaluation agent's act() with frozen weight preparation -- gets the SAME tensor back, so that
    the address-keyed caches downstream (to_bf16_nk, gemm_wfrag) hit instead of growing by one entry per call."""
    key = (W.data_ptr(), tuple(W.shape), 'conv_fwd')
    hit = _WCACHE.get(key)
    if hit is not None:
        return hit[0]
    Co, Ci = W.shape[:2]
    out = W.reshape(Co, Ci, -1).permute(2, 1, 0).reshape(-1, Co).contiguous()
    out._vxb_keep = True                     # (to_bf16_nk memoises only the layouts that are themselves kept)
    _WCACHE[key] = (out, W)                  # (keeps W alive: the key is its address)
    return out


def conv_weight_dgrad(W):
    """[Co,Ci,k,k,k] -> [(flipped tap, co)][ci]: weights of the data-gradient conv (stride 1)."""
    Co, Ci = W.shape[:2]
    return W.flip(2, 3, 4).reshape(Co, Ci, -1).permute(2, 0, 1).reshape(-1, Ci).contiguous()


def conv3d(src0, wt, N, B, S_in, S_out, kext, off, stride=1, replicate=True, bias=None, act=ACT_NONE, src1=None,
           out=None, ldc=None, accumulate=False, d2s=(0, 0), label=None):
    C0 = src0.shape[-1]
    C1 = src1.shape[-1] if src1 is not None else 0
    if _mm() and C0 % 32 == 0 and C1 % 32 == 0 and wt.dtype == torch.float32:
        return conv3d_bf16w(src0, to_bf16_nk(wt), N, B, S_in, S_out, kext, off, stride, replicate, bias, act, src1, out, ldc,
                            accumulate, d2s, label)
    if out is None:
        if d2s[0] > 0:
            Vf = S_out * d2s[0]
            out = torch.empty((B, Vf, Vf, Vf, d2s[1]), dtype=torch.float32, device=src0.device)
        else:
            out = torch.empty((B, S_out, S_out, S_out, N), dtype=torch.float32, device=src0.device)
    _f32c(src0, src1, wt, bias, out)
    _lib.set_meta(label or 'conv3d[k%d s%d %d->%d S%d%s]' % (kext, stride, C0 + C1, N, S_out, '' if replicate else ' dgrad'),
                  2.0 * B * S_out ** 3 * N * kext ** 3 * (C0 + C1))
    call('vxb_conv3d_f32', src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, int(replicate), wt, N, bias, out,
         ldc if ldc is not None else N, act, LRELU_SLOPE, int(accumulate), d2s[0], d2s[1])
    return out


def conv3d_wgrad(src0, dy, N, B, S_in, S_out, kext, off, stride=1, replicate=True, src1=None, ldy=None, d2s=(0, 0),
                 nsplit=None, label=None, force_bf16=False, phase_mask=None, flops_frac=1.0, dy_scale=None, possum_into=None,
                 grad_key=None, grad_is_src0=False, add_into=None):
    """returns dWt [(tap, ci)][N] (fp32, deterministic split reduction).  phase_mask (LDS-halo kernel, d2s only): int32
    [N / 64] tap masks of the polyphase structure -- the structurally zero (tap, phase) blocks come back as zeros."""
    C0 = src0.shape[-1]
    C1 = src1.shape[-1] if src1 is not None else 0
    K = kext ** 3 * (C0 + C1)
    P = B * S_out ** 3
    mode = force_bf16 if isinstance(force_bf16, str) else ('bf16' if force_bf16 else PRECISION)
    if (HALO_CONV and mode in ('bf16', 'bf16x3') and kext == 3 and stride == 1 and C0 % 16 == 0 and C1 % 16 == 0
            and N % 64 == 0 and (d2s[0] == 0 or d2s[1] == 64) and S_out >= 16 and (ldy is None or ldy % 4 == 0)):
        blocks = ((C0 + C1) // 16) * (N // 64)
        ntiles = int(_lib.lib().vxb_conv3_wgrad_halo_tiles(B, S_out, int(mode == 'bf16x3')))
        ns = nsplit if nsplit is not None else max(1, min(256, (1024 + blocks - 1) // blocks, ntiles // 8))
        if nsplit is None:
            # (column blocks x slices) a multiple of 8 lets the kernel co-locate the channel-chunk blocks that share dY
            # tiles on one XCD (see wgrad_halo.hip)
            import math
            step = 8 // math.gcd(N // 64, 8)
            ns = min(((ns + step - 1) // step) * step, max(step, (ntiles // step) * step))
        if phase_mask is not None and d2s[0] > 0:
            part = torch.zeros((ns, K, N), dtype=torch.float32, device=src0.device)      # skipped blocks stay zero
        else:
            part, phase_mask, flops_frac = torch.empty((ns, K, N), dtype=torch.float32, device=src0.device), None, 1.0
        lbl = label or 'conv3d_wgrad[k%d s%d %d->%d S%d]' % (kext, stride, C0 + C1, N, S_out)
        f16 = mode == 'bf16x3' and WGRAD_PRECISION == 'fp16' and dy.is_contiguous()
        if f16:
            sc = dy_scale
            if sc is None:                           # (the caller may already hold the scale of this dY: absmax_scale)
                _lib.set_meta(lbl, 0.0)
                sc = absmax_scale(dy)
            _lib.set_meta(lbl, 2.0 * P * N * K * flops_frac)
            call('vxb_conv3_wgrad_halo_f16_f32', src0, src1, C0, C1, B, S_in, S_out, off, int(replicate), dy, N,
                 ldy if ldy is not None else N, d2s[0], d2s[1], part, ns, phase_mask, sc)
            out = torch.empty((K, N), dtype=torch.float32, device=src0.device)
            sum_splits(part, ns, K * N, out, alpha=sc[1:])
            return out
        _lib.set_meta(lbl, 2.0 * P * N * K * flops_frac)
        call('vxb_conv3_wgrad_halo_bf16x3_f32' if mode == 'bf16x3' else 'vxb_conv3_wgrad_halo_bf16_f32', src0, src1, C0, C1,
             B, S_in, S_out, off, int(replicate), dy, N, ldy if ldy is not None else N, d2s[0], d2s[1], part, ns, phase_mask)
        if ns == 1:
            return part[0]
        out = torch.empty((K, N), dtype=torch.float32, device=src0.device)
        sum_splits(part, ns, K * N, out)
        return out
    if (HALO_CONV and K5_HALO_WGRAD and mode == 'bf16x3' and WGRAD_PRECISION == 'fp16' and kext == 5 and stride == 1 and S_in == S_out
            and S_out >= 16 and C0 % 16 == 0 and C1 % 16 == 0 and N % 64 == 0 and d2s[0] == 0 and dy.is_contiguous()
            and (ldy is None or ldy == N)):
        # 5^3 taps as eight shifted 3^3 blocks on the LDS-halo kernel, one launch (wgrad_halo.hip: vxb_conv3_wgrad_halo5_f16_f32)
        blocks = 8 * max(1, (C0 + C1) // 32) * (N // 64)
        ntiles = int(_lib.lib().vxb_conv3_wgrad_halo_tiles(B, S_out, 1))
        ns = nsplit if nsplit is not None else max(1, min(64, (512 + blocks - 1) // blocks, ntiles // 8))
        part = torch.empty((ns, K, N), dtype=torch.float32, device=src0.device)
        lbl = label or 'conv3d_wgrad[k%d s%d %d->%d S%d]' % (kext, stride, C0 + C1, N, S_out)
        sc = dy_scale
        if sc is None:
            _lib.set_meta(lbl, 0.0)
            sc = absmax_scale(dy)
        _lib.set_meta(lbl, 2.0 * P * N * K)
        call('vxb_conv3_wgrad_halo5_f16_f32', src0, src1, C0, C1, B, S_out, off, int(replicate), dy, N, N, part, ns,
             k5_shift_rows(src0.device), sc)
        out = torch.empty((K, N), dtype=torch.float32, device=src0.device)
        sum_splits(part, ns, K * N, out, alpha=sc[1:])
        return out
    if nsplit is None:
        tiles = ((K + 127) // 128) * ((N + 127) // 128 if N > 64 else 1)
        nsplit = max(1, min(64, 1024 // max(tiles, 1), (P + 4095) // 4096))
    part = torch.empty((nsplit, K, N), dtype=torch.float32, device=src0.device)
    _lib.set_meta(label or 'conv3d_wgrad[k%d s%d %d->%d S%d]' % (kext, stride, C0 + C1, N, S_out), 2.0 * P * N * K)
    entry = {'bf16': 'vxb_conv3d_wgrad_bf16_f32', 'bf16x3': 'vxb_conv3d_wgrad_bf16x3_f32'}.get(
        force_bf16 if isinstance(force_bf16, str) else ('bf16' if force_bf16 else PRECISION), 'vxb_conv3d_wgrad_f32')
    if (entry == 'vxb_conv3d_wgrad_bf16x3_f32' and WGRAD_PRECISION == 'fp16' and GENERIC_WGRAD_F16 and grad_key is not None
            and (dy if not grad_is_src0 else src0).is_contiguous()):
        grad = src0 if grad_is_src0 else dy
        use = _GRAD_USE.get(grad_key, 0)
        _GRAD_USE[grad_key] = use + 1
        skey = (grad_key, use)
        st = _GRAD_SCALE.get(skey)
        if st is None or st[0].device != grad.device:
            st = _GRAD_SCALE[skey] = [absmax_scale(grad), torch.empty(2, dtype=torch.float32, device=grad.device)]
        cur, nxt = st
        aws = torch.empty(int(_lib.lib().vxb_conv3d_wgrad_f16_amax_words(C0, C1, kext, N, nsplit, int(grad_is_src0))),
                          dtype=torch.float32, device=grad.device)
        psum = torch.empty((nsplit, K), dtype=torch.float32, device=src0.device) if possum_into is not None else None
        _lib.set_meta(label or 'conv3d_wgrad[k%d s%d %d->%d S%d]' % (kext, stride, C0 + C1, N, S_out), 2.0 * P * N * K)
        # the partial sums are folded (times 1 / scale) by the same finishing launch that turns this launch's maxima into the next scale
        out = add_into if add_into is not None else torch.empty((K, N), dtype=torch.float32, device=src0.device)
        call('vxb_conv3d_wgrad_f16_f32', src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, int(replicate), dy, N,
             ldy if ldy is not None else N, d2s[0], d2s[1], part, nsplit, psum, cur, int(grad_is_src0), nxt, aws, out, int(add_into is not None))
        st[0], st[1] = nxt, cur                       # the next call at this site uses the maximum this launch saw
        _LAST_GRAD_SCALE[0] = cur if grad_is_src0 else None
        if psum is not None:
            sum_splits(psum, nsplit, K, possum_into, accumulate=True)
        return None if add_into is not None else out
    if entry != 'vxb_conv3d_wgrad_f32':
        # possum_into [C0] (plain GEMM form: the weight gradient of a linear layer, src0 = its dY): += the sums of src0 over the
        # positions, i.e. the bias gradient, from this launch (no second pass over dY)
        psum = torch.empty((nsplit, K), dtype=torch.float32, device=src0.device) if possum_into is not None else None
        call(entry, src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, int(replicate), dy, N,
             ldy if ldy is not None else N, d2s[0], d2s[1], part, nsplit, psum)
        if psum is not None:
            sum_splits(psum, nsplit, K, possum_into, accumulate=True)
    else:
        assert possum_into is None
        call(entry, src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, int(replicate), dy, N,
             ldy if ldy is not None else N, d2s[0], d2s[1], part, nsplit)
    if nsplit == 1:
        return part[0]
    out = torch.empty((K, N), dtype=torch.float32, device=src0.device)
    sum_splits(part, nsplit, K * N, out)
    return out


K5_HALO_WGRAD = True      # 5^3 stride-1 weight gradients on the LDS-halo kernel (eight shifted 3^3 blocks); False: generic gather kernel
_K5_ROWS = {}


def k5_shift_rows(dev):
    """int32 [8][27]: row block (kd * 5 + kh) * 5 + kw of (shift, 3^3 tap), -1 where the tap belongs to another shift
    (include/voxactb_hip.h: vxb_conv3_wgrad_halo5_f16_f32)."""
    t = _K5_ROWS.get(dev)
    if t is None:
        rows = np.full((8, 27), -1, dtype=np.int32)
        for sh in range(8):
            bits = ((sh >> 2) & 1, (sh >> 1) & 1, sh & 1)
            for tap in range(27):
                loc = (tap // 9, (tap // 3) % 3, tap % 3)
                g = []
                for b, l in zip(bits, loc):
                    if b and l == 0:          # offset 0 belongs to the block without the shift
                        g = None
                        break
                    g.append(l + 2 * b)       # offsets {-2, -1, 0} -> 0..2; {+1, +2} -> 3..4
                if g is not None:
                    rows[sh, tap] = (g[0] * 5 + g[1]) * 5 + g[2]
        assert sorted(rows[rows >= 0].tolist()) == list(range(125))
        t = _K5_ROWS[dev] = torch.from_numpy(rows).to(dev)
    return t


def fold_pad(src, Sp, Cs, c0, dst, B, S, C, pad, accumulate=False, lrelu_of=None):
    call('vxb_fold_pad_f32', src, Sp, Cs, c0, dst, lrelu_of, B, S, C, pad, int(accumulate), LRELU_SLOPE)
    return dst


def polyphase_tables(k, s):
    """Per-axis matrices of upsample(x s, trilinear, align_corners=False) o conv(k, replicate pad k//2):
    fine output s*q + r, tap t reads fine position Y = r + t - k//2 (relative to s*q), which interpolates the
    low-res cells floor(src), floor(src)+1 with src = (Y + 0.5)/s - 0.5 (network_utils.py:245-250; replicate
    extension of the low-res grid reproduces both the border clamp of the interpolation and the conv padding).
    Returns (L [s][k][kl] float32, R) with kl = 2R+1 low-res taps, j = cell offset + R."""
    p = k // 2
    ent = []
    for r in range(s):
        for t in range(k):
            Y = r + t - p
            src = (Y + 0.5) / s - 0.5
            i0 = int(np.floor(src))
            lam = src - i0
            ent.append((r, t, i0, 1.0 - lam))
            ent.append((r, t, i0 + 1, lam))
    R = max(abs(e[2]) for e in ent if e[3] != 0.0)
    kl = 2 * R + 1
    L = np.zeros((s, k, kl), np.float64)
    for r, t, j, w in ent:
        if w != 0.0:
            L[r, t, j + R] += w
    return L.astype(np.float32), R


def polyphase_weights(W, L, s, kl):
    Co, Ci, k = W.shape[0], W.shape[1], W.shape[2]
    Weff = torch.empty((kl ** 3 * Ci, s ** 3 * Co), dtype=torch.float32, device=W.device)
    call('vxb_polyphase_weights_f32', W, L, Weff, Ci, Co, k, s, kl)
    return Weff


def polyphase_weights_bwd(dWeff, L, dW, s, kl):
    Co, Ci, k = dW.shape[0], dW.shape[1], dW.shape[2]
    call('vxb_polyphase_weights_bwd_f32', dWeff, L, dW, Ci, Co, k, s, kl)


def polyphase_dgrad_weights(Weff, Ci, Co, s, kl):
    """[(j3, ci)][(r3, co)] -> [(t'3, co)][ci] with t' = s*(kl-1-j) + r per axis (stride-s, zero-pad data-gradient conv)."""
    w = Weff.view(kl, kl, kl, Ci, s, s, s, Co).flip(0, 1, 2)
    return w.permute(0, 4, 1, 5, 2, 6, 7, 3).reshape((s * kl) ** 3 * Co, Ci).contiguous()


def polyphase_dgrad_weights_lowres(Weff, Ci, Co, s, kl):
    """[(j3, ci)][(r3, co)] -> [(j''3, (r3, co))][ci] with j'' = kl-1-j per axis: the same data gradient written as a
    stride-1 zero-pad conv over the LOW-RES grid whose input channels are the s^3 phases x co of the fine dY
    (read in place by space-to-depth, see conv3_s2d)."""
    w = Weff.view(kl, kl, kl, Ci, s, s, s, Co).flip(0, 1, 2)
    return w.permute(0, 1, 2, 4, 5, 6, 7, 3).reshape(kl ** 3 * s ** 3 * Co, Ci).contiguous()


_POLY = {}


def polyphase_structure(k, s, dev):
    """Zero structure of the polyphase weights (network_utils.py:245-250): along one axis fine phase r only reaches the
    low-res offsets j with L[r][:, j] != 0 (k = s = 5: r = 0 -> {-1, 0}, r = 1..3 -> {-1, 0, 1}, r = 4 -> {0, 1}), so only
    17.6 of the 27 (tap, phase) weight blocks are non-zero on average.  Returns
      phase_mask [s^3] int : bit (jd*kl + jh)*kl + jw set <=> phase (rd, rh, rw) has weights at that low-res tap
      perm [s^3] int32     : column-block order that pairs phases with (nearly) the same footprint in one 128-column tile
      tile_mask int32      : union of the two phase masks per 128-column tile of that order
      frac                 : non-zero fraction of the kl^3 * s^3 blocks (the algorithmic flops of the polyphase form)."""
    key = (k, s, str(dev))
    st = _POLY.get(key)
    if st is not None:
        return st
    L, R = polyphase_tables(k, s)
    kl = 2 * R + 1
    reach = (L != 0).any(axis=1)                                   # [s][kl]
    nph = s ** 3
    pm = []
    for ph in range(nph):
        rd, rh, rw = ph // (s * s), (ph // s) % s, ph % s
        m = 0
        for jd in range(kl):
            for jh in range(kl):
                for jw in range(kl):
                    if reach[rd, jd] and reach[rh, jh] and reach[rw, jw]:
                        m |= 1 << ((jd * kl + jh) * kl + jw)
        pm.append(m)
    # pair equal footprints first, then the left-overs greedily by the size of the union
    groups = {}
    for ph, m in enumerate(pm):
        groups.setdefault(m, []).append(ph)
    order, left = [], []
    for m in sorted(groups):
        l = list(groups[m])
        while len(l) >= 2:
            order += [l.pop(0), l.pop(0)]
        left += l
    pc = lambda x: bin(x).count('1')
    while len(left) >= 2:
        a = left.pop(0)
        b = min(left, key=lambda c: 2 * pc(pm[a] | pm[c]) - pc(pm[a]) - pc(pm[c]))
        left.remove(b)
        order += [a, b]
    order += left
    tm = [pm[order[i]] | (pm[order[i + 1]] if i + 1 < nph else 0) for i in range(0, nph, 2)]
    # the wide kernel (gemm_wide.hip) walks the UNION of eight phases' taps per workgroup: groups of eight grown greedily from the
    # largest remaining footprint by whatever adds the fewest taps (k = s = 5: 83 % of the wave slots do real work, 74 % in pair order)
    rem, order8 = set(range(nph)), []
    while rem:
        seed = max(rem, key=lambda p: (pc(pm[p]), -p))
        grp, u = [seed], pm[seed]
        rem.remove(seed)
        while len(grp) < 8 and rem:
            nxt = min(rem, key=lambda p: (pc(u | pm[p]) - pc(u), -pc(pm[p]), p))
            grp.append(nxt)
            rem.remove(nxt)
            u |= pm[nxt]
        order8 += grp
    st = dict(kl=kl, R=R, phase_mask=pm, order=order,
              perm=torch.tensor(order, dtype=torch.int32, device=dev), perm_long=torch.tensor(order, dtype=torch.int64, device=dev),
              tile_mask=torch.tensor(tm, dtype=torch.int32, device=dev), phase_mask_t=torch.tensor(pm, dtype=torch.int32, device=dev),
              order8=order8, perm8=torch.tensor(order8, dtype=torch.int32, device=dev),
              perm8_long=torch.tensor(order8, dtype=torch.int64, device=dev),
              block_mask8=torch.tensor([pm[p] for p in order8], dtype=torch.int32, device=dev),
              frac=sum(pc(m) for m in pm) / float(nph * kl ** 3), tile_frac=sum(pc(m) for m in tm) * 2 / float(2 * len(tm) * kl ** 3))
    _POLY[key] = st
    return st


WIDE_POLY = os.environ.get('VOXACTB_WIDE_POLY', '1') != '0'     # polyphase up-conv forward on the wide kernel (gemm_wide.hip); '0': direct-to-LDS 128^2 kernel
POLY_SPARSE = True       # skip the structurally zero (tap, phase) blocks of the polyphase up-conv


def polyphase_fwd_ok(C, Cout, kl, B, G):
    return (POLY_SPARSE and DL_GEMM and _mm() and C % 32 == 0 and Cout == 64 and kl ** 3 <= 32 and B * G ** 3 >= 128)


def conv3_polyphase_fwd(z, Weff, Cout, B, G, k, s, bias, act=ACT_NONE, label=None):
    """upsample(x s, trilinear) -> conv(k, replicate) as the low-res kl^3 conv with s^3 * Cout phase columns and a
    depth-to-space store (conv3d(z, Weff, ..., d2s=(s, Cout))), visiting only the non-zero (tap, phase) weight blocks:
    direct-to-LDS implicit GEMM with a tap mask per 128-column tile.  Bit-identical to the dense evaluation (the skipped
    products are exact zeros)."""
    C = z.shape[-1]
    st = polyphase_structure(k, s, z.device)
    kl, R = st['kl'], st['R']
    N, K = s ** 3 * Cout, kl ** 3 * C
    npl = 2 if PRECISION == 'bf16x3' else 1
    lbl = label or 'conv3_polyphase[k%d s%d %d->%d G%d]' % (k, s, C, Cout, G)
    _lib.set_meta(lbl, 0.0)
    out = torch.empty((B, G * s, G * s, G * s, Cout), dtype=torch.float32, device=z.device)
    if WIDE_POLY and GEMM_BD and npl == 2 and Cout == 64 and z.is_contiguous() and K % 16 == 0:
        # 128 x 512 workgroup tiles, one phase per wave, A gathered + split in the kernel (gemm_wide.hip: conv_poly_wide_x3_kernel);
        # column blocks in the order that keeps the eight footprints of a workgroup alike
        if WEIGHT_GATHER and Weff.is_contiguous() and K % 16 == 0:
            # Weff [K][N] -> column blocks in perm8 order, hi | lo planes, MFMA fragment order: one gather pass through a cached table
            # (was: transposing gather 166 us + split + fragment shuffle 54 us per step)
            numel = Weff.numel()
            ck = ('polywf', Weff.data_ptr(), tuple(Weff.shape))
            hit = _FCACHE.get(ck)          # (inference with frozen weights keeps W_eff and this; a training step clears it: new_step())
            if hit is not None:
                wf = hit[0]
            else:
                idx = traced_index(('polyf', k, s, C, Cout, str(z.device)), tuple(Weff.shape),
                                   lambda I: _wfrag_index(I.t().view(s ** 3, Cout, K).index_select(0, st['perm8_long']).view(N, K), numel), z.device)
                wf = gather_cvt(Weff, idx, 1, numel)
                _FCACHE[ck] = (wf, Weff)
        else:
            wt = Weff.t().view(s ** 3, Cout, K).index_select(0, st['perm8_long']).view(N, K)
            wf = gemm_wfrag(split_planes(wt, npl))
        _lib.set_meta(lbl, 2.0 * B * G ** 3 * N * kl ** 3 * C * st['frac'])
        call('vxb_conv3_poly_wide_bf16x3_f32', z, C, B, G, kl, -R, 1, wf, N, bias, out, act, LRELU_SLOPE, s, st['block_mask8'], st['perm8'])
        return out
    wt = Weff.t().view(s ** 3, Cout, K).index_select(0, st['perm_long']).view(N, K)      # [(phase in tile order, co)][K]
    wb = split_planes(wt, npl)
    _lib.set_meta(lbl, 0.0)
    planes = split_planes(z.reshape(-1, C), npl)
    _lib.set_meta(lbl, 2.0 * B * G ** 3 * N * kl ** 3 * C * st['frac'])
    call('vxb_conv3d_dl_f32', planes, C, B, G, G, 1, kl, -R, 1, wb, gemm_wfrag(wb), npl, N, bias, out, N, act, LRELU_SLOPE, 0, s, Cout,
         _zeros16(z.device), st['tile_mask'], st['perm'])
    return out


HALO_WD = True       # halo conv: B fragments straight from global memory (pre-shuffled weights), no barrier in the tap loop


def halo_wfrag(wb, Ct):
    """weights bf16 [N][27*Ct] ('bf16') or planes [2][N][27*Ct] ('bf16x3') -> the same values in MFMA fragment order
    [N/64][chunk][tap][column tile 2][k half | plane 2][lane = hi*32 + lq][8]: lane (lq, hi) of column tile j finds its
    8 k-values of tap t, chunk c at a lane-contiguous 16-byte slot (one coalesced 1 KB load per fragment)."""
    if not HALO_WD:
        return None
    if wb.dim() == 3:       # x3: chunk = 16 channels, second fragment = lo plane
        N = wb.shape[1]
        v = wb.view(2, N // 64, 2, 32, 27, Ct // 16, 2, 8)              # (p, nb, j, lq, tap, ch, hi, e)
        return v.permute(1, 5, 4, 2, 0, 6, 3, 7).contiguous()             # (nb, ch, tap, j, p, hi, lq, e)
    N = wb.shape[0]
    v = wb.view(N // 64, 2, 32, 27, Ct // 32, 2, 2, 8)                  # (nb, j, lq, tap, ch, f, hi, e)
    return v.permute(0, 4, 3, 1, 5, 6, 2, 7).contiguous()                 # (nb, ch, tap, j, f, hi, lq, e)


def halo_wfrag_x2(w16, Ct):
    """fp16 weights [N][27*Ct] -> MFMA fragment order for the 'fp16x2' product mode of the LDS-halo conv (the input as an fp16
    hi | lo pair, 16 channels per chunk, ONE weight fragment per (tap, column tile)): [N/64][chunk][tap][column tile 2][lane = hi*32 +
    lq][8]."""
    N = w16.shape[0]
    v = w16.view(N // 64, 2, 32, 27, Ct // 16, 2, 8)                    # (nb, j, lq, tap, ch, hi, e)
    return v.permute(0, 4, 3, 1, 5, 2, 6).contiguous()                  # (nb, ch, tap, j, hi, lq, e)


# product type of the data gradients that PROPAGATE (final's d(u0), the up-conv's): 'fp16x2' = the gradient operand as an fp16 hi + lo
# pair (pre-scaled by a device-side power of two), the weights as one fp16 value: two MFMAs per product instead of bf16x3's three;
# against the reference digests F5g / F5c3 rounding the WEIGHTS of a data gradient to 11 bits moves no gate, rounding dY does
# (tools/experiments/emu_precision.py --round5, profiles/r03_emu_6_two_product_dgrads.log).  'bf16x3': round 2's arithmetic.
DGRAD_PRECISION = os.environ.get('VOXACTB_DGRAD_PRECISION', 'fp16x2')


def dgrad_fold_ok(C_dy, N, S):
    """the fused data-gradient + fold kernel: bf16 modes, 3x3x3 / pad 1, border groups inside one 4x8x8 tile."""
    So = S + 2
    return (HALO_CONV and _mm() and C_dy % 32 == 0 and N in (64, 128) and S >= 16
            and (So - 2) // 4 == (So - 1) // 4 and (So - 2) // 8 == (So - 1) // 8)


def conv3_dgrad_fold(dy, wt_dgrad, B, S, N, dsts, label=None, dy_scale=None, leaf_blocks=(), scale_blocks=(), colsum_into=None,
                     wgin=None):
    """fold_pad(conv3d(dy, wt_dgrad, zero pad, S+2), pad=1) without the padded tensor: dsts = [(dst [B,S,S,S,64],
    accumulate, lrelu_of or None)] per 64-column block (1 or 2 entries).
    leaf_blocks: the column blocks whose result only feeds a weight gradient (nothing propagates from them); with
    WGRAD_PRECISION == 'fp16' those are evaluated with a single fp16 product per term, dy scaled by dy_scale (absmax_scale of dy,
    computed here when None) -- the others keep the bf16x3 triple.  scale_blocks: the (non-leaf) blocks whose destination's own fp16
    operand scale is wanted (taken in the epilogue while the tensor is written); returns {block: [scale, 1 / scale]}.
    colsum_into {block: tensor [64]}: += the column sums of that block's destination (the bias gradient of the conv it is the
    pre-activation gradient of) -- from the epilogue on the split path, by a colsum pass otherwise.
    wgin {block: (y, x, dW, db)} (leaf blocks on the fp16 path only; see wgin_fold_ok): the block is the data gradient of y = lrelu(1x1x1
    conv of the detached x [B,S,S,S,10]); instead of being stored to its dst (which may then be None) it is multiplied with
    LeakyReLU'(y) and x in the epilogue and dW [64][10] / db [64] are accumulated."""
    C0 = dy.shape[-1]
    wb = to_bf16_nk(wt_dgrad)
    x3 = wb.dim() == 3
    d0, a0, y0 = dsts[0]
    d1, a1, y1 = dsts[1] if len(dsts) > 1 else (None, False, None)
    lbl = label or 'conv3d_bf16[k3 s1 %d->%d S%d dgrad+fold]' % (C0, N, S + 2)
    # algorithmic work = the data gradient on the S^3 grid (the kernel evaluates it on the zero-padded (S+2)^3 domain and folds the
    # border back in its epilogue: those 6 % extra MFMAs are not counted)
    flops = 2.0 * B * S ** 3 * N * 27 * C0
    _lib.set_meta(lbl, flops)
    wf = halo_wfrag(wb, C0)
    scales = {}
    colsum_into = colsum_into or {}
    if x3 and wf is not None and WGRAD_PRECISION == 'fp16' and leaf_blocks and N == 64 * len(dsts) and dy.is_contiguous():
        sc = dy_scale
        for nb, (dst, acc, yv) in enumerate(dsts):
            if nb in leaf_blocks:
                if sc is None:
                    _lib.set_meta(lbl, 0.0)
                    sc = absmax_scale(dy)
                if LEAF_WINOGRAD and S % 2 == 0 and not (wgin and nb in wgin):
                    _lib.set_meta(lbl, flops / len(dsts))
                    call('vxb_conv3_dgrad_fold_f16_wg_f32', dy, C0, B, S, halo_wfrag_x2_wg(wt_dgrad[:, 64 * nb:64 * nb + 64], C0), dst, yv,
                         int(acc), LRELU_SLOPE, sc)
                    continue
                w16 = wt_dgrad[:, 64 * nb:64 * nb + 64].t().contiguous().half()           # [64][27 C0]
                wf16 = halo_wfrag(w16, C0)
                _lib.set_meta(lbl, flops / len(dsts))
                if wgin and nb in wgin:
                    yw, xw, dWw, dbw = wgin[nb]
                    ws = torch.empty(704 * (int(_lib.lib().vxb_conv3_dgrad_fold_blocks(B, S, 64)) + 512), dtype=torch.float32, device=dy.device)
                    call('vxb_conv3_dgrad_fold_f16_wgin_f32', dy, C0, B, S, wf16, yw, xw, LRELU_SLOPE, sc, ws, dWw, dbw)
                    continue
                call('vxb_conv3_dgrad_fold_f16_f32', dy, C0, B, S, wf16, dst, yv, int(acc), LRELU_SLOPE, sc)
            else:
                dsc = sws = cs = cws = None
                nblk = int(_lib.lib().vxb_conv3_dgrad_fold_blocks(B, S, 64))
                if nb in scale_blocks:
                    dsc = scales[nb] = torch.empty(2, dtype=torch.float32, device=dy.device)
                    sws = torch.empty(nblk, dtype=torch.float32, device=dy.device)
                if nb in colsum_into:
                    cs = colsum_into[nb]
                    cws = torch.empty(64 * (nblk + 64), dtype=torch.float32, device=dy.device)
                if DGRAD_PRECISION == 'fp16x2':
                    if sc is None:
                        _lib.set_meta(lbl, 0.0)
                        sc = absmax_scale(dy)
                    _lib.set_meta(lbl, flops / len(dsts))
                    if DGRAD_WINOGRAD and S % 2 == 0:
                        call('vxb_conv3_dgrad_fold_f16x2_wg_f32', dy, C0, B, S, halo_wfrag_x2_wg(wt_dgrad[:, 64 * nb:64 * nb + 64], C0), dst, yv,
                             int(acc), LRELU_SLOPE, sc, dsc, sws, cs, cws)
                        continue
                    wf2 = halo_wfrag_x2(wt_dgrad[:, 64 * nb:64 * nb + 64].t().contiguous().half(), C0)
                    call('vxb_conv3_dgrad_fold_f16x2_f32', dy, C0, B, S, wf2, dst, yv, int(acc), LRELU_SLOPE, sc, dsc, sws, cs, cws)
                    continue
                _lib.set_meta(lbl, flops / len(dsts))
                call('vxb_conv3_dgrad_fold_f32', dy, C0, B, S, wb, 1, 64, dst, None, yv, None, int(acc), 0, LRELU_SLOPE, wf[nb:nb + 1],
                     dsc, sws, cs, cws)
        return scales
    _lib.set_meta(lbl, flops)
    call('vxb_conv3_dgrad_fold_f32', dy, C0, B, S, wb, int(x3), N, d0, d1, y0, y1, int(a0), int(a1), LRELU_SLOPE, wf, None, None, None, None)
    for nb, into in colsum_into.items():
        colsum(dsts[nb][0].view(-1, 64), into, accumulate=True)
    return scales


WGIN_FOLD = os.environ.get('VOXACTB_WGIN_FOLD', '0') != '0'     # d(d0) of `final` straight into the input conv's weight gradient (no dd0 tensor)


def wgin_fold_ok(C, N, S, Cin):
    """conv3_dgrad_fold(..., leaf_blocks=(0,), wgin={0: ...}) will take the fused route (mirrors its fp16-path test)"""
    return (WGIN_FOLD and WGRAD_PRECISION == 'fp16' and PRECISION == 'bf16x3' and HALO_CONV and HALO_WD and C == 64 and N == 128
            and Cin == 10 and dgrad_fold_ok(C, N, S))


def s2d_halo_ok(kl, C, N):
    return HALO_CONV and _mm() and kl == 3 and C % 32 == 0 and N % 64 == 0


def s2d_taptab(k, s, dev, cpc, chunks_per_phase):
    """tap lists of the polyphase DATA gradient for the LDS-halo kernel (see voxactb_hip.h, taptab): the gradient's tap t''
    is the forward tap kl^3 - 1 - t'' (all three axes flipped).  Returns (table int32, ncls, tap_total, rows) with rows =
    index into the dense [chunk][27] fragment blocks of every listed tap, chunk by chunk."""
    key = ('tt', k, s, str(dev), cpc, chunks_per_phase)
    r = _POLY.get(key)
    if r is not None:
        return r
    st = polyphase_structure(k, s, dev)
    kl = st['kl']
    assert kl == 3
    nt = kl ** 3
    lists, cls_of, table, phase_rows = {}, [], [], []
    for ph, m in enumerate(st['phase_mask']):
        dm = sum(1 << (nt - 1 - t) for t in range(nt) if (m >> t) & 1)
        if dm not in lists:
            taps = [t for t in range(nt) if (dm >> t) & 1]
            pad = [t for t in range(nt) if not (dm >> t) & 1][:(-len(taps)) % 3]     # zero-weight taps of this footprint
            lists[dm] = (len(lists), taps + pad)
        cls_of.append(lists[dm][0])
    ncls = len(lists)
    for dm, (ci, taps) in sorted(lists.items(), key=lambda kv: kv[1][0]):
        ent = [(((t // 9) * 10 + (t // 3) % 3) * 12 + t % 3) * 40 for t in taps] + [0] * (32 - len(taps))
        ent[31] = len(taps)
        table += ent
    by_cls = {ci: taps for ci, taps in lists.values()}
    total, rows = 0, []
    for ph in range(s ** 3):
        taps = by_cls[cls_of[ph]]
        table += [cls_of[ph], total]
        for c in range(chunks_per_phase):
            rows += [(ph * chunks_per_phase + c) * nt + t for t in taps]
        total += chunks_per_phase * len(taps)
    r = (torch.tensor(table, dtype=torch.int32, device=dev), ncls, total, torch.tensor(rows, dtype=torch.int64, device=dev))
    _POLY[key] = r
    # listed taps per chunk, in chunk order (for the split-K boundaries)
    _POLY[('ttc',) + key[1:]] = [len(by_cls[cls_of[ph]]) for ph in range(s ** 3) for _ in range(chunks_per_phase)]
    return r


S2D_KSPLIT = int(os.environ.get('VOXACTB_S2D_KSPLIT', '8'))    # workgroups per tile of the tap-list data gradient (1 = no split)


def s2d_kparts(k, s, dev, cpc, chunks_per_phase, ksplit):
    """chunk boundaries [ksplit + 1] (device int32) that give every part about the same number of listed taps."""
    key = ('kp', k, s, str(dev), cpc, chunks_per_phase, ksplit)
    r = _POLY.get(key)
    if r is None:
        s2d_taptab(k, s, dev, cpc, chunks_per_phase)
        taps = _POLY[('ttc', k, s, str(dev), cpc, chunks_per_phase)]
        total, acc, bounds = sum(taps), 0, [0]
        for i, n in enumerate(taps):
            if len(bounds) < ksplit and acc + n / 2.0 >= total * len(bounds) / float(ksplit):
                bounds.append(i)
            acc += n
        while len(bounds) < ksplit:
            bounds.append(len(taps))
        bounds.append(len(taps))
        r = _POLY[key] = torch.tensor(bounds, dtype=torch.int32, device=dev)
    return r


def conv3_s2d(src_fine, wt, N, B, G, S_out, off, s, Cf, label=None, poly_k=None, dy_scale=None, weff_src=None):
    """out[B, S_out^3, N] = 3x3x3 zero-pad conv over the low-res grid G^3 whose input channel (phase, co) is read from
    src_fine [B, (G*s)^3, Cf] at fine voxel (q*s + r); wt fp32 [(tap, phase, co)][N].  LDS-halo kernel only (bf16 modes).
    poly_k: wt is the data gradient of the polyphase up-conv of a k^3 kernel -- only its non-zero (tap, phase) blocks are
    visited (POLY_SPARSE).  weff_src = (Weff, Ci, Co, s, kl) instead of wt (wt = None): wt = polyphase_dgrad_weights_lowres(*weff_src),
    never formed on the two-product path -- its fp16 fragments are gathered from Weff in one pass."""
    C0 = s ** 3 * Cf
    if (weff_src is not None and WEIGHT_GATHER and poly_k is not None and POLY_SPARSE and HALO_WD and S2D_KSPLIT > 1 and PRECISION == 'bf16x3'
            and DGRAD_PRECISION == 'fp16x2' and WGRAD_PRECISION == 'fp16' and src_fine.is_contiguous() and weff_src[0].is_contiguous()):
        Weff, Ci, Co, s_, kl = weff_src
        dev = src_fine.device
        lbl = label or 'conv3_s2d[k3 %d->%d S%d dgrad]' % (C0, N, S_out)
        tt, ncls, total, rows = s2d_taptab(poly_k, s, dev, 16, Cf // 16)
        frac = polyphase_structure(poly_k, s, dev)['frac']
        ks = S2D_KSPLIT
        kp = s2d_kparts(poly_k, s, dev, 16, Cf // 16, ks)

        def layout(I):
            f = halo_wfrag_x2(polyphase_dgrad_weights_lowres(I, Ci, Co, s_, kl).t().contiguous(), C0)
            return f.view(f.shape[0], f.shape[1] * 27, -1).index_select(1, rows).contiguous()
        idx = traced_index(('s2dw', poly_k, s, Ci, Co, kl, Cf, str(dev)), tuple(Weff.shape), layout, dev)
        wf2 = gather_cvt(Weff, idx, 0)
        out = torch.empty((B, S_out, S_out, S_out, N), dtype=torch.float32, device=dev)
        parts = torch.empty((ks, B, S_out, S_out, S_out, N), dtype=torch.float32, device=dev)
        sc = dy_scale
        if sc is None:
            _lib.set_meta(lbl, 0.0)
            sc = absmax_scale(src_fine)
        _lib.set_meta(lbl, 2.0 * B * S_out ** 3 * N * 27 * C0 * frac)
        call('vxb_conv3_s2d_splitk_f32', src_fine, C0, B, G, S_out, off, wf2, 3, N, parts, s, Cf, wf2, tt, ncls, total, ks, kp, sc)
        sum_splits(parts, ks, out.numel(), out)
        return out
    if wt is None:
        wt = polyphase_dgrad_weights_lowres(*weff_src)
    wb = to_bf16_nk(wt)
    out = torch.empty((B, S_out, S_out, S_out, N), dtype=torch.float32, device=src_fine.device)
    x3 = wb.dim() == 3
    lbl = label or 'conv3_s2d[k3 %d->%d S%d dgrad]' % (C0, N, S_out)
    _lib.set_meta(lbl, 0.0)
    wf = halo_wfrag(wb, C0)
    tt, ncls, total, frac = None, 0, 0, 1.0
    if poly_k is not None and POLY_SPARSE and wf is not None:
        cpc = 16 if x3 else 32
        tt, ncls, total, rows = s2d_taptab(poly_k, s, src_fine.device, cpc, Cf // cpc)
        wf = wf.view(wf.shape[0], wf.shape[1] * 27, -1).index_select(1, rows).contiguous()
        frac = polyphase_structure(poly_k, s, src_fine.device)['frac']
    if tt is not None and S2D_KSPLIT > 1:
        ks = S2D_KSPLIT
        kp = s2d_kparts(poly_k, s, src_fine.device, cpc, Cf // cpc, ks)
        parts = torch.empty((ks, B, S_out, S_out, S_out, N), dtype=torch.float32, device=src_fine.device)
        if x3 and DGRAD_PRECISION == 'fp16x2' and WGRAD_PRECISION == 'fp16' and src_fine.is_contiguous():
            # two fp16 products per term: src_fine * 2^k as an fp16 hi + lo pair, the weights as one fp16 value (single-plane fragments)
            sc = dy_scale
            if sc is None:
                _lib.set_meta(lbl, 0.0)
                sc = absmax_scale(src_fine)
            wf2 = halo_wfrag_x2(wt.t().contiguous().half(), C0)
            wf2 = wf2.view(wf2.shape[0], wf2.shape[1] * 27, -1).index_select(1, rows).contiguous()
            _lib.set_meta(lbl, 2.0 * B * S_out ** 3 * N * 27 * C0 * frac)
            call('vxb_conv3_s2d_splitk_f32', src_fine, C0, B, G, S_out, off, wf2, 3, N, parts, s, Cf, wf2, tt, ncls, total, ks, kp, sc)
        else:
            _lib.set_meta(lbl, 2.0 * B * S_out ** 3 * N * 27 * C0 * frac)
            call('vxb_conv3_s2d_splitk_f32', src_fine, C0, B, G, S_out, off, wb, int(x3), N, parts, s, Cf, wf, tt, ncls, total, ks, kp, None)
        sum_splits(parts, ks, out.numel(), out)
        return out
    _lib.set_meta(lbl, 2.0 * B * S_out ** 3 * N * 27 * C0 * frac)
    call('vxb_conv3_halo_bf16x3_f32' if x3 else 'vxb_conv3_halo_bf16w_f32', src_fine, None, C0, 0, B, G, S_out, off,
         0, wb, N, None, out, ACT_NONE, LRELU_SLOPE, s, Cf, 0, wf, tt, ncls, total)
    return out


def strided_dgrad_weights(W, s):
    """Data gradient of a stride-s conv as a stride-1 conv on the coarse grid with s^3 output phases:
    W [Co,Ci,k,k,k] -> wt [(u''3, co)][(r3, ci)], U = ceil(k/s) taps/axis, element = W[co,ci,t = s*(U-1-u'') + r] (0 if t >= k)."""
    Co, Ci, k = W.shape[0], W.shape[1], W.shape[2]
    U = (k + s - 1) // s
    Wp = torch.zeros((Co, Ci, U * s, U * s, U * s), dtype=W.dtype, device=W.device)
    Wp[:, :, :k, :k, :k] = W
    Wp = Wp.view(Co, Ci, U, s, U, s, U, s).flip(2, 4, 6)          # u'' = U-1-u
    return Wp.permute(2, 4, 6, 0, 3, 5, 7, 1).reshape(U ** 3 * Co, s ** 3 * Ci).contiguous(), U


PATCH_WGRAD_FUSE = os.environ.get('VOXACTB_PATCH_WGRAD_FUSE', '1') != '0'    # patchify data gradient folded into the input conv's weight gradient


def patch_dgrad_input_wgrad_ok(k, s, C, Cin):
    return PATCH_WGRAD_FUSE and WGRAD_PRECISION == 'fp16' and PRECISION == 'bf16x3' and k == s and C == 64 and Cin == 10


PATCH_WGRAD_WEIGHT = os.environ.get('VOXACTB_PATCH_WGRAD_WEIGHT', '1') != '0'      # '0': the patchify weight gradient as its own launch (A/B)


def patch_dgrad_input_wgrad(dpatch, Wp, d0, vox, dW, db, B, V, G, k, pad, dWp=None):
    """dW [64][10] += / db [64] += the patchify data gradient's share of the input conv's weight / bias gradient, straight from
    dpatch [B, G^3, 64] (patch_wgrad.hip: vxb_patch_dgrad_input_wgrad_f32) -- the 4.7 GB data gradient tensor and its padding
    adjoint are never formed.  dWp (the patchify weight's gradient tensor [64][64][k][k][k], contiguous): += the patchify conv's own
    weight gradient from the same pass over d0."""
    wt = Wp.reshape(Wp.shape[0], Wp.shape[1], k ** 3).permute(2, 1, 0).contiguous().to(torch.float16)      # [tap][c][kout]
    ns = max(1, min(64, 1024 // (k ** 3)))
    ws = torch.empty(int(_lib.lib().vxb_patch_dgrad_input_wgrad_ws_floats(k, ns)), dtype=torch.float32, device=dpatch.device)
    ws_wp = None
    if dWp is not None:
        assert dWp.is_contiguous() and tuple(dWp.shape) == (64, 64, k, k, k)
        ws_wp = torch.empty(int(_lib.lib().vxb_patch_wgrad_weight_ws_floats(k, ns)), dtype=torch.float32, device=dpatch.device)
    _lib.set_meta('patch_dgrad_input_wgrad[k%d V%d]' % (k, V), 0.0)
    sc = absmax_scale(dpatch)
    _lib.set_meta('patch_dgrad_input_wgrad[k%d V%d]' % (k, V), 2.0 * B * G ** 3 * k ** 3 * 64 * 64 * (2 if dWp is not None else 1))
    call('vxb_patch_dgrad_input_wgrad_f32', dpatch, wt, d0, vox, B, V, G, k, pad, LRELU_SLOPE, sc, ws, ns, dW, db, dWp, ws_wp)


# --------------------------------------------------------------------------------------------- voxel-sized ops
def pointwise_fwd(x, W, bias):
    nvox = x.numel() // x.shape[-1]
    Cout = W.shape[0]
    y = torch.empty(x.shape[:-1] + (Cout,), dtype=torch.float32, device=x.device)
    call('vxb_pointwise_fwd_f32', x, W, bias, y, nvox, x.shape[-1], Cout, LRELU_SLOPE)
    return y


def pointwise_wgrad(x, y, dy, dW, db):
    nvox = x.numel() // x.shape[-1]
    Cin = x.shape[-1]
    nb = (nvox + 4095) // 4096
    ws = torch.empty(nb * (64 * Cin + 64), dtype=torch.float32, device=x.device)
    call('vxb_pointwise_wgrad_f32', x, y, dy, dW, db, ws, nvox, Cin, y.shape[-1], LRELU_SLOPE)


_LIN = {}


def lin_table(S, device):
    key = (S, str(device))
    if key not in _LIN:
        # network_utils.py:782-792: np.linspace in float64, then .float()
        _LIN[key] = torch.from_numpy(np.linspace(-1., 1., S)).float().to(device)
    return _LIN[key]


def ss3d_max_fwd(x, bs, B, S, C):
    dev = x.device
    want = max(64, (1024 + B - 1) // B)            # same rule as vxb_ss3d_max_fwd_f32: >= 1024 workgroups per launch
    rpc = max(1, S * S // want)
    nchunk = (S * S + rpc - 1) // rpc
    ws = torch.empty(B * nchunk * C * 7, dtype=torch.float32, device=dev)
    out_ss = torch.empty((B, 3 * C), dtype=torch.float32, device=dev)
    out_max = torch.empty((B, C), dtype=torch.float32, device=dev)
    stats = torch.empty((B, C, 2), dtype=torch.float32, device=dev)
    argmax = torch.empty((B, C), dtype=torch.int32, device=dev)
    call('vxb_ss3d_max_fwd_f32', x, bs, B, S, C, lin_table(S, dev), ws, out_ss, out_max, stats, argmax)
    return out_ss, out_max, stats, argmax


FINAL_SS3D = os.environ.get('VOXACTB_FINAL_SS3D', '1') != '0'     # SS3D / max statistics of `final`'s output from the conv's epilogue


def conv3_ss3d_ok(C0, C1, N, S):
    return (FINAL_SS3D and HALO_CONV and HALO_WD and PRECISION == 'bf16x3' and N == 64 and C0 % 32 == 0 and C1 % 32 == 0 and 16 <= S <= 256
            and os.environ.get('VOXACTB_HALO_WN', '2') == '2')


# `final`'s forward with the depth taps by Winograd F(2, 3) (conv_halo_bf16.hip, WG: 2/3 of the MFMAs); needs whole depth pairs (even S)
FINAL_WINOGRAD = os.environ.get('VOXACTB_FINAL_WINOGRAD', '1') != '0'
# ... and the propagating (fp16x2) half of its data gradient; even S
DGRAD_WINOGRAD = os.environ.get('VOXACTB_DGRAD_WINOGRAD', '1') != '0'
# ... and the leaf (single fp16 product) half, in 16-channel chunks
LEAF_WINOGRAD = os.environ.get('VOXACTB_LEAF_WINOGRAD', '0') != '0'


def halo_wfrag_wg(wt_kn, Ct):
    """fp32 weights [(tap = kd*9 + kh*3 + kw, ci)][N = 64] -> the 36 Winograd taps (xi, kh, kw), xi over the depth taps
    {g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2}, as bf16 hi/lo planes in the fragment order of halo_wfrag
    [N/64][chunk of 16][36][column tile 2][plane 2][hi 2][lq 32][8].  Memoised per step like to_bf16_nk."""
    keep = getattr(wt_kn, '_vxb_keep', False)
    key = (wt_kn.data_ptr(), tuple(wt_kn.shape), 'wfrag_wg', PRECISION)
    if keep:
        hit = _WCACHE.get(key)
        if hit is not None:
            return hit[0]
    N = wt_kn.shape[1]
    g = wt_kn.view(3, 9 * Ct, N)
    half = 0.5 * (g[0] + g[2])
    t = torch.stack([g[0], half + 0.5 * g[1], half - 0.5 * g[1], g[2]], 0)          # [4][9 Ct][N]
    wb = split_bf16(t.view(36 * Ct, N).t().contiguous())                             # planes [2][N][36 Ct]
    v = wb.view(2, N // 64, 2, 32, 36, Ct // 16, 2, 8)                                # (p, nb, j, lq, tap, ch, hi, e)
    out = v.permute(1, 5, 4, 2, 0, 6, 3, 7).contiguous()                              # (nb, ch, tap, j, p, hi, lq, e)
    if keep:
        _WCACHE[key] = (out, wt_kn)
    return out


def halo_wfrag_x2_wg(wt_kn, Ct):
    """halo_wfrag_wg for the 'fp16x2' product mode: fp32 weights [(tap, ci)][64] -> the 36 Winograd taps, rounded to fp16 after the
    transform, in the fragment order of halo_wfrag_x2 [1][chunk][36][column tile 2][hi][lq][8]."""
    N = wt_kn.shape[1]
    g = wt_kn.reshape(3, 9 * Ct, N)
    half = 0.5 * (g[0] + g[2])
    t = torch.stack([g[0], half + 0.5 * g[1], half - 0.5 * g[1], g[2]], 0)          # [4][9 Ct][N]
    w16 = t.view(36 * Ct, N).t().contiguous().half()
    v = w16.view(N // 64, 2, 32, 36, Ct // 16, 2, 8)                                # (nb, j, lq, tap, ch, hi, e)
    return v.permute(0, 4, 3, 1, 5, 2, 6).contiguous()                              # (nb, ch, tap, j, hi, lq, e)


def conv3_ss3d_fwd(src0, src1, wt, bias, B, S, act=ACT_LRELU, label=None):
    """conv3d(src0 | src1, 3x3x3, replicate padding, 64 columns) + ss3d_max_fwd of its output, the statistics taken in the conv's
    epilogue (vxb_conv3_halo_ss3d_bf16x3_f32): -> (out [B,S,S,S,64], (out_ss, out_max, stats, argmax)).  With FINAL_WINOGRAD off (or an
    odd grid) `out` is bit-identical to conv3d's; with the depth axis by Winograd's F(2, 3) (the default on even grids since round 5) it
    agrees with it to ~1e-6 of the output maximum (tests/test_halo_winograd_gpu.py), not bit for bit.  The pooled features equal
    ss3d_max_fwd's up to the association of the partial sums."""
    dev = src0.device
    C0 = src0.shape[-1]
    C1 = src1.shape[-1] if src1 is not None else 0
    out = torch.empty((B, S, S, S, 64), dtype=torch.float32, device=dev)
    ws = torch.empty(int(_lib.lib().vxb_conv3_halo_ss3d_ws(B, S)), dtype=torch.float32, device=dev)
    out_ss = torch.empty((B, 3 * 64), dtype=torch.float32, device=dev)
    out_max = torch.empty((B, 64), dtype=torch.float32, device=dev)
    stats = torch.empty((B, 64, 2), dtype=torch.float32, device=dev)
    argmax = torch.empty((B, 64), dtype=torch.int32, device=dev)
    lbl, flops = label or 'conv3d_bf16[k3 s1 %d->64 S%d]' % (C0 + C1, S), 2.0 * B * S ** 3 * 64 * 27 * (C0 + C1)
    if FINAL_WINOGRAD and S % 2 == 0:
        wfw, lin = halo_wfrag_wg(wt, C0 + C1), lin_table(S, dev)        # (before the label: their own launches must not take it)
        _lib.set_meta(lbl, flops)
        call('vxb_conv3_halo_ss3d_wg_bf16x3_f32', src0, src1, C0, C1, B, S, bias, out, act, LRELU_SLOPE, wfw, lin, ws, out_ss, out_max,
             stats, argmax)
        return out, (out_ss, out_max, stats, argmax)
    wb = to_bf16_nk(wt)
    assert wb.dim() == 3
    wf = halo_wfrag(wb, C0 + C1)
    _lib.set_meta(lbl, flops)
    call('vxb_conv3_halo_ss3d_bf16x3_f32', src0, src1, C0, C1, B, S, wb, bias, out, act, LRELU_SLOPE, wf, lin_table(S, dev), ws,
         out_ss, out_max, stats, argmax)
    return out, (out_ss, out_max, stats, argmax)


def pointwise_ss3d_fwd(x, W, bias, B, S):
    """pointwise_fwd + ss3d_max_fwd of its output in one pass (x [B,S,S,S,Cin] -> y [B,S,S,S,64], (out_ss, out_max, stats, argmax))."""
    dev = x.device
    Cout = W.shape[0]
    y = torch.empty(x.shape[:-1] + (Cout,), dtype=torch.float32, device=dev)
    want = max(64, (1024 + B - 1) // B)
    rpc = max(1, S * S // want)
    nchunk = (S * S + rpc - 1) // rpc
    ws = torch.empty(B * nchunk * Cout * 7, dtype=torch.float32, device=dev)
    out_ss = torch.empty((B, 3 * Cout), dtype=torch.float32, device=dev)
    out_max = torch.empty((B, Cout), dtype=torch.float32, device=dev)
    stats = torch.empty((B, Cout, 2), dtype=torch.float32, device=dev)
    argmax = torch.empty((B, Cout), dtype=torch.int32, device=dev)
    call('vxb_pointwise_ss3d_fwd_f32', x, W, bias, y, B, S, x.shape[-1], Cout, LRELU_SLOPE, lin_table(S, dev), ws, out_ss, out_max,
         stats, argmax)
    return y, (out_ss, out_max, stats, argmax)


def pointwise_wgrad_ss3d(x, y, dy, dW, db, B, S, stats, out_ss, argmax, g_ss, g_max, fold_src=None, Sp=0, pad=0):
    """pointwise_wgrad with the ss3d_max_bwd term of y's pooled features added to dy on the fly (dy itself is not modified);
    fold_src [B, Sp^3, 64]: fold_pad(fold_src, Sp, ..., pad) is added to dy on the fly as well."""
    Cin = x.shape[-1]
    nb = B * ((S ** 3 + 4095) // 4096)
    ws = torch.empty(nb * (64 * Cin + 64), dtype=torch.float32, device=x.device)
    call('vxb_pointwise_wgrad_ss3d_f32', x, y, dy, dW, db, ws, B, S, Cin, y.shape[-1], LRELU_SLOPE, lin_table(S, x.device), stats,
         out_ss, argmax, g_ss, g_max, fold_src, Sp, pad)


def ss3d_max_bwd(x, bs, B, S, C, stats, out_ss, argmax, g_ss, g_max, dx, dbs, accumulate=False):
    call('vxb_ss3d_max_bwd_f32', x, bs, B, S, C, lin_table(S, x.device), stats, out_ss, argmax, g_ss, g_max, dx, dbs,
         int(accumulate))
    return dx


C1_MFMA = True       # Cout = 1 conv of the translation head on the matrix cores in the bf16 / bf16x3 precisions


def conv3_c1_fwd(u, w, bias, B, S):
    q = torch.empty((B, S, S, S), dtype=torch.float32, device=u.device)
    if C1_MFMA and _mm() and u.shape[-1] == 64:
        ws = torch.empty(4096, dtype=torch.bfloat16, device=u.device)
        _lib.set_meta('vxb_conv3_c1_fwd_mfma', 0.0)
        call('vxb_conv3_c1_fwd_mfma', u, w, bias, q, B, S, ws)
        return q
    call('vxb_conv3_c1_fwd_f32', u, w, bias, q, B, S, 64)
    return q


def conv3_c1_dgrad(dq, w, u, du, B, S, accumulate=True, mask=True):
    call('vxb_conv3_c1_dgrad_f32', dq, w, u, du, B, S, 64, int(accumulate), int(mask), LRELU_SLOPE)
    return du


def c1_dgrad_ss3d_ok(S, C):
    return C == 64 and S % 4 == 0


def conv3_c1_dgrad_ss3d(dq, w, u, du, B, S, stats, out_ss, argmax, g_ss, g_max, dbias, accumulate=False, want_scale=False):
    """du = lrelu'(u) * ([du] + c1 data gradient + the ss3d_max_bwd term of u); dbias += column sums of du.
    want_scale: also return the fp16 operand scale of du ([scale, 1 / scale] on the device, as absmax_scale(du)), taken while du
    is written -- no extra pass over the 4 GB tensor."""
    ws = torch.empty(int(_lib.lib().vxb_conv3_c1_dgrad_ss3d_ws_floats(B, S)), dtype=torch.float32, device=u.device)
    sc = torch.empty(2, dtype=torch.float32, device=u.device) if want_scale else None
    call('vxb_conv3_c1_dgrad_ss3d_f32', dq, w, u, du, B, S, 64, int(accumulate), LRELU_SLOPE, lin_table(S, u.device), stats, out_ss,
         argmax, g_ss, g_max, dbias, ws, sc)
    return (du, sc) if want_scale else du


C1_WGRAD_F16 = os.environ.get('VOXACTB_C1_WGRAD_F16', '1') != '0'    # trans_decoder weight gradient on single fp16 products ('0': bf16x3)
C1_WGRAD_F16_MIN_VOXELS = 1 << 19


def conv3_c1_wgrad(u, dq, dw, db, B, S):
    if C1_MFMA and _mm() and u.shape[-1] == 64:
        nb = int(_lib.lib().vxb_conv3_c1_wgrad_mfma_blocks(B, S))
        ws = torch.empty(nb * (64 * 27 + 1), dtype=torch.float32, device=u.device)
        # (from 2^19 voxels on -- round 6: below that nothing averages the 2^-12 operand rounding of a 64 x 27 gradient summed over a few 10^4
        # voxels, the one widened element gate of the suite (no_skip_connection at V = 32, 1.04 x) came from here; the bf16x3 kernel runs there)
        if C1_WGRAD_F16 and PRECISION == 'bf16x3' and WGRAD_PRECISION == 'fp16' and dq.is_contiguous() and B * S ** 3 >= C1_WGRAD_F16_MIN_VOXELS:
            # a leaf of the backward pass: one fp16 product per term, dq scaled by a device-side power of two (max |dq| -> [2^14, 2^15))
            _lib.set_meta('vxb_conv3_c1_wgrad_mfma', 0.0)
            sc = absmax_scale(dq)
            _lib.set_meta('vxb_conv3_c1_wgrad_mfma', 0.0)
            call('vxb_conv3_c1_wgrad_f16', u, dq, sc, dw, db, ws, B, S)
            return
        _lib.set_meta('vxb_conv3_c1_wgrad_mfma', 0.0)
        call('vxb_conv3_c1_wgrad_mfma', u, dq, dw, db, ws, B, S)
        return
    nb = (B * S * S + 63) // 64
    ws = torch.empty(nb * (64 * 27 + 1), dtype=torch.float32, device=u.device)
    call('vxb_conv3_c1_wgrad_f32', u, dq, dw, db, ws, B, S, 64)


def ctx_build(lang, patch, pp, pos, B, T0, T1, C):
    """pp [B, Cp]: one proprio embedding (Cp = C) or the right | left pair of the 2Robots encoder (Cp = 2 C)."""
    Cp = pp.shape[1]
    ctx = torch.empty((B, T0 + T1, C + Cp), dtype=torch.float32, device=lang.device)
    call('vxb_ctx_build_f32', lang, patch, pp, pos, ctx, B, T0, T1, C, Cp)
    return ctx


def ctx_bwd(dctx, dpos, B, T0, T1, C, Cp=None):
    dev = dctx.device
    Cp = C if Cp is None else Cp
    dlang = torch.empty((max(B * T0, 1), C + Cp), dtype=torch.float32, device=dev)      # (T0 = 0 -- lang_fusion_type 'concat' -- one unused row)
    dpatch = torch.empty((B * T1, C), dtype=torch.float32, device=dev)
    dpp = torch.empty((B, Cp), dtype=torch.float32, device=dev)
    ws = torch.empty(B * 32 * Cp, dtype=torch.float32, device=dev)
    call('vxb_ctx_bwd_f32', dctx, dlang, dpatch, dpp, dpos, ws, B, T0, T1, C, Cp)
    return dlang, dpatch, dpp


def ce_big(x, label, dx=None, gscale=1.0):
    """x [B, P] logits, label [B] int32 -> (loss [B], lse [B], argmax [B] int32); dx = gscale*(softmax - onehot)."""
    B, P = x.shape
    dev = x.device
    nchunk = (P + 65535) // 65536
    ws = torch.empty(B * nchunk * 4, dtype=torch.float32, device=dev)
    lse = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty(B, dtype=torch.float32, device=dev)
    arg = torch.empty(B, dtype=torch.int32, device=dev)
    call('vxb_ce_big_f32', x, P, B, label, ws, lse, loss, arg, dx, float(gscale))
    return loss, lse, arg


def ce_rows(logits, segs, labels, dlogits=None, gscale=1.0):
    """logits [rows, ld]; segs = [(col0, ncls), ...]; labels [rows, nseg] int32 -> (loss [rows,nseg], pred [rows,nseg])."""
    import ctypes
    rows = logits.shape[0]
    n = len(segs)
    dev = logits.device
    loss = torch.empty((rows, n), dtype=torch.float32, device=dev)
    pred = torch.empty((rows, n), dtype=torch.int32, device=dev)
    c0 = (ctypes.c_int32 * n)(*[s[0] for s in segs])
    nc = (ctypes.c_int32 * n)(*[s[1] for s in segs])
    with _lib.on_device(logits):
        rc = _lib.lib().vxb_ce_rows_f32(_lib.ptr(logits), logits.stride(0), rows, n, c0, nc, _lib.ptr(labels), _lib.ptr(loss),
                                        _lib.ptr(pred), _lib.ptr(dlogits), float(gscale), _lib.stream_ptr(dev))
    _lib.check(rc, 'vxb_ce_rows_f32')
    return loss, pred


# --------------------------------------------------------------------------------------------- bf16 matrix-core mode
GEMM256 = os.environ.get('VOXACTB_GEMM256', '1') != '0'      # 256 x 256-tile kernel for the big linear layers (A/B switch)
DL_GEMM = True       # direct-to-LDS kernels (gemm_dl.hip) for K % 32 == 0 GEMMs / single-source convs in the bf16 modes:
                     # True = where the split pass pays (rules below), 'force' = wherever legal (tests), False = never
_ZEROS = {}


def _zeros16(dev):
    z = _ZEROS.get(dev)
    if z is None:
        z = _ZEROS[dev] = torch.zeros(64, dtype=torch.bfloat16, device=dev)
    return z


def split_planes(x2d, nplanes):
    """fp32 [rows][cols] (row stride free) -> bf16 planes [nplanes][rows][cols]: hi (and lo = bf16(x - hi))."""
    rows, cols = x2d.shape
    planes = torch.empty((nplanes, rows, cols), dtype=torch.bfloat16, device=x2d.device)
    call('vxb_split_bf16_f32', x2d, x2d.stride(0), rows, cols, planes, nplanes)
    return planes


def gemm_bf16w(x, Wb, out=None, bias=None, act=ACT_NONE, residual=None, accumulate=False, label=None, f16_out=None):
    """out[M,N] = act(x[M,K] (fp32 -> bf16 on the fly) @ Wb[N,K]^T (bf16) + bias) (+ residual), fp32 accumulate.  f16_out (contiguous
    half [M][N] view): filled with the fp16 plane of the result where the wide kernel runs (its epilogue); returns out, or (out, filled)
    when f16_out is given."""
    if f16_out is not None:
        M_, K_ = x.shape
        N_ = Wb.shape[-2]
        if (WIDE_GEMM and Wb.dim() == 3 and N_ % 512 == 0 and K_ % 32 == 0 and K_ >= 256 and M_ >= WIDE_MIN_M and x.stride(1) == 1
                and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and not accumulate and f16_out.is_contiguous() and f16_out.data_ptr() % 16 == 0):
            wf = gemm_wfrag(Wb)
            if wf is not None:
                if out is None:
                    out = torch.empty((M_, N_), dtype=torch.float32, device=x.device)
                _lib.set_meta(label or 'gemm_bf16 %dx%dx%d' % (M_, N_, K_), 2.0 * M_ * N_ * K_)
                call('vxb_gemm_wide_bf16x3_f16out_f32', x, x.stride(0), wf, out, out.stride(0), bias, residual, M_, N_, K_, act, LRELU_SLOPE, 0,
                     f16_out)
                return out, True
        return gemm_bf16w(x, Wb, out, bias, act, residual, accumulate, label), False
    M, K = x.shape
    x3 = Wb.dim() == 3          # hi/lo planes of the bf16x3 split
    N = Wb.shape[-2]
    assert Wb.dtype == torch.bfloat16 and Wb.is_contiguous() and Wb.shape[-1] == K
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if (WIDE_GEMM and x3 and N % 512 == 0 and K % 32 == 0 and K >= 256 and M >= WIDE_MIN_M and x.stride(1) == 1 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0):
        wf = gemm_wfrag(Wb)
        if wf is not None:
            # 128 x 512 workgroup tiles, A read once and three k-tiles ahead, weights as fragments from global memory (gemm_wide.hip)
            _lib.set_meta(label or 'gemm_bf16 %dx%dx%d' % (M, N, K), 2.0 * M * N * K)
            call('vxb_gemm_wide_bf16x3_f32', x, x.stride(0), wf, out, out.stride(0), bias, residual, M, N, K, act, LRELU_SLOPE,
                 int(accumulate))
            return out
    # measured at M = 32768 in bf16x3 (tools/bench_gemm.py): N x K = 4096 x 512: 239 TF/s vs 217 (direct-to-LDS 128^2 + weight
    # fragments) / 174 (register-staged); at N <= 2048 the 128^2 kernels win (270 vs 244 at 2048 x 512, 283 vs 257 at 512 x 4096)
    if (GEMM256 and x3 and K % 32 == 0 and N % 256 == 0 and (N >= 4096 or GEMM256 == 'force') and M >= 2048 and x.stride(1) == 1
            and x.stride(0) % 4 == 0):
        # big linear layers: 256 x 256 tiles, 8 waves, both operands as bf16 planes through direct-to-LDS loads (gemm256.hip)
        npl = 2 if x3 else 1
        _lib.set_meta(label or 'gemm_bf16 %dx%dx%d' % (M, N, K), 0.0)
        planes = split_planes(x, npl)
        _lib.set_meta(label or 'gemm_bf16 %dx%dx%d' % (M, N, K), 2.0 * M * N * K)
        call('vxb_gemm256_f32', planes, K, Wb, npl, out, out.stride(0), bias, residual, M, N, K, act, LRELU_SLOPE, int(accumulate))
        return out
    _lib.set_meta(label or 'gemm_bf16 %dx%dx%d' % (M, N, K), 2.0 * M * N * K)
    # the split pass costs M*K*(6|8) bytes of HBM traffic; it pays when every A tile is reused by many column tiles
    # (measured at M = 32768 in bf16x3: N x K = 4096x512 5.3 -> 4.6 ms, 2048x512 2.65 -> 2.40, but 512x4096 3.2 -> 4.4)
    if (DL_GEMM and K % 32 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and M >= 128
            and (DL_GEMM == 'force' or not x3 or N >= 2 * K)):
        # both operands as bf16 planes, global -> LDS without a register round trip (one streaming split pass for x,
        # timed under the same label with zero FLOPs)
        _lib.set_meta(label or 'gemm_bf16 %dx%dx%d' % (M, N, K), 0.0)
        planes = split_planes(x, 2 if x3 else 1)
        _lib.set_meta(label or 'gemm_bf16 %dx%dx%d' % (M, N, K), 2.0 * M * N * K)
        call('vxb_gemm_dl_f32', planes, Wb, gemm_wfrag(Wb), 2 if x3 else 1, out, out.stride(0), bias, residual, M, N, K, act,
             LRELU_SLOPE, int(accumulate))
        return out
    call('vxb_gemm_bf16x3_f32' if x3 else 'vxb_gemm_bf16w_f32', x, x.stride(0), Wb, out, out.stride(0), bias, residual, M, N, K, act, LRELU_SLOPE,
         int(accumulate))
    return out


def conv3d_bf16w(src0, wb, N, B, S_in, S_out, kext, off, stride=1, replicate=True, bias=None, act=ACT_NONE, src1=None,
                 out=None, ldc=None, accumulate=False, d2s=(0, 0), label=None):
    """same contract as conv3d(), weights wb = bf16 [N][(tap, ci)]."""
    C0 = src0.shape[-1]
    C1 = src1.shape[-1] if src1 is not None else 0
    assert wb.dtype == torch.bfloat16 and wb.is_contiguous()
    if out is None:
        if d2s[0] > 0:
            Vf = S_out * d2s[0]
            out = torch.empty((B, Vf, Vf, Vf, d2s[1]), dtype=torch.float32, device=src0.device)
        else:
            out = torch.empty((B, S_out, S_out, S_out, N), dtype=torch.float32, device=src0.device)
    _lib.set_meta(label or 'conv3d_bf16[k%d s%d %d->%d S%d%s]' % (kext, stride, C0 + C1, N, S_out, '' if replicate else ' dgrad'),
                  2.0 * B * S_out ** 3 * N * kext ** 3 * (C0 + C1))
    x3 = wb.dim() == 3
    if (HALO_CONV and kext == 3 and stride == 1 and (d2s[0] == 0 or (d2s[1] == 64 and HALO_D2S)) and not accumulate and N % 64 == 0
            and (ldc is None or ldc == N) and S_out >= 16):
        wf = halo_wfrag(wb, C0 + C1)
        _lib.set_meta(label or 'conv3d_bf16[k%d s%d %d->%d S%d%s]' % (kext, stride, C0 + C1, N, S_out, '' if replicate else ' dgrad'),
                      2.0 * B * S_out ** 3 * N * kext ** 3 * (C0 + C1))
        call('vxb_conv3_halo_bf16x3_f32' if x3 else 'vxb_conv3_halo_bf16w_f32', src0, src1, C0, C1, B, S_in, S_out, off,
             int(replicate), wb, N, bias, out, act, LRELU_SLOPE, 0, 0, d2s[0], wf, None, 0, 0)
        return out
    # ... same trade for convs: every input voxel must feed enough products ((kext/stride)^3 * N per channel) to amortise
    # the split pass (up-conv forward 17.7 -> 15.3 ms; the stride-5 patchify would lose 1.5 ms and stays register-staged)
    if (DL_GEMM and src1 is None and C0 % 32 == 0 and src0.is_contiguous() and B * S_out ** 3 >= 128
            and (DL_GEMM == 'force' or (kext / float(stride)) ** 3 * N >= 512)):
        lbl = label or 'conv3d_bf16[k%d s%d %d->%d S%d%s]' % (kext, stride, C0, N, S_out, '' if replicate else ' dgrad')
        npl = 2 if x3 else 1
        _lib.set_meta(lbl, 0.0)
        planes = split_planes(src0.view(-1, C0), npl)
        _lib.set_meta(lbl, 2.0 * B * S_out ** 3 * N * kext ** 3 * C0)
        call('vxb_conv3d_dl_f32', planes, C0, B, S_in, S_out, stride, kext, off, int(replicate), wb, gemm_wfrag(wb), npl, N, bias, out,
             ldc if ldc is not None else N, act, LRELU_SLOPE, int(accumulate), d2s[0], d2s[1], _zeros16(src0.device), None, None)
        return out
    call('vxb_conv3d_bf16x3_f32' if x3 else 'vxb_conv3d_bf16w_f32', src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, int(replicate), wb, N, bias, out,
         ldc if ldc is not None else N, act, LRELU_SLOPE, int(accumulate), d2s[0], d2s[1])
    return out


def to_bf16_nk(wt_kn):
    """[K][N] fp32 weight layout of the fp32 kernels -> bf16 [N][K] (one small transposing copy per step); in 'bf16x3'
    mode the hi/lo planes [2][N][K]."""
    if getattr(wt_kn, '_vxb_keep', False):
        key = (wt_kn.data_ptr(), tuple(wt_kn.shape), 'nk', PRECISION)
        hit = _WCACHE.get(key)
        if hit is None:
            hit = _WCACHE[key] = (split_bf16(wt_kn.t().contiguous()), wt_kn)
        return hit[0]
    return split_bf16(wt_kn.t().contiguous())
