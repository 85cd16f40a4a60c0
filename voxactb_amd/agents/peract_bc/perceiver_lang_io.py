"""PerceiverVoxelLangEncoder -- drop-in for the reference class of the same name
(reference: peract/agents/peract_bc/perceiver_lang_io.py:135-485) for the configuration the VoxAct-B
single-arm / acting / stabilizing policies use (lang_fusion_type='seq', pos_encoding_with_lang=True,
no ablation flags; launch_utils.py:744-774).

Same constructor arguments, same parameter / buffer names and shapes (so reference checkpoints load with
`load_state_dict`), same `forward(ins, proprio, lang_goal_emb, lang_token_embs, prev_layer_voxel_grid,
bounds, prev_layer_bounds, mask=None)` outputs.  The arithmetic does NOT run through torch.nn: parameters
are plain storage, and forward/backward are explicit sequences of hand-written gfx950 kernels
(voxactb_amd/csrc/*.hip through the C ABI), orchestrated by `PerceiverEngine` below:

  * channels-last activations end to end (the voxelizer already emits [B,V,V,V,10]);
  * every Conv3DBlock is an implicit GEMM on the matrix cores with replicate padding done by address
    clamping; data gradients are zero-padded convs followed by the padding adjoint ("fold");
  * `Upsample(x s, trilinear) -> Conv3d(k)` of Conv3DUpsampleBlock (network_utils.py:245-250) is evaluated in
    polyphase form: a (2R+1)^3 replicate-padded conv on the LOW-res grid with s^3*64 phase channels and a
    depth-to-space store -- 4.6x fewer FLOPs at k=s=5 and the 4.1 GB upsampled tensor is never materialised;
  * no autograd graph: the backward pass is written out, parameter gradients accumulate into views of one
    flat buffer (one RCCL all-reduce, one fused LAMB launch sequence).
"""
import math

import numpy as np
import torch
from torch import nn

from ... import flash, ops
from ..._lib import VoxactbHipError, require_cuda

import os as _os
# the pooled features of d0 (SpatialSoftmax3D + max, perceiver :360) ride on the input conv's forward / weight-gradient kernels
# instead of making their own passes over the grid ('0': the separate kernels, for A/B runs and the equality test)
FUSE_INPUT_SS = _os.environ.get('VOXACTB_FUSE_INPUT_SS', '1') != '0'
# backward of u = final(...): pooled-feature term + translation head's data gradient + LeakyReLU' + bias column sums in one pass
FUSE_U_BWD = _os.environ.get('VOXACTB_FUSE_U_BWD', '1') != '0'

LRELU_SLOPE = 0.02
LANG_FEAT_DIM, LANG_EMB_DIM, LANG_MAX_SEQ_LEN = 1024, 512, 77


# ----------------------------------------------------------------------------------------------------------------------
# parameter holders: same attribute paths as the reference modules, same initialisation, no forward()
# ----------------------------------------------------------------------------------------------------------------------
def _init_like_reference(weight, bias, activation):
    """network_utils.py:140-154 / :263-276."""
    if activation is None:
        nn.init.xavier_uniform_(weight, gain=nn.init.calculate_gain('linear'))
    elif activation == 'tanh':
        nn.init.xavier_uniform_(weight, gain=nn.init.calculate_gain('tanh'))
    elif activation == 'lrelu':
        nn.init.kaiming_uniform_(weight, a=LRELU_SLOPE, nonlinearity='leaky_relu')
    elif activation == 'relu':
        nn.init.kaiming_uniform_(weight, nonlinearity='relu')
    else:
        raise ValueError()
    nn.init.zeros_(bias)


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise VoxactbHipError('parameter holder: the arithmetic runs in PerceiverEngine (HIP kernels), not torch.nn')


class Conv3DBlock(_Holder):
    """storage twin of network_utils.Conv3DBlock (:128-170)."""

    def __init__(self, in_channels, out_channels, kernel_sizes=3, strides=1, norm=None, activation=None):
        super().__init__()
        if norm is not None:
            raise NotImplementedError('Norm not implemented.')
        self.conv3d = nn.Conv3d(in_channels, out_channels, kernel_sizes, strides, padding=kernel_sizes // 2,
                                padding_mode='replicate')
        _init_like_reference(self.conv3d.weight, self.conv3d.bias, activation)
        self.activation = activation
        self.out_channels = out_channels


class Conv3DUpsampleBlock(_Holder):
    """storage twin of network_utils.Conv3DUpsampleBlock (:237-254): conv_up.0, [Upsample = conv_up.1], conv_up.2."""

    def __init__(self, in_channels, out_channels, strides, kernel_sizes=3, norm=None, activation=None):
        super().__init__()
        layer = [Conv3DBlock(in_channels, out_channels, kernel_sizes, 1, norm, activation)]
        if strides > 1:
            layer.append(nn.Upsample(scale_factor=strides, mode='trilinear', align_corners=False))
        layer.append(Conv3DBlock(out_channels, out_channels, kernel_sizes, 1, norm, activation))
        self.conv_up = nn.Sequential(*layer)


class DenseBlock(_Holder):
    def __init__(self, in_features, out_features, norm=None, activation=None):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)
        _init_like_reference(self.linear.weight, self.linear.bias, activation)
        self.activation = activation


class SpatialSoftmax3D(_Holder):
    """only the registered buffers (state_dict compatibility, network_utils.py:780-795)."""

    def __init__(self, depth, height, width, channel):
        super().__init__()
        self.temperature = 0.01
        px, py, pz = np.meshgrid(np.linspace(-1., 1., depth), np.linspace(-1., 1., height), np.linspace(-1., 1., width))
        for n, a in (('pos_x', px), ('pos_y', py), ('pos_z', pz)):
            self.register_buffer(n, torch.from_numpy(a.reshape(depth * height * width)).float())


class Attention(_Holder):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_kv = nn.Linear(context_dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, query_dim)
        self.dropout_p = dropout


class GEGLU(_Holder):
    pass


class FeedForward(_Holder):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), GEGLU(), nn.Linear(dim * mult, dim))


class PreNorm(_Holder):
    def __init__(self, dim, fn, context_dim=None):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)
        self.norm_context = nn.LayerNorm(context_dim) if context_dim is not None else None


class PerceiverVoxelLangEncoder(nn.Module):

    def __init__(self, depth, iterations, voxel_size, initial_dim, low_dim_size, layer=0, num_rotation_classes=72,
                 num_grip_classes=2, num_collision_classes=2, input_axis=3, num_latents=512, im_channels=64,
                 latent_dim=512, cross_heads=1, latent_heads=8, cross_dim_head=64, latent_dim_head=64, activation='relu',
                 weight_tie_layers=False, pos_encoding_with_lang=True, input_dropout=0.1, attn_dropout=0.1,
                 decoder_dropout=0.0, lang_fusion_type='seq', voxel_patch_size=9, voxel_patch_stride=8,
                 no_skip_connection=False, no_perceiver=False, no_language=False, final_dim=64, arm_pred_loss=False,
                 _two_robots=False):
        super().__init__()
        if lang_fusion_type not in ('seq', 'concat') or (lang_fusion_type == 'concat' and (pos_encoding_with_lang or _two_robots)) \
                or iterations < 1 or activation != 'lrelu' or low_dim_size <= 0 \
                or num_rotation_classes <= 0:
            raise NotImplementedError(
                'voxactb_amd covers the configuration VoxAct-B trains (launch_utils.py:744-774, PERACT_BC.yaml): '
                "lang_fusion_type='seq' (or 'concat' with pos_encoding_with_lang=False, the only combination the reference's forward accepts: perceiver :391 vs :422), activation='lrelu' (+ transformer_iterations >= 1 and the no_language / no_skip_connection / no_perceiver ablations)")
        if im_channels != 64 or final_dim != 64 or int(initial_dim) > 16:
            raise NotImplementedError('kernels are specialised for im_channels = final_dim = 64, initial_dim <= 16')
        if voxel_size % voxel_patch_stride or voxel_patch_size % 2 == 0:
            raise ValueError('voxel_patch_size must be odd and voxel_patch_stride must divide voxel_size '
                             '(the reference fails at perceiver_lang_io.py:422 otherwise)')
        self.depth, self.layer, self.init_dim, self.iterations = depth, layer, int(initial_dim), iterations
        self.input_axis, self.voxel_size, self.low_dim_size, self.im_channels = input_axis, voxel_size, low_dim_size, im_channels
        self.pos_encoding_with_lang, self.lang_fusion_type = pos_encoding_with_lang, lang_fusion_type
        self.voxel_patch_size, self.voxel_patch_stride = voxel_patch_size, voxel_patch_stride
        self.num_rotation_classes, self.num_grip_classes = num_rotation_classes, num_grip_classes
        self.num_collision_classes, self.final_dim = num_collision_classes, final_dim
        self.input_dropout, self.attn_dropout, self.decoder_dropout = input_dropout, attn_dropout, decoder_dropout
        self.no_skip_connection, self.no_perceiver, self.no_language = no_skip_connection, no_perceiver, no_language
        self.arm_pred_loss = arm_pred_loss
        self.two_robots = bool(_two_robots)       # one_policy_more_heads baseline (reference class PerceiverVoxelLang2RobotsEncoder)
        self.cross_heads, self.latent_heads = cross_heads, latent_heads
        self.cross_dim_head, self.latent_dim_head = cross_dim_head, latent_dim_head
        self.num_latents, self.latent_dim = num_latents, latent_dim

        spatial_size = voxel_size // voxel_patch_stride
        # context width: patch features + one proprio embedding, or + the right and the left arm's (perceiver :547, :721-727)
        self.input_dim_before_seq = im_channels * (3 if (self.two_robots or lang_fusion_type == 'concat') else 2)     # (:201)
        if pos_encoding_with_lang:
            self.pos_encoding = nn.Parameter(torch.randn(1, LANG_MAX_SEQ_LEN + spatial_size ** 3, self.input_dim_before_seq))
        else:                                     # (perceiver :212-216: the grid tokens only; the language tokens get none)
            self.pos_encoding = nn.Parameter(torch.randn(1, spatial_size, spatial_size, spatial_size, self.input_dim_before_seq))
        self.input_preprocess = Conv3DBlock(self.init_dim, im_channels, kernel_sizes=1, strides=1, activation=activation)
        self.patchify = Conv3DBlock(im_channels, im_channels, kernel_sizes=voxel_patch_size, strides=voxel_patch_stride,
                                    activation=activation)
        if lang_fusion_type == 'concat':          # the sentence embedding, tiled over the grid as 64 more channels (perceiver :228-229, :379-384)
            self.lang_preprocess = nn.Linear(LANG_FEAT_DIM, im_channels)
        else:
            self.lang_preprocess = nn.Linear(LANG_EMB_DIM, self.input_dim_before_seq)
        self.proprio_preprocess = DenseBlock(low_dim_size, im_channels, None, activation)
        self.ss0 = SpatialSoftmax3D(voxel_size, voxel_size, voxel_size, im_channels)
        flat_size = im_channels * 4
        self.latents = nn.Parameter(torch.randn(num_latents, latent_dim))
        self.cross_attend_blocks = nn.ModuleList([
            PreNorm(latent_dim, Attention(latent_dim, self.input_dim_before_seq, heads=cross_heads, dim_head=cross_dim_head,
                                          dropout=input_dropout), context_dim=self.input_dim_before_seq),
            PreNorm(latent_dim, FeedForward(latent_dim))])
        self.layers = nn.ModuleList([])
        self.weight_tie_layers = bool(weight_tie_layers)
        tied = None
        for _ in range(depth):
            if tied is None or not weight_tie_layers:      # weight_tie_layers (perceiver :262-276, cache_fn): ONE attention / feed-forward pair,
                tied = (PreNorm(latent_dim, Attention(latent_dim, heads=latent_heads, dim_head=latent_dim_head, dropout=attn_dropout)),
                        PreNorm(latent_dim, FeedForward(latent_dim)))             # listed `depth` times (state_dict keys for every layer, parameters once)
            self.layers.append(nn.ModuleList([tied[0], tied[1]]))
        self.decoder_cross_attn = PreNorm(self.input_dim_before_seq,
                                          Attention(self.input_dim_before_seq, latent_dim, heads=cross_heads,
                                                    dim_head=cross_dim_head, dropout=decoder_dropout),
                                          context_dim=latent_dim)
        self.up0 = Conv3DUpsampleBlock(self.input_dim_before_seq, final_dim, kernel_sizes=voxel_patch_size,
                                       strides=voxel_patch_stride, activation=activation)
        self.ss1 = SpatialSoftmax3D(spatial_size, spatial_size, spatial_size, self.input_dim_before_seq)
        flat_size += self.input_dim_before_seq * 4
        self.final = Conv3DBlock(im_channels if (no_perceiver or no_skip_connection) else im_channels * 2, im_channels, kernel_sizes=3,
                                 strides=1, activation=activation)          # (perceiver :302-306: one source under either ablation)
        self.trans_decoder = Conv3DBlock(final_dim, 1, kernel_sizes=3, strides=1, activation=None)
        self.ss_final = SpatialSoftmax3D(voxel_size, voxel_size, voxel_size, im_channels)
        flat_size += im_channels * 4
        self.dense0 = DenseBlock(flat_size, 256, None, activation)
        self.dense1 = DenseBlock(256, final_dim, None, activation)
        self.rot_grip_collision_ff = DenseBlock(final_dim, num_rotation_classes * 3 + num_grip_classes + num_collision_classes,
                                                None, None)
        if arm_pred_loss:
            self.dense2 = DenseBlock(flat_size, final_dim, None, activation)
            self.arm_ff = DenseBlock(final_dim, 2, None, None)
        if self.two_robots:
            # second head set on the same trunk (perceiver :668-690, same registration order)
            self.trans_decoder_left_arm = Conv3DBlock(final_dim, 1, kernel_sizes=3, strides=1, activation=None)
            self.ss_final_left_arm = SpatialSoftmax3D(voxel_size, voxel_size, voxel_size, im_channels)
            self.dense0_left_arm = DenseBlock(flat_size, 256, None, activation)
            self.dense1_left_arm = DenseBlock(256, final_dim, None, activation)
            self.rot_grip_collision_ff_left_arm = DenseBlock(
                final_dim, num_rotation_classes * 3 + num_grip_classes + num_collision_classes, None, None)
        self._engine = None

    # ------------------------------------------------------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            self._engine = PerceiverEngine(self)
        return self._engine

    def forward(self, ins, proprio, lang_goal_emb, lang_token_embs, prev_layer_voxel_grid, bounds, prev_layer_bounds,
                mask=None):
        """ins: [B,10,V,V,V] (the channels-first VIEW QFunction passes, agent :100) or channels-last [B,V,V,V,10].
        `lang_goal_emb`, `prev_layer_*`, `bounds` are dead inputs for lang_fusion_type='seq' (perceiver :345-354)."""
        if mask is not None:
            raise NotImplementedError('attention mask is never passed by the agent')
        eng = self.engine()
        outs, _ = eng.forward(eng.to_channels_last(ins), proprio, lang_token_embs, training=False, save=False, lang_goal_emb=lang_goal_emb)
        return outs


class PerceiverVoxelLang2RobotsEncoder(PerceiverVoxelLangEncoder):
    """The `one_policy_more_heads` baseline (reference perceiver_lang_io.py:488-860): one trunk for both arms -- context
    tokens carry the patch features and BOTH arms' proprio embeddings (192 wide; `proprio_preprocess` is shared), and the
    translation / rotation-gripper-collision heads exist twice (`*_left_arm`).  Same parameter names as the reference."""

    def __init__(self, depth, iterations, voxel_size, initial_dim, low_dim_size, layer=0, num_rotation_classes=72,
                 num_grip_classes=2, num_collision_classes=2, input_axis=3, num_latents=512, im_channels=64,
                 latent_dim=512, cross_heads=1, latent_heads=8, cross_dim_head=64, latent_dim_head=64, activation='relu',
                 weight_tie_layers=False, pos_encoding_with_lang=True, input_dropout=0.1, attn_dropout=0.1,
                 decoder_dropout=0.0, lang_fusion_type='seq', voxel_patch_size=9, voxel_patch_stride=8,
                 no_skip_connection=False, no_perceiver=False, no_language=False, final_dim=64):
        super().__init__(depth, iterations, voxel_size, initial_dim, low_dim_size, layer, num_rotation_classes,
                         num_grip_classes, num_collision_classes, input_axis, num_latents, im_channels, latent_dim,
                         cross_heads, latent_heads, cross_dim_head, latent_dim_head, activation, weight_tie_layers,
                         pos_encoding_with_lang, input_dropout, attn_dropout, decoder_dropout, lang_fusion_type,
                         voxel_patch_size, voxel_patch_stride, no_skip_connection, no_perceiver, no_language, final_dim,
                         arm_pred_loss=False, _two_robots=True)

    def forward(self, ins, proprio_right, proprio_left, lang_goal_emb, lang_token_embs, prev_layer_voxel_grid, bounds,
                prev_layer_bounds, mask=None):
        """-> (trans_right, rot_and_grip_right, collision_right, trans_left, rot_and_grip_left, collision_left) (:860)."""
        if mask is not None:
            raise NotImplementedError('attention mask is never passed by the agent')
        eng = self.engine()
        outs, _ = eng.forward(eng.to_channels_last(ins), proprio_right, lang_token_embs, training=False, save=False,
                              proprio_left=proprio_left)
        return outs


# ----------------------------------------------------------------------------------------------------------------------
def _r4(n):
    return (n + 3) & ~3


KV_F16_FROM_EPILOGUE = _os.environ.get('VOXACTB_KV_F16_EPILOGUE', '1') != '0'     # to_kv's GEMM writes the attention kernels' fp16 k | v plane itself (round 6)
ATTN_F16_MIN_SCORES = 1 << 22      # attn_kernel 'auto': score elements (B * H * Nq * Nk) from which the single-fp16 attention forward runs


def _mix32(a, b):
    """two-round 32-bit finaliser of (a, b): per-layer / per-rank dropout seeds that share no low-bit structure."""
    x = (int(a) * 0x9E3779B1 + int(b) * 0x85EBCA77 + 0xC2B2AE3D) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


class PerceiverEngine:
    """Explicit forward / backward of the encoder on HIP kernels.  Parameter storage = the module's Parameters."""

    def __init__(self, module: PerceiverVoxelLangEncoder):
        self.m = module
        self.P = dict(module.named_parameters())
        k, s = module.voxel_patch_size, module.voxel_patch_stride
        self.k, self.s = k, s
        self.V = module.voxel_size
        self.G = self.V // s
        self.C = module.im_channels
        self.Cx = module.input_dim_before_seq
        self.two = bool(getattr(module, 'two_robots', False))
        self.D = module.latent_dim
        self.L = module.num_latents
        self.concat = getattr(module, 'lang_fusion_type', 'seq') == 'concat'
        self._tied = bool(getattr(module, 'weight_tie_layers', False))
        self.T0 = 0 if self.concat else LANG_MAX_SEQ_LEN          # 'concat': no language tokens in the sequence
        if s > 1:
            Lt, self.R = ops.polyphase_tables(k, s)
            self.kl = 2 * self.R + 1
            self._Lt_host = Lt
        self._Lt = None
        self._on_bucket = None
        self._lin_weights = None          # names of the linear-layer weights whose bf16 planes are made in one launch per step
        self.step_seed = 0
        # Precision of the matrix-core kernels (tensors, accumulators, softmax / norm statistics are fp32 in every mode):
        #   'bf16x3' (default): every fp32 operand is used as hi + lo bf16 halves, three bf16 MFMAs per product
        #             (hi*hi + hi*lo + lo*hi, <= 2^-16 relative per product) -- convs, linears and the fused attention;
        #             held to the same 1e-4 Q-value bound as 'fp32' (tests/test_encoder_gpu.py)
        #   'fp32'  : exact fp32 matrix cores (v_mfma_f32_32x32x2_f32) everywhere, unfused attention
        #   'bf16'  : plain bf16 operands (throughput mode, ~4e-3 on Q-values)
        import os
        self.precision = os.environ.get('VOXACTB_PRECISION', 'bf16x3')
        if self.precision not in ('fp32', 'bf16', 'bf16x3'):
            raise ValueError('VOXACTB_PRECISION must be fp32, bf16x3 or bf16')
        self.fused_attention = os.environ.get('VOXACTB_FUSED_ATTENTION', '1') != '0'   # bf16 / bf16x3 modes, head dim 64
        # experiment switch: 'bf16' runs only the attention core (QK^T, PV and their gradients) on plain bf16 operands while
        # every other product stays bf16x3 -- measured against the reference digest in DESIGN.md section 6 (it does NOT hold
        # the 1e-4 bound, so it is not a shipped mode)
        self.attn_precision = os.environ.get('VOXACTB_ATTN_PRECISION', '')
        # precision of the matrix products of the BACKWARD pass ('' = same as the forward); see DESIGN.md section 4a
        self.bwd_precision = os.environ.get('VOXACTB_BWD_PRECISION', '')
        self.attn_bwd_precision = os.environ.get('VOXACTB_ATTN_BWD_PRECISION', '')
        # attention core: 'r3' = round 3's kernels (bf16x3 triples, or plain bf16 in the 'bf16' precision); 'f16' / 'bf16' = the pipelined
        # kernels of round 4 (csrc/flash2_*.hip) on single fp16 / bf16 products; attn_bwd_gx: dO and dS as hi + lo pairs in their backward.
        # 'auto' (round 5, a NAMED mode, not the default): the pipelined single-fp16 forward where B * H * Nq * Nk >= 2^22 scores (every
        # attention of configs[1] .. [4] and of the released recipe), round 3's bf16x3 forward below that.  Measured: Q-values
        # stay inside 1e-4 on every reference fixture at those sizes (max 8.1e-5 against 7.0e-5) and the gradients hold the float64 gate on
        # all eight batches once the backward is evaluated at the reference's LeakyReLU choices -- but the element gates of the F5c3 digest
        # (0.3 % of a small tensor's maximum) are missed by 2 x (0.65 %), so the default stays round 3's bf16x3 forward
        self.attn_kernel = os.environ.get('VOXACTB_ATTN_KERNEL', 'auto')
        if self.attn_kernel not in ('auto', 'r3', 'r3bf16', 'f16', 'bf16', 'bf16x3'):
            raise ValueError('VOXACTB_ATTN_KERNEL must be auto, r3, r3bf16, f16, bf16 or bf16x3')
        self.attn_bwd_gx = os.environ.get('VOXACTB_ATTN_BWD_GX', '0') != '0'
        # backward of the attention core when the forward ran round 3's kernels: '' = round 3's backward too, 'f16' / 'bf16' = the
        # pipelined backward (it only needs q, k | v, O, lse and the dropout seed of the forward)
        self.attn_bwd_kernel = os.environ.get('VOXACTB_ATTN_BWD_KERNEL', 'f16')
        if self.attn_bwd_kernel not in ('', 'f16', 'bf16'):
            raise ValueError("VOXACTB_ATTN_BWD_KERNEL must be '' (round 3's backward: VOXACTB_ATTN_BWD_PRECISION then selects bf16x3 / bf16), f16 or bf16")
        if self.attn_bwd_precision and self.attn_bwd_kernel and 'VOXACTB_ATTN_BWD_KERNEL' not in os.environ:
            self.attn_bwd_kernel = ''         # an explicitly chosen precision of round 3's backward is not silently overridden by the new default
        # weight gradients of the two big 3x3x3 convs (`final`, the polyphase up-conv) when the backward runs in 'bf16x3':
        # 'fp16' (default) = one fp16 product per term, the gradient operand scaled by a power of two taken from its largest
        # magnitude on the device; 'bf16x3' = the triple.  Leaves of the backward pass: nothing downstream sees their rounding.
        self.wgrad_precision = os.environ.get('VOXACTB_WGRAD_PRECISION', 'fp16')
        if self.wgrad_precision not in ('fp16', 'bf16x3'):
            raise ValueError('VOXACTB_WGRAD_PRECISION must be fp16 or bf16x3')
        # ... also the weight gradients of the linear layers (>= 1024 rows) and of the two 5^3 convs (generic kernel, delayed scaling)
        self.generic_wgrad_f16 = os.environ.get('VOXACTB_GENERIC_WGRAD_F16', '1') != '0'
        self._grad_scales = {}            # per call site: the delayed fp16 operand scales of THIS engine's gradients (ops._GRAD_SCALE)
        # inference with frozen weights (the evaluation agent's act()): the prepared forms of the weights -- bf16 planes / fragments of every
        # linear layer, conv weight layouts, the polyphase W_eff and its fragments: 0.5 ms of a 6.5 ms act() -- are kept from call to call while
        # the parameters' version counters, their storage and the precision are unchanged (load_state_dict bumps the counters)
        self.freeze_weight_prep = False
        self._prep_sig = None
        self._weff_keep = None
        self._prepared_for_step = False

    # -------------------------------------------------------------------------------------------------- helpers
    def _draw_seed(self):
        """One 32-bit dropout seed per training step, drawn from torch's default CPU generator (so `torch.manual_seed`
        governs it and a resumed run does not replay the first steps' masks) and mixed with the data-parallel rank (the
        reference's nn.Dropout draws independent masks on every rank, perceiver :124-128).  A CPU draw: no device sync."""
        s = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        self.step_seed = _mix32(s, 0x51ED27 + rank)
        return self.step_seed

    def _resolve(self, name):
        prm = self.P.get(name)
        if prm is None and self._tied and name.startswith('layers.'):          # weight_tie_layers: every layer is layer 0's parameters
            prm = self.P['layers.0.' + name.split('.', 2)[2]]
        if prm is None:
            raise KeyError(name)
        return prm

    def p(self, name):
        return self._resolve(name).data

    def g(self, name):
        prm = self._resolve(name)
        if prm.grad is None:
            prm.grad = torch.zeros_like(prm.data)
        return prm.grad

    def Lt(self, dev):
        if self._Lt is None or self._Lt.device != dev:
            self._Lt = torch.from_numpy(self._Lt_host).to(dev)
        return self._Lt

    def to_channels_last(self, ins):
        require_cuda(ins)
        V = self.V
        if ins.dim() != 5:
            raise VoxactbHipError('voxel grid must be 5-D')
        if ins.shape[-1] == self.m.init_dim and ins.shape[1] == V:
            return ins.contiguous()
        x = ins.permute(0, 2, 3, 4, 1)        # channels-first view of an NDHWC buffer -> zero-copy
        return x.contiguous()

    # -------------------------------------------------------------------------------------------------- attention
    def _attn_fwd(self, pre, xq, ctxn, H, d, p, seed, residual, save):
        """xq [B,Nq,Dq], ctxn [B,Nk,Dc] (already normalised).  Returns out [B*Nq, Dout], cache."""
        B, Nq, Dq = xq.shape
        Nk = ctxn.shape[1]
        Wq, Wkv = self.p(pre + '.fn.to_q.weight'), self.p(pre + '.fn.to_kv.weight')
        Wo, bo = self.p(pre + '.fn.to_out.weight'), self.p(pre + '.fn.to_out.bias')
        inner = H * d
        q = ops.linear(xq.view(B * Nq, Dq), Wq)
        # (the plain-bf16 throughput mode takes the pipelined forward by default: single bf16 products either way, VOXACTB_ATTN_KERNEL=r3bf16 keeps round 3's)
        kern = self.attn_kernel
        if kern == 'auto':
            kern = 'f16' if B * H * Nq * Nk >= ATTN_F16_MIN_SCORES else 'r3'
        if (self.precision in ('bf16', 'bf16x3') and d == 64 and self.fused_attention
                and (kern not in ('r3', 'r3bf16') or (self.precision == 'bf16' and kern == 'r3'))):
            mode = 'bf16' if self.precision == 'bf16' else kern
            kvp = None
            if mode == 'f16' and KV_F16_FROM_EPILOGUE:
                # the fp16 k | v plane of the attention kernels out of to_kv's epilogue where the wide GEMM runs (else: a pass over kv, flash.py)
                kvp = torch.empty((1, B * Nk, 2 * inner), dtype=torch.float16, device=xq.device)
                kv, filled = ops.linear(ctxn.reshape(B * Nk, ctxn.shape[2]), Wkv, f16_out=kvp[0])
                if not filled:
                    kvp = None
            else:
                kv = ops.linear(ctxn.reshape(B * Nk, ctxn.shape[2]), Wkv)
            # (save: the forward also stores the dropout keep words for the backward -- same mask, no second hash of it; flash.py)
            O, lse, kvp, dmask = flash.flash2_attn_fwd(q, kv, B, H, Nq, Nk, d ** -0.5, p, seed, mode=mode, planes=kvp, return_planes=True,
                                                       return_mask=True, store_mask=bool(save))
            out = ops.linear(O, Wo, bo, residual=residual)
            cache = dict(q=q, kv=kv, kvp=kvp, O=O, lse=lse, flash=2, mode=mode, dims=(B, Nq, Nk, H, d, 0), p=p, seed=seed, dmask=dmask) if save else None
            return out, cache
        kv = ops.linear(ctxn.reshape(B * Nk, ctxn.shape[2]), Wkv)
        if self.precision in ('bf16', 'bf16x3') and d == 64 and self.fused_attention:
            # fused attention on the bf16 matrix cores, no [B*h, i, j] tensor (csrc/flash_attn.hip); 'bf16x3' carries
            # q, k, v, dO, P and dS as hi + lo halves
            x3 = self.precision == 'bf16x3' and self.attn_precision != 'bf16'
            O, lse, kvp = flash.flash_attn_fwd_dl(q, kv, B, H, Nq, Nk, d ** -0.5, p, seed, x3=x3, return_planes=True)
            out = ops.linear(O, Wo, bo, residual=residual)
            cache = dict(q=q, kv=kv, kvp=kvp, O=O, lse=lse, flash=True, x3=x3, dims=(B, Nq, Nk, H, d, 0), p=p, seed=seed) if save else None
            return out, cache
        ld = _r4(Nk)
        S = torch.empty((B * H, Nq, ld), dtype=torch.float32, device=xq.device)
        ops.gemm(q, kv, S, Nq, Nk, d, inner, 1, 1, 2 * inner, ld, batch=B * H, H=H, bA=(Nq * inner, d),
                 bB=(Nk * 2 * inner, d), bC=(H * Nq * ld, Nq * ld), alpha=d ** -0.5, label='attn_core')
        Pd = ops.softmax_rows(S, B * H * Nq, Nk, ld, p, seed)
        O = torch.empty((B * Nq, inner), dtype=torch.float32, device=xq.device)
        ops.gemm(Pd, kv[:, inner:], O, Nq, d, Nk, ld, 1, 2 * inner, 1, inner, batch=B * H, H=H,
                 bA=(H * Nq * ld, Nq * ld), bB=(Nk * 2 * inner, d), bC=(Nq * inner, d), label='attn_core')
        out = ops.linear(O, Wo, bo, residual=residual)
        cache = dict(q=q, kv=kv, P=S, Pd=Pd, O=O, dims=(B, Nq, Nk, H, d, ld), p=p, seed=seed) if save else None
        return out, cache

    def _attn_bwd(self, pre, c, dout, xq2d, ctx2d, same_src):
        """dout [B*Nq, Dout] -> (dxq [B*Nq,Dq], dctx [B*Nk,Dc] or None if same_src (added into dxq))."""
        B, Nq, Nk, H, d, ld = c['dims']
        inner = H * d
        Wq, Wkv, Wo = self.p(pre + '.fn.to_q.weight'), self.p(pre + '.fn.to_kv.weight'), self.p(pre + '.fn.to_out.weight')
        dev = dout.device
        dO = torch.empty((B * Nq, inner), dtype=torch.float32, device=dev)
        ops.linear_bwd(c['O'], Wo, dout, self.g(pre + '.fn.to_out.weight'), self.g(pre + '.fn.to_out.bias'), dO)
        if c.get('flash') == 2:
            mode, planes = c['mode'], c['kvp']
            if mode == 'bf16x3':                    # (the three-product pipelined forward has no backward twin: the fp16 one runs behind it)
                mode = self.attn_bwd_kernel if self.attn_bwd_kernel in ('f16', 'bf16') else 'f16'
                planes = flash.kv_planes(c['kv'], mode)
            dq, dkv = flash.flash2_attn_bwd(c['q'], c['kv'], c['O'], dO, c['lse'], planes, B, H, Nq, Nk, d ** -0.5, c['p'], c['seed'],
                                            mode=mode, gx=self.attn_bwd_gx or B * H * Nq * Nk < (1 << 22), drop_mask=c.get('dmask'))
            return self._attn_bwd_proj(pre, dq, dkv, xq2d, ctx2d, same_src)
        if c.get('flash') and self.attn_bwd_kernel in ('f16', 'bf16') and (self.bwd_precision or self.precision) != 'fp32':
            mode = 'bf16' if self.precision == 'bf16' else self.attn_bwd_kernel
            planes = flash.kv_planes(c['kv'], mode)
            # gradient operands as hi + lo pairs where asked for, and always on small problems (below ~4 M scores the second product is
            # free and the rounding of single operands is averaged over too few terms: the V = 8 .. 32 fixtures' element gates)
            gx = self.attn_bwd_gx or B * H * Nq * Nk < (1 << 22)
            dq, dkv = flash.flash2_attn_bwd(c['q'], c['kv'], c['O'], dO, c['lse'], planes, B, H, Nq, Nk, d ** -0.5, c['p'], c['seed'],
                                            mode=mode, gx=gx)
            return self._attn_bwd_proj(pre, dq, dkv, xq2d, ctx2d, same_src)
        if c.get('flash'):
            bp = self.attn_bwd_precision or self.bwd_precision or self.precision
            dq, dkv = flash.flash_attn_bwd_dl(c['q'], c['kv'], c['O'], dO, c['lse'], B, H, Nq, Nk, d ** -0.5, c['p'], c['seed'],
                                              x3=c['x3'] and bp == 'bf16x3', kv_planes=c['kvp'])
            return self._attn_bwd_proj(pre, dq, dkv, xq2d, ctx2d, same_src)
        kv, q, P, Pd = c['kv'], c['q'], c['P'], c['Pd']
        dkv = torch.empty_like(kv)
        # dV[j,:] = sum_i Pd[i,j] dO[i,:]
        ops.gemm(Pd, dO, dkv[:, inner:], Nk, d, Nq, 1, ld, inner, 1, 2 * inner, batch=B * H, H=H,
                 bA=(H * Nq * ld, Nq * ld), bB=(Nq * inner, d), bC=(Nk * 2 * inner, d), label='attn_core')
        # dPd[i,j] = sum_d dO[i,d] V[j,d]
        dP = Pd if c['p'] > 0 else torch.empty_like(P)
        ops.gemm(dO, kv[:, inner:], dP, Nq, Nk, d, inner, 1, 1, 2 * inner, ld, batch=B * H, H=H, bA=(Nq * inner, d),
                 bB=(Nk * 2 * inner, d), bC=(H * Nq * ld, Nq * ld), label='attn_core')
        dS = ops.softmax_bwd_rows(P, dP, B * H * Nq, Nk, ld, d ** -0.5, c['p'], c['seed'])
        dq = torch.empty_like(q)
        ops.gemm(dS, kv, dq, Nq, d, Nk, ld, 1, 2 * inner, 1, inner, batch=B * H, H=H, bA=(H * Nq * ld, Nq * ld),
                 bB=(Nk * 2 * inner, d), bC=(Nq * inner, d), label='attn_core')
        ops.gemm(dS, q, dkv, Nk, d, Nq, 1, ld, inner, 1, 2 * inner, batch=B * H, H=H, bA=(H * Nq * ld, Nq * ld),
                 bB=(Nq * inner, d), bC=(Nk * 2 * inner, d), label='attn_core')
        return self._attn_bwd_proj(pre, dq, dkv, xq2d, ctx2d, same_src)

    def _attn_bwd_proj(self, pre, dq, dkv, xq2d, ctx2d, same_src):
        Wq, Wkv = self.p(pre + '.fn.to_q.weight'), self.p(pre + '.fn.to_kv.weight')
        dxq = torch.empty_like(xq2d)
        ops.linear_bwd(xq2d, Wq, dq, self.g(pre + '.fn.to_q.weight'), None, dxq)
        if same_src:
            ops.linear_bwd(ctx2d, Wkv, dkv, self.g(pre + '.fn.to_kv.weight'), None, dxq, dx_accumulate=True)
            return dxq, None
        dctx = torch.empty_like(ctx2d)
        ops.linear_bwd(ctx2d, Wkv, dkv, self.g(pre + '.fn.to_kv.weight'), None, dctx)
        return dxq, dctx

    def _ff_fwd(self, pre, x2d, save):
        xn, mean, rstd = ops.layernorm_fwd(x2d, self.p(pre + '.norm.weight'), self.p(pre + '.norm.bias'))
        h, gg = ops.linear_geglu(xn, self.p(pre + '.fn.net.0.weight'), self.p(pre + '.fn.net.0.bias'))
        out = ops.linear(gg, self.p(pre + '.fn.net.2.weight'), self.p(pre + '.fn.net.2.bias'), residual=x2d)
        return out, (dict(x=x2d, mean=mean, rstd=rstd, xn=xn, h=h, gg=gg) if save else None)

    def _ff_bwd(self, pre, c, dx):
        """dx: gradient wrt the block output (also the residual path); updated in place to the input gradient."""
        W2 = self.p(pre + '.fn.net.2.weight')
        # the weight gradient first (its launch reports the operand scale of dx that the two-product data gradient takes), then
        # d(gg) = dx @ W2 and GEGLU's backward in one launch -- where that applies (ops.geglu_bwd_fusable)
        dh = None
        if ops.geglu_bwd_fusable(dx, W2, c['h']):
            ops._LAST_LIN_DY_SCALE[0] = None
            ops.linear_bwd(c['gg'], W2, dx, self.g(pre + '.fn.net.2.weight'), self.g(pre + '.fn.net.2.bias'), None)
            dh = ops.linear_dgrad_geglu_bwd(dx, W2, c['h'], ops._LAST_LIN_DY_SCALE[0])
            if dh is None:                      # (no fused form after all: the plain data gradient, then GEGLU's backward)
                dgg = torch.empty_like(c['gg'])
                ops.linear_dgrad(dx, W2, dgg, False, ops._LAST_LIN_DY_SCALE[0])
                dh = ops.geglu_bwd(c['h'], dgg)
        else:
            dgg = torch.empty_like(c['gg'])
            ops.linear_bwd(c['gg'], W2, dx, self.g(pre + '.fn.net.2.weight'), self.g(pre + '.fn.net.2.bias'), dgg)
            dh = ops.geglu_bwd(c['h'], dgg)
        dxn = torch.empty_like(c['xn'])
        ops.linear_bwd(c['xn'], self.p(pre + '.fn.net.0.weight'), dh, self.g(pre + '.fn.net.0.weight'),
                       self.g(pre + '.fn.net.0.bias'), dxn)
        ops.layernorm_bwd(dxn, c['x'], self.p(pre + '.norm.weight'), c['mean'], c['rstd'], self.g(pre + '.norm.weight'),
                          self.g(pre + '.norm.bias'), dx=dx, accumulate_dx=True)
        return dx

    def prepare_step(self):
        """The per-step forms of the linear layers' weights (bf16 planes / MFMA fragments: one 0.28 ms launch), made BEFORE the step's first
        data-dependent kernel.  update() calls this ahead of the SE(3) relabel and the voxelizer's chain of dependent launches: the step
        starts on an empty queue (the runner's `.item()` on the previous loss), where every launch waits for the host to enqueue it;
        behind 0.28 ms of weight preparation the host is ahead when the chain runs.  (Round 6.  It does NOT change what bench.py's event
        pair around the voxelizer call reads -- 0.150 - 0.158 ms before and after: that figure is the chain's kernels running on cold
        caches, `vt_tiles` 74 us in the step against 67 stand-alone, plus its launch boundaries)
        forward(training=True, save=True) of the same step skips its own preparation."""
        ops.new_step()
        self._weff_keep = None
        ops.PRECISION = self.precision
        try:
            if self._lin_weights is None:
                self._lin_weights = [n for n, prm in self.P.items() if prm.dim() == 2 and n.endswith('.weight') and prm.numel() >= 4096]
            ops.prepare_linear_weights([self.p(n) for n in self._lin_weights],
                                       geglu=[self.p(n) for n in self._lin_weights if n.endswith('.fn.net.0.weight')],
                                       f16_dgrad=self.precision == 'bf16x3' and self.wgrad_precision == 'fp16')
            self._prep_sig = None
            ops.CACHE_OWNER = None
        finally:
            ops.PRECISION = 'fp32'
        self._prepared_for_step = True

    # -------------------------------------------------------------------------------------------------- forward
    def forward(self, vox, proprio, lang_token_embs, training=False, save=True, seed=None, proprio_left=None, lang_goal_emb=None):
        """vox [B,V,V,V,10] channels-last.  Returns ((trans [B,1,V,V,V], rot_and_grip, collision[, arm]), cache); for the
        2Robots encoder `proprio` is the right arm's, `proprio_left` the left arm's, and the outputs are (trans_right,
        rot_and_grip_right, collision_right, trans_left, rot_and_grip_left, collision_left)."""
        require_cuda(vox, proprio, lang_token_embs)
        if self.two != (proprio_left is not None):
            raise VoxactbHipError('proprio_left is given exactly for the 2Robots encoder')
        sig = None
        if self.freeze_weight_prep and not training and not save:
            first = next(iter(self.P.values()))
            sig = (self.precision, sum(prm._version for prm in self.P.values()), first.data_ptr(), len(self.P))
        reuse = sig is not None and sig == self._prep_sig and ops.CACHE_OWNER is self
        if self._prepared_for_step and training and save and ops.CACHE_OWNER is None:
            reuse = True                  # prepare_step() ran for this very step (update() calls it before the voxelizer)
        self._prepared_for_step = False
        if not reuse:
            ops.new_step()
            self._weff_keep = None
        ops.PRECISION = self.precision
        try:
            if not reuse:
                if self._lin_weights is None:
                    self._lin_weights = [n for n, prm in self.P.items() if prm.dim() == 2 and n.endswith('.weight') and prm.numel() >= 4096]
                ops.prepare_linear_weights([self.p(n) for n in self._lin_weights],
                                           geglu=[self.p(n) for n in self._lin_weights if n.endswith('.fn.net.0.weight')],
                                           f16_dgrad=save and self.precision == 'bf16x3' and self.wgrad_precision == 'fp16')
                self._prep_sig = sig
                ops.CACHE_OWNER = self if sig is not None else None
            self._frozen_now = sig is not None
            if self.concat:
                if lang_goal_emb is None:
                    raise VoxactbHipError("lang_fusion_type='concat' reads lang_goal_emb")
                lang_token_embs = lang_goal_emb       # (the language input of this fusion type: [B, 1024])
            return self._forward(vox, proprio, lang_token_embs, training, save, seed, proprio_left)
        finally:
            ops.PRECISION = 'fp32'

    def _forward(self, vox, proprio, lang_token_embs, training, save, seed, proprio_left=None):
        m = self.m
        B = vox.shape[0]
        V, G, C, Cx, D, L, T0, k, s = self.V, self.G, self.C, self.Cx, self.D, self.L, self.T0, self.k, self.s
        T1 = G ** 3
        Nctx = T0 + T1
        dev = vox.device
        if seed is None:
            seed = self._draw_seed() if training else 0
        p_in = m.input_dropout if training else 0.0
        p_at = m.attn_dropout if training else 0.0
        p_de = m.decoder_dropout if training else 0.0
        proprio = proprio.float().contiguous()
        if self.two:
            # both arms go through the ONE proprio_preprocess block (perceiver :721-727): rows [right | left]
            proprio = torch.cat((proprio, proprio_left.float().contiguous()), dim=0)
        lang = (lang_token_embs.float().contiguous().view(B, LANG_FEAT_DIM) if self.concat
                else lang_token_embs.float().contiguous().view(B * T0, LANG_EMB_DIM))
        if m.no_language:                                # language ablation (perceiver :374-376): the token embeddings are zeroed
            lang = torch.zeros_like(lang)
        c = {}
        # 1. input 1x1x1 conv + lrelu (perceiver :357)
        # 2. SpatialSoftmax3D + max (perceiver :360) -- taken while d0 is written (one pass over the 256 B per voxel)
        if C == 64 and FUSE_INPUT_SS:
            d0, ss0 = ops.pointwise_ss3d_fwd(vox, self.p('input_preprocess.conv3d.weight').view(C, -1),
                                             self.p('input_preprocess.conv3d.bias'), B, V)
        else:
            d0 = ops.pointwise_fwd(vox, self.p('input_preprocess.conv3d.weight').view(C, -1), self.p('input_preprocess.conv3d.bias'))
            ss0 = ops.ss3d_max_fwd(d0, V ** 3 * C, B, V, C)
        # 3. patchify (perceiver :363)
        patch = ops.conv3d(d0, ops.conv_weight_fwd(self.p('patchify.conv3d.weight')), C, B, V, G, k, -(k // 2), stride=s,
                           bias=self.p('patchify.conv3d.bias'), act=ops.ACT_LRELU)
        # 4-6. proprio, language, context assembly (perceiver :370-422)
        pp = ops.linear(proprio, self.p('proprio_preprocess.linear.weight'), self.p('proprio_preprocess.linear.bias'), ops.ACT_LRELU)
        lg = ops.linear(lang, self.p('lang_preprocess.weight'), self.p('lang_preprocess.bias'))
        ppc = torch.cat((pp[:B], pp[B:]), dim=1) if self.two else pp          # [B, C] or [B, right C | left C]
        if self.concat:                                                       # [B, proprio C | language C]: channels of every grid token
            ppc, lg = torch.cat((pp, lg), dim=1), torch.zeros(1, dtype=torch.float32, device=dev)
        pos = self.p('pos_encoding')
        if not m.pos_encoding_with_lang:          # grid-only encoding (perceiver :391-392) = the full-length one with zero rows for the language tokens
            pos = torch.cat((torch.zeros((T0, Cx), dtype=torch.float32, device=dev), pos.reshape(T1, Cx)), dim=0)
        ctx = ops.ctx_build(lg, patch, ppc, pos, B, T0, T1, C)
        ctx2d = ctx.view(B * Nctx, Cx)
        # 7. latents
        x = self.p('latents').unsqueeze(0).expand(B, L, D).contiguous().view(B * L, D)
        # 8. cross attention block + self attention stack, `iterations` times over the SAME weights (perceiver :429-437)
        pre = 'cross_attend_blocks.0'
        cn, cm, cr = ops.layernorm_fwd(ctx2d, self.p(pre + '.norm_context.weight'), self.p(pre + '.norm_context.bias'))
        c['ctx_norm'] = (cn, cm, cr)                     # (the context and its PreNorm are the same in every iteration)
        c['iters'] = []
        for it in range(m.iterations):
            sd_it = seed if it == 0 else _mix32(seed, 7919 * it)
            pre = 'cross_attend_blocks.0'
            xn, xm, xr = ops.layernorm_fwd(x, self.p(pre + '.norm.weight'), self.p(pre + '.norm.bias'))
            x1, ca = self._attn_fwd(pre, xn.view(B, L, D), cn.view(B, Nctx, Cx), m.cross_heads, m.cross_dim_head, p_in,
                                    _mix32(sd_it, 1), x, save)
            ci = dict(cross=dict(x=x, xn=xn, xm=xm, xr=xr, attn=ca) if save else None)
            x, ci['cross_ff'] = self._ff_fwd('cross_attend_blocks.1', x1, save)
            # self attention stack (perceiver :435-437)
            ci['layers'] = []
            for i in range(m.depth):
                pre = 'layers.%d.0' % i
                xn, xm, xr = ops.layernorm_fwd(x, self.p(pre + '.norm.weight'), self.p(pre + '.norm.bias'))
                x1, sa = self._attn_fwd(pre, xn.view(B, L, D), xn.view(B, L, D), m.latent_heads, m.latent_dim_head, p_at,
                                        _mix32(sd_it, 2 + i), x, save)
                x2, fc = self._ff_fwd('layers.%d.1' % i, x1, save)
                if save:
                    ci['layers'].append(dict(x=x, xn=xn, xm=xm, xr=xr, attn=sa, ff=fc))
                x = x2
            c['iters'].append(ci)
        # decoder cross attention (perceiver :440-448): queries = all context tokens, no residual
        pre = 'decoder_cross_attn'
        qn, qm, qr = ops.layernorm_fwd(ctx2d, self.p(pre + '.norm.weight'), self.p(pre + '.norm.bias'))
        ln, lm, lr = ops.layernorm_fwd(x, self.p(pre + '.norm_context.weight'), self.p(pre + '.norm_context.bias'))
        z, da = self._attn_fwd(pre, qn.view(B, Nctx, Cx), ln.view(B, L, D), m.cross_heads, m.cross_dim_head, p_de,
                               _mix32(seed, 100), None, save)
        z = z.view(B, Nctx, Cx)
        zv = z[:, T0:]                                   # [B, G^3, Cx] view, batch stride Nctx*Cx
        ss1 = ops.ss3d_max_fwd(zv, Nctx * Cx, B, G, Cx)  # perceiver :451
        zc = zv.contiguous().view(B, G, G, G, Cx)
        # up0 = conv(k) -> upsample(s) -> conv(k)   (perceiver :454; network_utils :242-250)
        z1 = ops.conv3d(zc, ops.conv_weight_fwd(self.p('up0.conv_up.0.conv3d.weight')), C, B, G, G, k, -(k // 2),
                        bias=self.p('up0.conv_up.0.conv3d.bias'), act=ops.ACT_LRELU)
        up2 = 'up0.conv_up.%d.conv3d' % (2 if s > 1 else 1)
        if s > 1:
            if self._frozen_now and self._weff_keep is not None:
                Weff = self._weff_keep
            else:
                Weff = ops.polyphase_weights(self.p(up2 + '.weight'), self.Lt(dev), s, self.kl)
                self._weff_keep = Weff if self._frozen_now else None
            if ops.polyphase_fwd_ok(C, C, self.kl, B, G):
                u0 = ops.conv3_polyphase_fwd(z1, Weff, C, B, G, k, s, self.p(up2 + '.bias').repeat(s ** 3), act=ops.ACT_LRELU)
            else:
                u0 = ops.conv3d(z1, Weff, s ** 3 * C, B, G, G, self.kl, -self.R, bias=self.p(up2 + '.bias').repeat(s ** 3),
                                act=ops.ACT_LRELU, d2s=(s, C))
        else:
            Weff = None
            u0 = ops.conv3d(z1, ops.conv_weight_fwd(self.p(up2 + '.weight')), C, B, G, G, k, -(k // 2),
                            bias=self.p(up2 + '.bias'), act=ops.ACT_LRELU)
        # ablations (perceiver :457-460): `final` reads u0 alone (no_skip_connection) or d0 alone (no_perceiver: the transformer still runs,
        # its latents reach the heads through ss1) -- one 64-channel source on the generic kernels
        skip = 'u0' if m.no_skip_connection else ('d0' if m.no_perceiver else 'cat')
        if skip != 'cat':
            u = ops.conv3d(u0 if skip == 'u0' else d0, ops.conv_weight_fwd(self.p('final.conv3d.weight')), C, B, V, V, 3, -1,
                           bias=self.p('final.conv3d.bias'), act=ops.ACT_LRELU)
            ss2 = ops.ss3d_max_fwd(u, V ** 3 * C, B, V, C)
        # final conv over cat([d0, u0]) without the cat (perceiver :462)
        elif ops.conv3_ss3d_ok(C, C, C, V):
            # ... with the pooled features of its output (:470) taken in the conv's epilogue: no statistics pass over u
            u, ss2 = ops.conv3_ss3d_fwd(d0, u0, ops.conv_weight_fwd(self.p('final.conv3d.weight')), self.p('final.conv3d.bias'), B, V)
        else:
            u = ops.conv3d(d0, ops.conv_weight_fwd(self.p('final.conv3d.weight')), C, B, V, V, 3, -1,
                           bias=self.p('final.conv3d.bias'), act=ops.ACT_LRELU, src1=u0)
            ss2 = ops.ss3d_max_fwd(u, V ** 3 * C, B, V, C)
        # translation head (perceiver :465)
        q_trans = ops.conv3_c1_fwd(u, self.p('trans_decoder.conv3d.weight'), self.p('trans_decoder.conv3d.bias'), B, V)
        feats = torch.cat([ss0[0], ss0[1], ss1[0], ss1[1], ss2[0], ss2[1]], dim=1)
        h0 = ops.linear(feats, self.p('dense0.linear.weight'), self.p('dense0.linear.bias'), ops.ACT_LRELU)
        h1 = ops.linear(h0, self.p('dense1.linear.weight'), self.p('dense1.linear.bias'), ops.ACT_LRELU)
        o = ops.linear(h1, self.p('rot_grip_collision_ff.linear.weight'), self.p('rot_grip_collision_ff.linear.bias'))
        nc = m.num_collision_classes
        outs = (q_trans.view(B, 1, V, V, V), o[:, :-nc], o[:, -nc:])
        left = None
        if self.two:
            # left arm: its own translation conv and MLP head over the SAME pooled features (ss_final_left_arm(u) is
            # ss_final(u): SpatialSoftmax3D has no parameters; perceiver :845-858)
            q_left = ops.conv3_c1_fwd(u, self.p('trans_decoder_left_arm.conv3d.weight'), self.p('trans_decoder_left_arm.conv3d.bias'), B, V)
            h0l = ops.linear(feats, self.p('dense0_left_arm.linear.weight'), self.p('dense0_left_arm.linear.bias'), ops.ACT_LRELU)
            h1l = ops.linear(h0l, self.p('dense1_left_arm.linear.weight'), self.p('dense1_left_arm.linear.bias'), ops.ACT_LRELU)
            ol = ops.linear(h1l, self.p('rot_grip_collision_ff_left_arm.linear.weight'), self.p('rot_grip_collision_ff_left_arm.linear.bias'))
            outs = outs + (q_left.view(B, 1, V, V, V), ol[:, :-nc], ol[:, -nc:])
            left = dict(h0=h0l, h1=h1l, o=ol)
        h2 = None
        if m.arm_pred_loss:
            h2 = ops.linear(feats, self.p('dense2.linear.weight'), self.p('dense2.linear.bias'), ops.ACT_LRELU)
            outs = outs + (ops.linear(h2, self.p('arm_ff.linear.weight'), self.p('arm_ff.linear.bias')),)
        if save:
            c.update(B=B, vox=vox, proprio=proprio, lang=lang, d0=d0, ss0=ss0, patch=patch, pp=pp, ctx2d=ctx2d,
                     dec=dict(qn=qn, qm=qm, qr=qr, ln=ln, lm=lm, lr=lr, x=x, attn=da), z=z, ss1=ss1, zc=zc, z1=z1,
                     Weff=Weff, u0=u0, u=u, ss2=ss2, feats=feats, h0=h0, h1=h1, h2=h2, o=o, left=left)
        return outs, (c if save else None)

    # -------------------------------------------------------------------------------------------------- backward
    # parameter-name prefixes of the gradient-exchange buckets, in the order the backward pass completes them (= from the end of
    # the module's registration order towards its start, so every bucket is one contiguous slice of the flat gradient buffer)
    def grad_buckets(self):
        out = [('tail', ['decoder_cross_attn.', 'up0.', 'final.', 'trans_decoder.', 'dense0.', 'dense1.',
                         'rot_grip_collision_ff.', 'dense2.', 'arm_ff.', 'trans_decoder_left_arm.', 'dense0_left_arm.',
                         'dense1_left_arm.', 'rot_grip_collision_ff_left_arm.'])]
        out += [('layers.%d' % i, ['layers.%d.' % i]) for i in reversed(range(self.m.depth))]
        out.append(('head', ['pos_encoding', 'latents', 'input_preprocess.', 'patchify.', 'lang_preprocess.',
                             'proprio_preprocess.', 'cross_attend_blocks.']))
        return out

    def backward(self, c, dq_trans, d_o, d_arm=None, on_bucket_ready=None, dq_trans_left=None, d_o_left=None):
        """dq_trans [B,V,V,V] (or [B,1,V,V,V]), d_o [B, 3*rot+grip+coll] (grad of the concatenated MLP head output),
        d_arm [B,2] or None.  Accumulates into every parameter's .grad.  `on_bucket_ready(name)` is called as soon as the
        last kernel that writes gradients of bucket `name` (see grad_buckets) has been enqueued."""
        ops.PRECISION = self.bwd_precision or self.precision
        ops.WGRAD_PRECISION = 'fp16' if (ops.PRECISION == 'bf16x3' and self.wgrad_precision == 'fp16') else ''
        ops.GENERIC_WGRAD_F16 = bool(ops.WGRAD_PRECISION) and self.generic_wgrad_f16
        ops._GRAD_SCALE = self._grad_scales
        ops.begin_backward()
        self._on_bucket = on_bucket_ready
        try:
            return self._backward(c, dq_trans, d_o, d_arm, dq_trans_left, d_o_left)
        finally:
            ops.PRECISION = 'fp32'
            ops.WGRAD_PRECISION = ''
            ops.GENERIC_WGRAD_F16 = False

    def _backward(self, c, dq_trans, d_o, d_arm=None, dq_trans_left=None, d_o_left=None):
        m = self.m
        B = c['B']
        V, G, C, Cx, D, L, T0, k, s = self.V, self.G, self.C, self.Cx, self.D, self.L, self.T0, self.k, self.s
        T1 = G ** 3
        Nctx = T0 + T1
        dev = dq_trans.device
        dq_trans = dq_trans.contiguous().view(B, V, V, V)

        def E(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        # ---- MLP heads (perceiver :472-483)
        dh1 = E(B, c['h1'].shape[1])
        ops.linear_bwd(c['h1'], self.p('rot_grip_collision_ff.linear.weight'), d_o.contiguous(),
                       self.g('rot_grip_collision_ff.linear.weight'), self.g('rot_grip_collision_ff.linear.bias'), dh1)
        ops.lrelu_bwd_(dh1, c['h1'])
        dh0 = E(B, 256)
        ops.linear_bwd(c['h0'], self.p('dense1.linear.weight'), dh1, self.g('dense1.linear.weight'), self.g('dense1.linear.bias'), dh0)
        ops.lrelu_bwd_(dh0, c['h0'])
        dfe = E(B, c['feats'].shape[1])
        ops.linear_bwd(c['feats'], self.p('dense0.linear.weight'), dh0, self.g('dense0.linear.weight'), self.g('dense0.linear.bias'), dfe)
        if m.arm_pred_loss and d_arm is not None:
            dh2 = E(B, C)
            ops.linear_bwd(c['h2'], self.p('arm_ff.linear.weight'), d_arm.contiguous(), self.g('arm_ff.linear.weight'),
                           self.g('arm_ff.linear.bias'), dh2)
            ops.lrelu_bwd_(dh2, c['h2'])
            ops.linear_bwd(c['feats'], self.p('dense2.linear.weight'), dh2, self.g('dense2.linear.weight'),
                           self.g('dense2.linear.bias'), dfe, dx_accumulate=True)
        if self.two:
            if dq_trans_left is None or d_o_left is None:
                raise VoxactbHipError('the 2Robots encoder needs the left arm\'s loss gradients')
            cl = c['left']
            dh1l = E(B, cl['h1'].shape[1])
            ops.linear_bwd(cl['h1'], self.p('rot_grip_collision_ff_left_arm.linear.weight'), d_o_left.contiguous(),
                           self.g('rot_grip_collision_ff_left_arm.linear.weight'), self.g('rot_grip_collision_ff_left_arm.linear.bias'), dh1l)
            ops.lrelu_bwd_(dh1l, cl['h1'])
            dh0l = E(B, 256)
            ops.linear_bwd(cl['h0'], self.p('dense1_left_arm.linear.weight'), dh1l, self.g('dense1_left_arm.linear.weight'),
                           self.g('dense1_left_arm.linear.bias'), dh0l)
            ops.lrelu_bwd_(dh0l, cl['h0'])
            ops.linear_bwd(c['feats'], self.p('dense0_left_arm.linear.weight'), dh0l, self.g('dense0_left_arm.linear.weight'),
                           self.g('dense0_left_arm.linear.bias'), dfe, dx_accumulate=True)
        o0 = 0
        gs = []
        for width in (3 * C, C, 3 * Cx, Cx, 3 * C, C):
            gs.append(dfe[:, o0:o0 + width].contiguous())
            o0 += width
        # ---- u: ss_final/max + trans_decoder
        u, d0, u0 = c['u'], c['d0'], c['u0']
        du = E(B, V, V, V, C)
        ss, mx, st, am = c['ss2']
        wt = self.p('trans_decoder.conv3d.weight')
        # one pass over u: the pooled features' term, the translation head's data gradient, final's LeakyReLU' and the column
        # sums of the result (final's bias gradient) -- instead of ss3d_max_bwd, conv3_c1_dgrad and colsum one after the other
        fuse_u = FUSE_U_BWD and ops.c1_dgrad_ss3d_ok(V, C)
        if not fuse_u:
            ops.ss3d_max_bwd(u, V ** 3 * C, B, V, C, st, ss, am, gs[4], gs[5], du, V ** 3 * C)
        ops.conv3_c1_wgrad(u, dq_trans, self.g('trans_decoder.conv3d.weight'), self.g('trans_decoder.conv3d.bias'), B, V)
        if self.two:
            dql = dq_trans_left.contiguous().view(B, V, V, V)
            ops.conv3_c1_wgrad(u, dql, self.g('trans_decoder_left_arm.conv3d.weight'), self.g('trans_decoder_left_arm.conv3d.bias'), B, V)
            ops.conv3_c1_dgrad(dql, self.p('trans_decoder_left_arm.conv3d.weight'), u, du, B, V, accumulate=not fuse_u, mask=False)
        f16_leaf = ops.WGRAD_PRECISION == 'fp16' and C == 64 and ops.dgrad_fold_ok(C, 2 * C, V)
        sc_du = None
        if fuse_u:
            r = ops.conv3_c1_dgrad_ss3d(dq_trans, wt, u, du, B, V, st, ss, am, gs[4], gs[5], self.g('final.conv3d.bias'),
                                        accumulate=self.two, want_scale=f16_leaf)
            if f16_leaf:
                sc_du = r[1]
        else:
            ops.conv3_c1_dgrad(dq_trans, wt, u, du, B, V, accumulate=True, mask=True)      # du is now d(pre-activation of `final`)
        # ---- final conv (two sources)
        Wf = self.p('final.conv3d.weight')
        # one |du| maximum serves both single-fp16-product kernels that read du (the weight gradient and the d(d0) half of the data
        # gradient); None in the other precisions
        if f16_leaf and sc_du is None:
            sc_du = ops.absmax_scale(du)
        skip = 'u0' if m.no_skip_connection else ('d0' if m.no_perceiver else 'cat')
        if skip == 'cat':
            dWt = ops.conv3d_wgrad(d0, du, C, B, V, V, 3, -1, src1=u0, dy_scale=sc_du)
            self.g('final.conv3d.weight').add_(dWt.view(27, 2 * C, C).permute(2, 1, 0).reshape(Wf.shape))
        else:
            dWt = ops.conv3d_wgrad(u0 if skip == 'u0' else d0, du, C, B, V, V, 3, -1, dy_scale=sc_du)
            self.g('final.conv3d.weight').add_(dWt.view(27, C, C).permute(2, 1, 0).reshape(Wf.shape))
        if not fuse_u:
            ops.colsum(du.view(-1, C), self.g('final.conv3d.bias'), accumulate=True)
        sc_du0 = None
        du0_bias_done = False
        # the pooled-feature gradient of d0 (ss0) is added inside the input conv's weight-gradient kernel, the last reader of
        # dd0; otherwise it is dd0's first writer
        fuse_ss0 = C == 64 and FUSE_INPUT_SS and c['vox'].shape[-1] == 10
        Wp = self.p('patchify.conv3d.weight')
        gW_in, gb_in = self.g('input_preprocess.conv3d.weight').view(C, -1), self.g('input_preprocess.conv3d.bias')
        # d(d0) only feeds the input conv's weight gradient (the voxel grid is a detached input): when every conv path into d0 can add
        # its share of dW_in / db_in itself -- `final`'s data gradient in its fold epilogue, the patchify data gradient in
        # patch_wgrad.hip -- the 4.1 GB tensor dd0 never exists and the input conv's kernel only adds the pooled-feature term
        no_dd0 = (skip == 'cat' and fuse_ss0 and ops.wgin_fold_ok(C, 2 * C, V, c['vox'].shape[-1])
                  and ops.patch_dgrad_input_wgrad_ok(k, s, C, c['vox'].shape[-1]))
        dd0 = None if no_dd0 else E(B, V, V, V, C)
        if skip == 'u0' and fuse_ss0:
            dd0.zero_()                      # (no conv writes d(d0) first under no_skip_connection: the patchify fold and ss0 only add)
        if not fuse_ss0:
            ss, mx, st, am = c['ss0']
            ops.ss3d_max_bwd(d0, V ** 3 * C, B, V, C, st, ss, am, gs[0], gs[1], dd0, V ** 3 * C)
        du0 = E(B, V, V, V, C) if skip != 'd0' else None
        if skip != 'cat':
            dsrc = ops.conv3d(du, ops.conv_weight_dgrad(Wf), C, B, V, V + 2, 3, -2, replicate=False)
            if skip == 'u0':
                ops.fold_pad(dsrc, V + 2, C, 0, du0, B, V, C, 1, lrelu_of=u0)         # d(pre-activation of up0's last conv)
            else:
                ops.fold_pad(dsrc, V + 2, C, 0, dd0, B, V, C, 1, accumulate=not fuse_ss0)
            del dsrc
        elif C == 64 and ops.dgrad_fold_ok(C, 2 * C, V):
            # data gradient and the adjoint of the replicate padding in one kernel: the first 64 columns go (add) into dd0,
            # the other 64 become d(pre-activation of up0's last conv) through u0's LeakyReLU'
            # d(d0) feeds nothing but the weight gradient of the 1x1x1 input conv (the voxel grid is a detached input, agent :100):
            # a leaf -- its column block may run on single fp16 products; d(u0) propagates through the decoder: two fp16 products (dY hi + lo, ops.DGRAD_PRECISION)
            up2b = 'up0.conv_up.%d.conv3d.bias' % (2 if s > 1 else 1)
            sc_du0 = ops.conv3_dgrad_fold(du, ops.conv_weight_dgrad(Wf), B, V, 2 * C, [(dd0, not fuse_ss0, None), (du0, False, u0)],
                                          dy_scale=sc_du, leaf_blocks=(0,), scale_blocks=(1,), colsum_into={1: self.g(up2b)},
                                          wgin={0: (d0, c['vox'], gW_in, gb_in)} if no_dd0 else None).get(1)
            du0_bias_done = True
        else:
            dcat = ops.conv3d(du, ops.conv_weight_dgrad(Wf), 2 * C, B, V, V + 2, 3, -2, replicate=False)
            ops.fold_pad(dcat, V + 2, 2 * C, 0, dd0, B, V, C, 1, accumulate=not fuse_ss0)
            ops.fold_pad(dcat, V + 2, 2 * C, C, du0, B, V, C, 1, lrelu_of=u0)         # d(pre-activation of up0's last conv)
            del dcat
        del du
        dzp = None
        pk = k // 2
        if skip != 'd0':                 # (no_perceiver: u0 feeds nothing, the up-block has no gradient -- as in the reference, where its .grad stays None)
            # ---- up0: second conv (polyphase) -> first conv
            z1, zc = c['z1'], c['zc']
            up2 = 'up0.conv_up.%d.conv3d' % (2 if s > 1 else 1)
            W2 = self.p(up2 + '.weight')
            if not du0_bias_done:
                ops.colsum(du0.view(-1, C), self.g(up2 + '.bias'), accumulate=True)
            dz1 = E(B, G, G, G, C)
            if s > 1:
                kl, R = self.kl, self.R
                pst = ops.polyphase_structure(k, s, dev) if ops.POLY_SPARSE else None
                dWeff = ops.conv3d_wgrad(z1, du0, s ** 3 * C, B, G, G, kl, -R, d2s=(s, C),
                                         phase_mask=pst['phase_mask_t'] if pst else None, flops_frac=pst['frac'] if pst else 1.0,
                                         dy_scale=sc_du0)
                ops.polyphase_weights_bwd(dWeff, self.Lt(dev), self.g(up2 + '.weight'), s, kl)
                Sp = G + 2 * R
                if ops.s2d_halo_ok(kl, C, C):
                    # same gradient as a 3^3 conv over the low-res grid reading the fine dY by space-to-depth (LDS-halo kernel)
                    dzp = ops.conv3_s2d(du0, None, C, B, G, Sp, -(kl - 1), s, C, poly_k=k, dy_scale=sc_du0, weff_src=(c['Weff'], C, C, s, kl))
                else:
                    wd = ops.polyphase_dgrad_weights(c['Weff'], C, C, s, kl)
                    dzp = ops.conv3d(du0, wd, C, B, V, Sp, s * kl, -s * (kl - 1), stride=s, replicate=False)
                ops.fold_pad(dzp, Sp, C, 0, dz1, B, G, C, R, lrelu_of=z1)
            else:
                dWt = ops.conv3d_wgrad(z1, du0, C, B, G, G, k, -(k // 2))
                self.g(up2 + '.weight').add_(dWt.view(k ** 3, C, C).permute(2, 1, 0).reshape(W2.shape))
                dzp = ops.conv3d(du0, ops.conv_weight_dgrad(W2), C, B, G, G + 2 * (k // 2), k, -(k - 1), replicate=False)
                ops.fold_pad(dzp, G + 2 * (k // 2), C, 0, dz1, B, G, C, k // 2, lrelu_of=z1)
            del du0, dzp
            W1 = self.p('up0.conv_up.0.conv3d.weight')
            dWt = ops.conv3d_wgrad(zc, dz1, C, B, G, G, k, -(k // 2), grad_key=('conv', W1.data_ptr()))
            self.g('up0.conv_up.0.conv3d.weight').add_(dWt.view(k ** 3, Cx, C).permute(2, 1, 0).reshape(W1.shape))
            ops.colsum(dz1.view(-1, C), self.g('up0.conv_up.0.conv3d.bias'), accumulate=True)
            pk = k // 2
            dzp = ops.conv3d(dz1, ops.conv_weight_dgrad(W1), Cx, B, G, G + 2 * pk, k, -(k - 1), replicate=False)
        dzv = E(B, T1, Cx)
        ss, mx, st, am = c['ss1']
        ops.ss3d_max_bwd(c['z'][:, T0:], Nctx * Cx, B, G, Cx, st, ss, am, gs[2], gs[3], dzv, T1 * Cx)
        if dzp is not None:
            ops.fold_pad(dzp, G + 2 * pk, Cx, 0, dzv, B, G, Cx, pk, accumulate=True)
        dz = torch.zeros((B, Nctx, Cx), dtype=torch.float32, device=dev)
        dz[:, T0:] = dzv
        # ---- decoder cross attention
        dc = c['dec']
        pre = 'decoder_cross_attn'
        dqn, dln = self._attn_bwd(pre, dc['attn'], dz.view(B * Nctx, Cx), dc['qn'], dc['ln'], False)
        dctx = ops.layernorm_bwd(dqn, c['ctx2d'], self.p(pre + '.norm.weight'), dc['qm'], dc['qr'], self.g(pre + '.norm.weight'),
                                 self.g(pre + '.norm.bias'))
        dx = ops.layernorm_bwd(dln, dc['x'], self.p(pre + '.norm_context.weight'), dc['lm'], dc['lr'],
                               self.g(pre + '.norm_context.weight'), self.g(pre + '.norm_context.bias'))
        self._bucket_ready('tail')                  # heads, trans_decoder, final, up0, decoder cross attention: complete
        # ---- per iteration, last first: self-attention stack reversed, then the cross-attention block
        cn, cm, cr = c['ctx_norm']
        dcn_sum = None
        for it in reversed(range(m.iterations)):
            ci = c['iters'][it]
            for i in reversed(range(m.depth)):
                lc = ci['layers'][i]
                dx = self._ff_bwd('layers.%d.1' % i, lc['ff'], dx)
                pre = 'layers.%d.0' % i
                dxn, _ = self._attn_bwd(pre, lc['attn'], dx, lc['xn'], lc['xn'], True)
                ops.layernorm_bwd(dxn, lc['x'], self.p(pre + '.norm.weight'), lc['xm'], lc['xr'], self.g(pre + '.norm.weight'),
                                  self.g(pre + '.norm.bias'), dx=dx, accumulate_dx=True)
                if it == 0 and (i == 0 or not self._tied):
                    self._bucket_ready('layers.%d' % i)      # (shared weights: their gradients are complete after the FIRST iteration's turn)
            dx = self._ff_bwd('cross_attend_blocks.1', ci['cross_ff'], dx)
            cc = ci['cross']
            pre = 'cross_attend_blocks.0'
            dxn, dcn = self._attn_bwd(pre, cc['attn'], dx, cc['xn'], cn, False)
            ops.layernorm_bwd(dxn, cc['x'], self.p(pre + '.norm.weight'), cc['xm'], cc['xr'], self.g(pre + '.norm.weight'),
                              self.g(pre + '.norm.bias'), dx=dx, accumulate_dx=True)
            if dcn_sum is None:
                dcn_sum = dcn
            else:
                ops.axpy_(dcn_sum, dcn)                      # the context feeds every iteration's cross attention
        pre = 'cross_attend_blocks.0'
        ops.layernorm_bwd(dcn_sum, c['ctx2d'], self.p(pre + '.norm_context.weight'), cm, cr,
                          self.g(pre + '.norm_context.weight'), self.g(pre + '.norm_context.bias'), dx=dctx, accumulate_dx=True)
        ops.sum_splits(dx, B, L * D, self.g('latents'), accumulate=True)
        # ---- context assembly, language, proprio
        if m.pos_encoding_with_lang:
            dlang, dpatch, dpp = ops.ctx_bwd(dctx, self.g('pos_encoding'), B, T0, T1, C, Cx - C)
        else:
            dpos = torch.zeros((Nctx, Cx), dtype=torch.float32, device=dev)
            dlang, dpatch, dpp = ops.ctx_bwd(dctx, dpos, B, T0, T1, C, Cx - C)
            self.g('pos_encoding').view(T1, Cx).add_(dpos[T0:])
        if self.two:
            dpp = torch.cat((dpp[:, :C], dpp[:, C:]), dim=0).contiguous()      # rows [right | left], as c['proprio'] / c['pp']
        if self.concat:
            dlang, dpp = dpp[:, C:].contiguous(), dpp[:, :C].contiguous()
        ops.linear_bwd(c['lang'], self.p('lang_preprocess.weight'), dlang, self.g('lang_preprocess.weight'),
                       self.g('lang_preprocess.bias'))
        ops.lrelu_bwd_(dpp, c['pp'])
        ops.linear_bwd(c['proprio'], self.p('proprio_preprocess.linear.weight'), dpp, self.g('proprio_preprocess.linear.weight'),
                       self.g('proprio_preprocess.linear.bias'))
        # ---- patchify
        ops.lrelu_bwd_(dpatch, c['patch'])
        fuse_patch = fuse_ss0 and ops.patch_dgrad_input_wgrad_ok(k, s, C, c['vox'].shape[-1]) and (dpatch.is_contiguous() or no_dd0)
        # (fused path: the patchify weight gradient comes out of the launch that already reads d0 for the input conv's gradient)
        wp_in_fused = fuse_patch and ops.PATCH_WGRAD_WEIGHT and self.g('patchify.conv3d.weight').is_contiguous()
        if not wp_in_fused:
            dWt = ops.conv3d_wgrad(d0, dpatch, C, B, V, G, k, -pk, stride=s, grad_key=('conv', Wp.data_ptr()))
            self.g('patchify.conv3d.weight').add_(dWt.view(k ** 3, C, C).permute(2, 1, 0).reshape(Wp.shape))
        ops.colsum(dpatch, self.g('patchify.conv3d.bias'), accumulate=True)
        dxp, Sp = None, 0
        if fuse_patch:
            dpatch = dpatch.contiguous()
            # the patchify data gradient only feeds the input conv's weight gradient (the voxel grid is a detached input): its share
            # of dW_in / db_in straight from dpatch, no 105^3 x 64 gradient tensor (patch_wgrad.hip)
            ops.patch_dgrad_input_wgrad(dpatch, Wp, d0, c['vox'], gW_in, gb_in, B, V, G, k, pk,
                                        dWp=self.g('patchify.conv3d.weight') if wp_in_fused else None)
        elif s > 1:
            wtp, U = ops.strided_dgrad_weights(Wp, s)
            Gp = (V + 2 * pk + s - 1) // s
            dxp = ops.conv3d(dpatch, wtp, s ** 3 * C, B, G, Gp, U, -(U - 1), replicate=False, d2s=(s, C))
            Sp = Gp * s
        else:
            dxp = ops.conv3d(dpatch, ops.conv_weight_dgrad(Wp), C, B, G, V + 2 * pk, k, -(k - 1), replicate=False)
            Sp = V + 2 * pk
        if not fuse_ss0:          # (fused: the padding adjoint of dxp is gathered inside the input conv's weight-gradient kernel)
            ops.fold_pad(dxp, Sp, C, 0, dd0, B, V, C, pk, accumulate=True)
        # ---- input conv (its LeakyReLU' is applied inside the weight-gradient kernel)
        if fuse_ss0:
            ss, mx, st, am = c['ss0']
            ops.pointwise_wgrad_ss3d(c['vox'], d0, dd0, gW_in, gb_in, B, V, st, ss, am, gs[0], gs[1], fold_src=dxp, Sp=Sp, pad=pk)
        else:
            ops.pointwise_wgrad(c['vox'], d0, dd0, gW_in, gb_in)
        self._bucket_ready('head')

    def _bucket_ready(self, name):
        if self._on_bucket is not None:
            self._on_bucket(name)
