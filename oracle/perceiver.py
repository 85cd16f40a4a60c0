"""Oracle: PerceiverVoxelLangEncoder.forward (TEST INFRASTRUCTURE).

Functional PyTorch-CPU restatement of
/root/reference/peract/agents/peract_bc/perceiver_lang_io.py:345-485 and the
blocks it uses from /root/reference/peract/helpers/network_utils.py
(Conv3DBlock :128-170, Conv3DUpsampleBlock :237-254, DenseBlock :257-289,
SpatialSoftmax3D :773-809).  Parameters come in a flat dict keyed by the
reference's own parameter names (e.g. 'layers.3.1.fn.net.0.weight'), so a
reference state_dict can be fed in unchanged.  Only the configuration the
VoxAct-B single-arm / acting / stabilizing policies use is covered:
lang_fusion_type='seq', pos_encoding_with_lang=True, no ablation flags.
Dropout is the identity here (fixtures are generated with p=0 / eval()).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.02  # network_utils.py:12


def param_shapes(depth, voxel_size, low_dim_size, num_latents=512, latent_dim=512,
                 im_channels=64, cross_heads=1, latent_heads=8, cross_dim_head=64,
                 latent_dim_head=64, voxel_patch_size=5, voxel_patch_stride=5,
                 final_dim=64, initial_dim=10, num_rotation_classes=72,
                 num_grip_classes=2, num_collision_classes=2, arm_pred_loss=False):
    """Parameter name -> shape, as created by perceiver_lang_io.py:137-334."""
    C, D = im_channels, latent_dim
    Cx = 2 * C                                   # input_dim_before_seq (:176)
    G = voxel_size // voxel_patch_stride         # :173
    k = voxel_patch_size
    s = {}
    s['pos_encoding'] = (1, 77 + G ** 3, Cx)     # :183-185
    s['latents'] = (num_latents, D)              # :234
    s['input_preprocess.conv3d.weight'] = (C, initial_dim, 1, 1, 1)
    s['input_preprocess.conv3d.bias'] = (C,)
    s['patchify.conv3d.weight'] = (C, C, k, k, k)
    s['patchify.conv3d.bias'] = (C,)
    s['lang_preprocess.weight'] = (Cx, 512)
    s['lang_preprocess.bias'] = (Cx,)
    s['proprio_preprocess.linear.weight'] = (C, low_dim_size)
    s['proprio_preprocess.linear.bias'] = (C,)

    def attn(prefix, qdim, cdim, heads, dh, ctx_norm):
        inner = heads * dh
        s[prefix + '.fn.to_q.weight'] = (inner, qdim)
        s[prefix + '.fn.to_kv.weight'] = (2 * inner, cdim)
        s[prefix + '.fn.to_out.weight'] = (qdim, inner)
        s[prefix + '.fn.to_out.bias'] = (qdim,)
        s[prefix + '.norm.weight'] = (qdim,)
        s[prefix + '.norm.bias'] = (qdim,)
        if ctx_norm:
            s[prefix + '.norm_context.weight'] = (cdim,)
            s[prefix + '.norm_context.bias'] = (cdim,)

    def ff(prefix, dim):
        s[prefix + '.fn.net.0.weight'] = (dim * 8, dim)
        s[prefix + '.fn.net.0.bias'] = (dim * 8,)
        s[prefix + '.fn.net.2.weight'] = (dim, dim * 4)
        s[prefix + '.fn.net.2.bias'] = (dim,)
        s[prefix + '.norm.weight'] = (dim,)
        s[prefix + '.norm.bias'] = (dim,)

    attn('cross_attend_blocks.0', D, Cx, cross_heads, cross_dim_head, True)
    ff('cross_attend_blocks.1', D)
    for i in range(depth):
        attn('layers.%d.0' % i, D, D, latent_heads, latent_dim_head, False)
        ff('layers.%d.1' % i, D)
    attn('decoder_cross_attn', Cx, D, cross_heads, cross_dim_head, True)
    s['up0.conv_up.0.conv3d.weight'] = (final_dim, Cx, k, k, k)
    s['up0.conv_up.0.conv3d.bias'] = (final_dim,)
    last = 2 if voxel_patch_stride > 1 else 1     # network_utils.py:243-251
    s['up0.conv_up.%d.conv3d.weight' % last] = (final_dim, final_dim, k, k, k)
    s['up0.conv_up.%d.conv3d.bias' % last] = (final_dim,)
    s['final.conv3d.weight'] = (C, 2 * C, 3, 3, 3)
    s['final.conv3d.bias'] = (C,)
    s['trans_decoder.conv3d.weight'] = (1, final_dim, 3, 3, 3)
    s['trans_decoder.conv3d.bias'] = (1,)
    flat = C * 4 + Cx * 4 + C * 4
    nout = num_rotation_classes * 3 + num_grip_classes + num_collision_classes
    s['dense0.linear.weight'] = (256, flat)
    s['dense0.linear.bias'] = (256,)
    s['dense1.linear.weight'] = (final_dim, 256)
    s['dense1.linear.bias'] = (final_dim,)
    s['rot_grip_collision_ff.linear.weight'] = (nout, final_dim)
    s['rot_grip_collision_ff.linear.bias'] = (nout,)
    if arm_pred_loss:
        s['dense2.linear.weight'] = (final_dim, flat)
        s['dense2.linear.bias'] = (final_dim,)
        s['arm_ff.linear.weight'] = (2, final_dim)
        s['arm_ff.linear.bias'] = (2,)
    return s


def lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def conv3d_block(x, w, b, stride=1, act=True):
    """network_utils.py:128-170: padding=k//2, padding_mode='replicate'."""
    p = w.shape[-1] // 2
    if p > 0:
        x = F.pad(x, (p,) * 6, mode='replicate')
    y = F.conv3d(x, w, b, stride=stride)
    return lrelu(y) if act else y


def spatial_softmax3d(x):
    """network_utils.py:773-809 incl. the np.meshgrid 'xy' quirk (:782-786):
    over flat index i*H*W + j*W + k, pos_x varies with j, pos_y with i, pos_z with k.
    x [B,C,D,H,W] (D=H=W) -> [B, 3C] laid out (c0_x, c0_y, c0_z, c1_x, ...)."""
    B, C, D, H, W = x.shape
    px, py, pz = np.meshgrid(np.linspace(-1., 1., D), np.linspace(-1., 1., H), np.linspace(-1., 1., W))
    px = torch.from_numpy(px.reshape(-1)).float()
    py = torch.from_numpy(py.reshape(-1)).float()
    pz = torch.from_numpy(pz.reshape(-1)).float()
    f = x.reshape(-1, D * H * W)
    a = F.softmax(f / 0.01, dim=-1)
    ex = torch.sum(px * a, dim=1, keepdim=True)
    ey = torch.sum(py * a, dim=1, keepdim=True)
    ez = torch.sum(pz * a, dim=1, keepdim=True)
    return torch.cat([ex, ey, ez], 1).view(-1, C * 3)


def global_maxpool(x):
    return x.amax(dim=(2, 3, 4))


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def attention(P, prefix, x, context, heads):
    """perceiver_lang_io.py:107-132 with PreNorm :56-71 applied by the caller."""
    q = x @ P[prefix + '.fn.to_q.weight'].t()
    kv = context @ P[prefix + '.fn.to_kv.weight'].t()
    k, v = kv.chunk(2, dim=-1)
    B, n, inner = q.shape
    dh = inner // heads

    def split(t):
        return t.view(B, t.shape[1], heads, dh).permute(0, 2, 1, 3).reshape(B * heads, t.shape[1], dh)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum('bid,bjd->bij', q, k) * (dh ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum('bij,bjd->bid', attn, v)
    out = out.view(B, heads, n, dh).permute(0, 2, 1, 3).reshape(B, n, inner)
    return out @ P[prefix + '.fn.to_out.weight'].t() + P[prefix + '.fn.to_out.bias']


def prenorm_attention(P, prefix, x, context, heads):
    xn = layer_norm(x, P[prefix + '.norm.weight'], P[prefix + '.norm.bias'])
    if context is None:
        cn = xn
    else:
        cn = layer_norm(context, P[prefix + '.norm_context.weight'], P[prefix + '.norm_context.bias'])
    return attention(P, prefix, xn, cn, heads)


def prenorm_ff(P, prefix, x):
    """perceiver_lang_io.py:74-90: Linear -> GEGLU (x * gelu_erf(gates)) -> Linear."""
    xn = layer_norm(x, P[prefix + '.norm.weight'], P[prefix + '.norm.bias'])
    h = xn @ P[prefix + '.fn.net.0.weight'].t() + P[prefix + '.fn.net.0.bias']
    a, g = h.chunk(2, dim=-1)
    h = a * F.gelu(g)
    return h @ P[prefix + '.fn.net.2.weight'].t() + P[prefix + '.fn.net.2.bias']


def dense(P, prefix, x, act=True):
    y = x @ P[prefix + '.linear.weight'].t() + P[prefix + '.linear.bias']
    return lrelu(y) if act else y


def forward(P, ins, proprio, lang_token_embs, *, depth, voxel_patch_stride=5,
            cross_heads=1, latent_heads=8, num_collision_classes=2,
            arm_pred_loss=False, iterations=1, return_intermediates=False):
    """ins [B,10,V,V,V] (channels-first view of the voxel grid), proprio [B,low_dim],
    lang_token_embs [B,77,512].  Returns (trans [B,1,V,V,V], rot_and_grip [B,218],
    collision [B,2][, arm [B,2]]).  perceiver_lang_io.py:345-485."""
    I = {}
    s = voxel_patch_stride
    d0 = conv3d_block(ins, P['input_preprocess.conv3d.weight'], P['input_preprocess.conv3d.bias'])   # :357
    feats = [spatial_softmax3d(d0), global_maxpool(d0)]                                             # :360
    x = conv3d_block(d0, P['patchify.conv3d.weight'], P['patchify.conv3d.bias'], stride=s)           # :363
    B, C, G, _, _ = x.shape
    p = dense(P, 'proprio_preprocess', proprio)                                                      # :370
    p = p.view(B, -1, 1, 1, 1).expand(B, p.shape[1], G, G, G)
    x = torch.cat([x, p], dim=1)                                                                     # :372-373
    tok = x.permute(0, 2, 3, 4, 1).reshape(B, G ** 3, -1)                                            # :389,:412
    l = lang_token_embs @ P['lang_preprocess.weight'].t() + P['lang_preprocess.bias']              # :417
    ctx = torch.cat([l, tok], dim=1) + P['pos_encoding']                                             # :418-422
    I['d0'], I['ctx'] = d0, ctx
    lat = P['latents'].unsqueeze(0).expand(B, -1, -1)                                                # :425
    for _ in range(iterations):
        lat = prenorm_attention(P, 'cross_attend_blocks.0', lat, ctx, cross_heads) + lat            # :431
        lat = prenorm_ff(P, 'cross_attend_blocks.1', lat) + lat                                      # :432
        for i in range(depth):
            lat = prenorm_attention(P, 'layers.%d.0' % i, lat, None, latent_heads) + lat            # :436
            lat = prenorm_ff(P, 'layers.%d.1' % i, lat) + lat                                        # :437
    I['latents_out'] = lat
    z = prenorm_attention(P, 'decoder_cross_attn', ctx, lat, cross_heads)                            # :440
    z = z[:, l.shape[1]:]                                                                            # :444
    z = z.view(B, G, G, G, -1).permute(0, 4, 1, 2, 3)                                                # :447-448
    I['z'] = z
    feats += [spatial_softmax3d(z.contiguous()), global_maxpool(z)]                                  # :451
    u0 = conv3d_block(z, P['up0.conv_up.0.conv3d.weight'], P['up0.conv_up.0.conv3d.bias'])           # network_utils.py:242
    I['z1'] = u0
    if s > 1:
        u0 = F.interpolate(u0, scale_factor=s, mode='trilinear', align_corners=False)               # :245-247
        u0 = conv3d_block(u0, P['up0.conv_up.2.conv3d.weight'], P['up0.conv_up.2.conv3d.bias'])      # :248-250
    else:
        u0 = conv3d_block(u0, P['up0.conv_up.1.conv3d.weight'], P['up0.conv_up.1.conv3d.bias'])
    I['u0'] = u0
    u = conv3d_block(torch.cat([d0, u0], dim=1), P['final.conv3d.weight'], P['final.conv3d.bias'])   # :462
    I['u'] = u
    trans = conv3d_block(u, P['trans_decoder.conv3d.weight'], P['trans_decoder.conv3d.bias'], act=False)  # :465
    feats += [spatial_softmax3d(u.contiguous()), global_maxpool(u)]                                  # :470
    fcat = torch.cat(feats, dim=1)
    I['feats'] = fcat
    h = dense(P, 'dense0', fcat)                                                                     # :472
    h = dense(P, 'dense1', h)                                                                        # :473
    o = dense(P, 'rot_grip_collision_ff', h, act=False)                                              # :475
    rot_and_grip = o[:, :-num_collision_classes]                                                     # :476
    collision = o[:, -num_collision_classes:]                                                        # :477
    outs = (trans, rot_and_grip, collision)
    if arm_pred_loss:
        arm = dense(P, 'arm_ff', dense(P, 'dense2', fcat), act=False)                               # :479-483
        outs = outs + (arm,)
    if return_intermediates:
        return outs, I
    return outs
