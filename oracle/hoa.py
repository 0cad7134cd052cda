"""ORACLE — test infrastructure only (see oracle/__init__.py).

numpy restatement (inference / eval mode) of the Height-aware Opacity-based Attention blocks:

  HeightAttention              mmdet3d/models/necks/view_transformer_ocrf.py:421-461
  OpacityVoxelToBEVConverter   view_transformer_ocrf.py:463-518                      (HOA-2)
  ObatinOpacityMask + gate     view_transformer_ocrf.py:230-242, applied :1197-1199   (HOA-3)
  DeformableAttention2D + CPB  mmdet3d/ops/cross_attention_2d.py:49-220              (HOA-1)
  HOA-1 glue (bilinear resize, residual)  view_transformer_ocrf.py:1159-1161

Weights come in as a flat dict ``{name: ndarray}`` with the reference's state_dict keys under a
prefix.  Pinned by tests/golden/hoa.npz (outputs of the reference modules with seeded weights).
"""
import numpy as np

f32 = np.float32


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)


def conv2d(x, w, b=None, stride=1, padding=0, groups=1):
    """x (B,Cin,H,W), w (Cout,Cin/groups,kh,kw) -> (B,Cout,Ho,Wo); float64 accumulation."""
    B, Cin, H, W = x.shape
    Cout, Cg, kh, kw = w.shape
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (padding, padding), (padding, padding)))
    Ho = (H + 2 * padding - kh) // stride + 1
    Wo = (W + 2 * padding - kw) // stride + 1
    out = np.zeros((B, Cout, Ho, Wo), np.float64)
    opg = Cout // groups
    for g in range(groups):
        xs = xp[:, g * Cg:(g + 1) * Cg]
        ws = w[g * opg:(g + 1) * opg].astype(np.float64)
        for i in range(kh):
            for j in range(kw):
                patch = xs[:, :, i:i + stride * Ho:stride, j:j + stride * Wo:stride]
                out[:, g * opg:(g + 1) * opg] += np.einsum('bchw,oc->bohw', patch, ws[:, :, i, j])
    if b is not None:
        out += b.astype(np.float64)[None, :, None, None]
    return out.astype(f32)


def conv_transpose2d_k2s2(x, w, b):
    """nn.ConvTranspose2d(kernel=2, stride=2): w (Cin,Cout,2,2)."""
    B, Cin, H, W = x.shape
    Cout = w.shape[1]
    out = np.zeros((B, Cout, 2 * H, 2 * W), np.float64)
    for i in range(2):
        for j in range(2):
            out[:, :, i::2, j::2] = np.einsum('bchw,co->bohw', x.astype(np.float64), w[:, :, i, j].astype(np.float64))
    return (out + b.astype(np.float64)[None, :, None, None]).astype(f32)


def batchnorm_eval(x, p, prefix, eps=1e-5):
    g, b = p[prefix + '.weight'], p[prefix + '.bias']
    m, v = p[prefix + '.running_mean'], p[prefix + '.running_var']
    s = (g / np.sqrt(v.astype(np.float64) + eps))
    return ((x.astype(np.float64) - m[None, :, None, None]) * s[None, :, None, None] + b[None, :, None, None]).astype(f32)


def maxpool2(x):
    B, C, H, W = x.shape
    return x[:, :, :H // 2 * 2, :W // 2 * 2].reshape(B, C, H // 2, 2, W // 2, 2).max((3, 5))


def height_attention(x, p, prefix):
    """view_transformer_ocrf.py:447-461: per channel quarter, global max pool -> 1x1 -> ReLU ->
    1x1 -> concat -> sigmoid.  Returns (B,C,1,1)."""
    B, C = x.shape[:2]
    q = C // 4
    outs = []
    for k in range(4):
        pooled = x[:, k * q:(k + 1) * q].max((2, 3), keepdims=True)                   # AdaptiveMaxPool2d(1)
        h = np.maximum(conv2d(pooled, p[f'{prefix}.conv{k + 1}.0.weight']), 0)
        outs.append(conv2d(h, p[f'{prefix}.conv{k + 1}.2.weight']))
    return sigmoid(np.concatenate(outs, 1))


def _conv_block(x, p, prefix):
    """depthwise 3x3 -> 1x1 -> BN -> ReLU (view_transformer_ocrf.py:485-491)."""
    c = x.shape[1]
    x = conv2d(x, p[prefix + '.0.weight'], p[prefix + '.0.bias'], padding=1, groups=c)
    x = conv2d(x, p[prefix + '.1.weight'], p[prefix + '.1.bias'])
    return np.maximum(batchnorm_eval(x, p, prefix + '.2'), 0)


def opacity_voxel_to_bev(x, position, p, prefix='v2b'):
    """view_transformer_ocrf.py:497-518: (B,13,Y,X) + (B,4,Y,X) -> (B,1,Y,X)."""
    enc1 = _conv_block(x, p, prefix + '.encoder1') + position
    enc1 = height_attention(enc1, p, prefix + '.ca1') * enc1
    enc2 = _conv_block(maxpool2(enc1), p, prefix + '.encoder2')
    enc2 = height_attention(enc2, p, prefix + '.ca2') * enc2
    bott = _conv_block(maxpool2(enc2), p, prefix + '.bottleneck')
    bott = height_attention(bott, p, prefix + '.ca_bottleneck') * bott
    dec2 = conv_transpose2d_k2s2(bott, p[prefix + '.upconv2.weight'], p[prefix + '.upconv2.bias'])
    dec2 = _conv_block(np.concatenate((dec2, enc2), 1), p, prefix + '.decoder2')
    dec2 = height_attention(dec2, p, prefix + '.ca_dec2') * dec2
    dec1 = conv_transpose2d_k2s2(dec2, p[prefix + '.upconv1.weight'], p[prefix + '.upconv1.bias'])
    dec1 = _conv_block(np.concatenate((dec1, enc1), 1), p, prefix + '.decoder1')
    dec1 = height_attention(dec1, p, prefix + '.ca_dec1') * dec1
    return conv2d(dec1, p[prefix + '.output_conv.weight'], p[prefix + '.output_conv.bias'])


def opacity_mask(x, opacity_bev, p, prefix='mask'):
    """view_transformer_ocrf.py:236-242: sigmoid(conv7x7([mean_c, max_c]) + opacity_bev)."""
    avg = x.astype(np.float64).mean(1, keepdims=True).astype(f32)
    mx = x.max(1, keepdims=True)
    y = conv2d(np.concatenate((avg, mx), 1), p[prefix + '.conv.weight'], padding=p[prefix + '.conv.weight'].shape[-1] // 2)
    return sigmoid(y + opacity_bev)


# ------------------------------------------------------------------------------------------------
# HOA-1
# ------------------------------------------------------------------------------------------------
def interpolate_bilinear_ac(x, size):
    """F.interpolate(mode='bilinear', align_corners=True) for (B,C,H,W)."""
    B, C, H, W = x.shape
    Ho, Wo = size

    def axis(n_in, n_out):
        if n_out == 1:
            src = np.zeros(1)
        else:
            src = np.arange(n_out) * ((n_in - 1) / (n_out - 1))
        i0 = np.minimum(np.floor(src).astype(int), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, (src - i0)
    y0, y1, wy = axis(H, Ho)
    x0, x1, wx = axis(W, Wo)
    xd = x.astype(np.float64)
    top = xd[:, :, y0][:, :, :, x0] * (1 - wx) + xd[:, :, y0][:, :, :, x1] * wx
    bot = xd[:, :, y1][:, :, :, x0] * (1 - wx) + xd[:, :, y1][:, :, :, x1] * wx
    return (top * (1 - wy)[:, None] + bot * wy[:, None]).astype(f32)


def grid_sample_bilinear_zeros(x, grid):
    """F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False):
    x (B,C,H,W), grid (B,Ho,Wo,2) in [-1,1] (x then y)."""
    B, C, H, W = x.shape
    gx = ((grid[..., 0].astype(np.float64) + 1) * W - 1) / 2
    gy = ((grid[..., 1].astype(np.float64) + 1) * H - 1) / 2
    x0, y0 = np.floor(gx).astype(int), np.floor(gy).astype(int)
    out = np.zeros((B, C) + grid.shape[1:3], np.float64)
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            w = (1 - np.abs(gx - xi)) * (1 - np.abs(gy - yi))
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            xi_c, yi_c = np.clip(xi, 0, W - 1), np.clip(yi, 0, H - 1)
            for b in range(B):
                out[b] += x[b][:, yi_c[b], xi_c[b]].astype(np.float64) * (w[b] * ok[b])[None]
    return out.astype(f32)


def gelu(x):
    from math import sqrt
    from scipy.special import erf
    return (0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / sqrt(2)))).astype(f32)


def deformable_attention_2d(x_q, x_kv, p, prefix='dca', heads=1, dim_head=8, downsample_factor=4,
                            offset_scale=4, offset_kernel_size=6):
    """cross_attention_2d.py:142-220 with offset_groups = heads (= 1 in OcRFDet, call site
    view_transformer_ocrf.py:639-648), eval mode (dropout off)."""
    B, dim, h, w = x_q.shape
    q = conv2d(x_q, p[prefix + '.to_q.weight'], groups=heads)
    # offsets (:113-119): depthwise conv k,s -> GELU -> 1x1 (no bias) -> tanh -> * offset_scale
    off = conv2d(q, p[prefix + '.to_offsets.0.weight'], p[prefix + '.to_offsets.0.bias'],
                 stride=downsample_factor, padding=(offset_kernel_size - downsample_factor) // 2, groups=q.shape[1])
    off = np.tanh(conv2d(gelu(off), p[prefix + '.to_offsets.2.weight']).astype(np.float64)).astype(f32) * f32(offset_scale)
    hk, wk = off.shape[-2:]
    gx, gy = np.meshgrid(np.arange(wk, dtype=f32), np.arange(hk, dtype=f32), indexing='xy')
    vgrid = np.stack((gx, gy), 0)[None] + off                                          # (B,2,hk,wk), (x,y)
    # normalize_grid (:30-38) — NB it divides the x channel by (h-1) and y by (w-1) as written
    vx = 2.0 * vgrid[:, 0] / max(hk - 1, 1) - 1.0
    vy = 2.0 * vgrid[:, 1] / max(wk - 1, 1) - 1.0
    vgrid_scaled = np.stack((vx, vy), -1).astype(f32)                                   # (B,hk,wk,2)
    kv = grid_sample_bilinear_zeros(x_kv, vgrid_scaled)
    k = conv2d(kv, p[prefix + '.to_k.weight'], groups=heads)
    v = conv2d(kv, p[prefix + '.to_v.weight'], groups=heads)
    q = q * f32(dim_head ** -0.5)
    qf = q.reshape(B, heads, dim_head, h * w).transpose(0, 1, 3, 2).astype(np.float64)
    kf = k.reshape(B, heads, dim_head, hk * wk).transpose(0, 1, 3, 2).astype(np.float64)
    vf = v.reshape(B, heads, dim_head, hk * wk).transpose(0, 1, 3, 2).astype(np.float64)
    sim = np.einsum('bhid,bhjd->bhij', qf, kf)
    # CPB relative position bias (:49-89): grid of x_kv normalised with dim=0
    gx, gy = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32), indexing='xy')
    gq = np.stack((2.0 * gx / max(h - 1, 1) - 1.0, 2.0 * gy / max(w - 1, 1) - 1.0), -1).reshape(1, h * w, 2)
    gk = vgrid_scaled.reshape(B, hk * wk, 2)
    pos = gq[:, :, None, :].astype(np.float64) - gk[:, None, :, :].astype(np.float64)
    bias = np.sign(pos) * np.log(np.abs(pos) + 1)
    n_layers = sum(1 for key in p if key.startswith(prefix + '.rel_pos_bias.mlp.') and key.endswith('weight'))
    for li in range(n_layers):
        wk_ = p.get(f'{prefix}.rel_pos_bias.mlp.{li}.0.weight')
        if wk_ is not None:
            bias = np.maximum(bias @ wk_.T.astype(np.float64) + p[f'{prefix}.rel_pos_bias.mlp.{li}.0.bias'], 0)
        else:
            bias = bias @ p[f'{prefix}.rel_pos_bias.mlp.{li}.weight'].T.astype(np.float64) + p[f'{prefix}.rel_pos_bias.mlp.{li}.bias']
    sim = sim + bias.transpose(0, 3, 1, 2)                                               # (B,heads,i,j)
    sim = sim - sim.max(-1, keepdims=True)
    attn = np.exp(sim)
    attn /= attn.sum(-1, keepdims=True)
    out = np.einsum('bhij,bhjd->bhid', attn, vf)
    out = out.transpose(0, 1, 3, 2).reshape(B, heads * dim_head, h, w).astype(f32)
    return conv2d(out, p[prefix + '.to_out.weight'], p[prefix + '.to_out.bias'])


def hoa1(opacity, alpha_lidar, p, heights, Y, X, prefix='dca'):
    """view_transformer_ocrf.py:1159-1161: opacity (heights*Y*X,1), alpha_lidar (1,heights,Y,X)
    -> (1,heights,Y,X) = upsample(attn(down(opacity), down(alpha))) + opacity."""
    o = opacity.reshape(1, heights, Y, X)
    size = (int(Y / 6), int(X / 6))
    o_up = interpolate_bilinear_ac(o, size)
    a_up = interpolate_bilinear_ac(alpha_lidar, size)
    att = deformable_attention_2d(o_up, a_up, p, prefix)
    return interpolate_bilinear_ac(att, (Y, X)) + o, o_up
