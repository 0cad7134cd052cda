"""ORACLE — test infrastructure only (see oracle/__init__.py).

numpy restatement (inference / eval mode) of the stages of ``view_transform_core`` that sit between
the two poolings, the render and HOA:

  prefilter                     mmdet3d/models/necks/view_transformer_ocrf.py:1327-1331
  VoxelFeatureExtractor         view_transformer_ocrf.py:520-531, call :1051
  lidar_points_to_image_values  view_transformer_ocrf.py:924-942
  color_voxels                  view_transformer_ocrf.py:945-971
  retain_valid_pixels           view_transformer_ocrf.py:1004-1024
  Gaussian heads                view_transformer_ocrf.py:272-320, calls :1130-1133
  ResizeNetwork + NeRF branch   view_transformer_ocrf.py:534-554, 1094-1126
  DualFeatFusion / ProbNet / BEVGeomAttention   view_transformer_ocrf.py:36-228, calls :1183-1190

Every function follows the reference literally (materialises what the reference materialises); the
product fuses and re-associates, which is exactly what these functions check.  Weights come in as a
flat dict ``{state_dict key: ndarray}``.  Pinned by tests/golden/{heads,color_cfg0,core_small}.npz
(outputs of the reference's own Python).
"""
import numpy as np

from .hoa import batchnorm_eval, conv2d, conv_transpose2d_k2s2, sigmoid

f32 = np.float32


def softmax(x, axis):
    x = x.astype(np.float64)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return (e / e.sum(axis=axis, keepdims=True)).astype(f32)


def softplus(x):
    """nn.Softplus(beta=1, threshold=20)."""
    x = x.astype(np.float64)
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0)))).astype(f32)


def linear(x, p, prefix):
    return (x.astype(np.float64) @ p[prefix + '.weight'].astype(np.float64).T + p[prefix + '.bias']).astype(f32)


def relu(x):
    return np.maximum(x, 0)


# ---------------------------------------------------------------------------------------- a28
def prefilter(x, D, C, depth_threshold, semantic_threshold):
    """view_transformer_ocrf.py:1323-1331.  x (BN, D+2+C, H, W) ->
    depth (softmax), filter_depth, semantic (softmax over 2), filter_feat (BN,C,H,W)."""
    depth = softmax(x[:, :D], 1)
    semantic = softmax(x[:, D:D + 2], 1)
    tran_feat = x[:, D + 2:D + 2 + C]
    filter_depth = np.where(depth < f32(depth_threshold), f32(0), depth)
    img_mask = semantic[:, 1:2] >= f32(semantic_threshold)
    return depth, filter_depth, semantic, (img_mask * tran_feat).astype(f32)


# ---------------------------------------------------------------------------------------- a11
def voxel_lift(bev, p, prefix='ObtainVoxelFeature', eps=1e-5):
    """VoxelFeatureExtractor on ``bev.permute(0,2,3,1).unsqueeze(1)`` (:1051): Conv3d(1->Zh, k=1) +
    BatchNorm3d (eval) + ReLU.  bev (B,C,Y,X) -> (B,Zh,Y,X,C)."""
    w = p[prefix + '.conv.0.weight'].reshape(-1).astype(np.float64)
    b = p[prefix + '.conv.0.bias'].astype(np.float64)
    g, beta = p[prefix + '.conv.1.weight'].astype(np.float64), p[prefix + '.conv.1.bias'].astype(np.float64)
    mean, var = p[prefix + '.conv.1.running_mean'].astype(np.float64), p[prefix + '.conv.1.running_var'].astype(np.float64)
    xin = bev.transpose(0, 2, 3, 1)[:, None].astype(np.float64)                   # (B,1,Y,X,C)
    y = xin * w[None, :, None, None, None] + b[None, :, None, None, None]
    y = (y - mean[None, :, None, None, None]) / np.sqrt(var + eps)[None, :, None, None, None]
    y = y * g[None, :, None, None, None] + beta[None, :, None, None, None]
    return relu(y).astype(f32)


# ---------------------------------------------------------------------------------------- a12
def grid_sample_bilinear(img, px, py):
    """F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) of one image
    (C,H,W) at pixel coordinates produced exactly as :929-931 + ATen's un-normalisation do:
    ``xn = (px/(W-1))*2-1`` then ``ix = ((xn+1)/2)*(W-1)``, all float32."""
    C, H, W = img.shape
    with np.errstate(invalid='ignore', over='ignore'):      # non-finite coordinates take the zero-padding path
        return _grid_sample_bilinear(img, px, py, C, H, W)


def _grid_sample_bilinear(img, px, py, C, H, W):
    xn = (px.astype(f32) / f32(W - 1)) * f32(2) - f32(1)
    yn = (py.astype(f32) / f32(H - 1)) * f32(2) - f32(1)
    ix = ((xn + f32(1)) / f32(2)) * f32(W - 1)
    iy = ((yn + f32(1)) / f32(2)) * f32(H - 1)
    x0, y0 = np.floor(ix), np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wnw = (x1 - ix) * (y1 - iy)
    wne = (ix - x0) * (y1 - iy)
    wsw = (x1 - ix) * (iy - y0)
    wse = (ix - x0) * (iy - y0)
    out = np.zeros((C,) + ix.shape, np.float64)
    for xs, ys, wgt in ((x0, y0, wnw), (x1, y0, wne), (x0, y1, wsw), (x1, y1, wse)):
        ok = (xs >= 0) & (xs <= W - 1) & (ys >= 0) & (ys <= H - 1) & np.isfinite(ix) & np.isfinite(iy)
        xi = np.where(ok, xs, 0).astype(np.int64)
        yi = np.where(ok, ys, 0).astype(np.int64)
        out += np.where(ok, wgt, 0).astype(np.float64) * img[:, yi, xi]
    return out.astype(f32)


def lidar_points_to_image_values(pillars, imgs, mask):
    """:924-942.  pillars (B,N,Zh,Q,2) pixel coordinates, imgs (B,N,C,H,W), mask (B,N,Zh,Q,1) bool
    -> (B,N,Zh,Q,C)."""
    B, N, Zh, Q, _ = pillars.shape
    C = imgs.shape[2]
    out = np.zeros((B, N, Zh, Q, C), f32)
    for b in range(B):
        for n in range(N):
            v = grid_sample_bilinear(imgs[b, n].astype(f32), pillars[b, n, ..., 0].reshape(-1),
                                     pillars[b, n, ..., 1].reshape(-1))
            out[b, n] = v.T.reshape(Zh, Q, C)
    return out * mask.astype(f32)


# ---------------------------------------------------------------------------------------- a13
def color_voxels_avg(img_values, mask):
    """:945-959, the ``avg_color`` output: mean over the cameras whose mask holds.  -> (B,Zh,Q,C)."""
    m = mask[..., 0].astype(bool)                               # (B,N,Zh,Q)
    vals = np.where(m[..., None], img_values, f32(0)).astype(f32)
    cnt = m.sum(1).astype(f32)
    cnt = np.where(cnt == 0, f32(1), cnt)
    acc = np.zeros(vals.shape[:1] + vals.shape[2:], f32)
    for n in range(vals.shape[1]):                              # sequential fp32 sum over cameras
        acc = acc + vals[:, n]
    return (acc / cnt[..., None]).astype(f32)


# ---------------------------------------------------------------------------------------- a14
def retain_valid_pixels(images, pix, mask):
    """:1004-1024.  images (B,N,3,H,W), pix (B,N,Zh,Q,2) pixel coordinates, mask (B,N,Zh,Q,1) ->
    255 everywhere except the pixels some valid projection lands on (truncated, clamped to
    [0, max(W,H)-1]), which keep the image value."""
    B, N, _, H, W = images.shape
    out = np.full_like(images, 255)
    hi = max(W, H) - 1
    for b in range(B):
        for n in range(N):
            m = mask[b, n, ..., 0].astype(bool).reshape(-1)
            p = pix[b, n].reshape(-1, 2)[m]
            p = p[p[:, 0] != -1]                                # the reference's sentinel test (:1017)
            xy = np.clip(p.astype(np.int64), 0, hi)             # .long() truncates toward zero
            out[b, n][:, xy[:, 1], xy[:, 0]] = images[b, n][:, xy[:, 1], xy[:, 0]]
    return out


# ---------------------------------------------------------------------------------------- a15
def gauss_heads(voxel_feat, rgb01, p):
    """:1130-1133 with :272-320.  voxel_feat (P,80), rgb01 (P,3) = colour/255 ->
    opacity (P,1), scales (P,3), rotations (P,4), colour (P,3)."""
    def mlp(x, name):
        return linear(relu(linear(x, p, name + '.fc1')), p, name + '.fc2')
    opacity = sigmoid(mlp(voxel_feat, 'A_MLP'))
    scales = softplus(mlp(voxel_feat, 'S_MLP'))
    r = mlp(voxel_feat, 'R_MLP').astype(np.float64)
    rot = (r / np.maximum(np.sqrt((r * r).sum(-1, keepdims=True)), 1e-12)).astype(f32)
    color = sigmoid(mlp(np.concatenate((voxel_feat, rgb01), -1), 'C_MLP'))
    return opacity, scales, rot, color


# ---------------------------------------------------------------------------------------- a23
def conv_transpose2d_k4s4(x, w, b):
    """nn.ConvTranspose2d(kernel=4, stride=4): w (Cin,Cout,4,4)."""
    B, Cin, H, W = x.shape
    out = np.zeros((B, w.shape[1], 4 * H, 4 * W), np.float64)
    for i in range(4):
        for j in range(4):
            out[:, :, i::4, j::4] = np.einsum('bchw,co->bohw', x.astype(np.float64), w[:, :, i, j].astype(np.float64))
    return (out + b.astype(np.float64)[None, :, None, None]).astype(f32)


def resize_network(x, p, prefix='image_feat_resize'):
    """ResizeNetwork :534-554 (no non-linearity anywhere).  x (B,256,h,w) -> (B,80,16h,16w)."""
    x = conv2d(x, p[prefix + '.conv1.weight'], p[prefix + '.conv1.bias'], padding=1)
    x = conv_transpose2d_k2s2(x, p[prefix + '.upsample1.weight'], p[prefix + '.upsample1.bias'])
    x = conv2d(x, p[prefix + '.conv2.weight'], p[prefix + '.conv2.bias'], padding=1)
    x = conv_transpose2d_k2s2(x, p[prefix + '.upsample2.weight'], p[prefix + '.upsample2.bias'])
    return conv_transpose2d_k4s4(x, p[prefix + '.upsample3.weight'], p[prefix + '.upsample3.bias'])


def nerf_alpha(feat, p):
    """:1096-1102: sigma = Softplus(Linear(Linear(feat))) (no activation in between, :605);
    alpha = 1 - exp(-sigma).  feat (B,80,H,W) -> (B,H,W)."""
    f = feat.transpose(0, 2, 3, 1)
    s = softplus(linear(linear(f, p, 'sigma.0'), p, 'sigma.1'))
    return (1.0 - np.exp(-s.astype(np.float64)))[..., 0].astype(f32)


def nerf_render(feat, alpha, sparse_rgb, p):
    """:1104-1121 for the selected camera.  feat (80,H,W), alpha (H,W), sparse_rgb (3,H,W) in 0..255
    -> render_image_N (3,H,W), render_depth_N (1,H,W).  ``T`` is identically 1 (cumprod over a
    size-1 dim) and the depth weight is a softmax over a size-1 dim, i.e. 1."""
    x = np.concatenate((feat.transpose(1, 2, 0), sparse_rgb.transpose(1, 2, 0) / f32(255.0)), -1)
    cw = softmax(sigmoid(linear(relu(linear(x, p, 'C_MLP_nerf.fc1')), p, 'C_MLP_nerf.fc2')), -1)
    rad = relu(linear(relu(linear(x, p, 'img_feat_resize1.fc1')), p, 'img_feat_resize1.fc2')) * cw
    rad1 = relu(linear(relu(linear(x, p, 'img_feat_resize2.fc1')), p, 'img_feat_resize2.fc2'))
    w = alpha[..., None]
    return (w * rad).transpose(2, 0, 1).astype(f32), (w * rad1).transpose(2, 0, 1).astype(f32)


# ---------------------------------------------------------------------------------------- a27
def ms_cam(x, p, prefix):
    """MS_CAM :36-66 (eval)."""
    def branch(t, name):
        t = conv2d(t, p[f'{name}.0.weight'], p[f'{name}.0.bias'])
        t = relu(batchnorm_eval(t, p, f'{name}.1'))
        t = conv2d(t, p[f'{name}.3.weight'], p[f'{name}.3.bias'])
        return batchnorm_eval(t, p, f'{name}.4')
    xl = branch(x, prefix + '.local_att')
    pooled = x.astype(np.float64).mean((2, 3), keepdims=True).astype(f32)
    # global_att = [AdaptiveAvgPool2d, conv, bn, relu, conv, bn]: indices shift by one
    t = conv2d(pooled, p[f'{prefix}.global_att.1.weight'], p[f'{prefix}.global_att.1.bias'])
    t = relu(batchnorm_eval(t, p, f'{prefix}.global_att.2'))
    t = conv2d(t, p[f'{prefix}.global_att.4.weight'], p[f'{prefix}.global_att.4.bias'])
    xg = batchnorm_eval(t, p, f'{prefix}.global_att.5')
    return sigmoid(xl + xg)


def dual_feat_fusion(x1, x2, p, prefix='fuser'):
    """DualFeatFusion :203-213."""
    cf = ms_cam(np.concatenate((x1, x2), 1), p, prefix + '.ca')
    return (cf * x1 + (1 - cf) * x2).astype(f32)


def _channel_attention(x, p, prefix):
    def fc(t):
        return conv2d(relu(conv2d(t, p[prefix + '.fc.0.weight'])), p[prefix + '.fc.2.weight'])
    avg = x.astype(np.float64).mean((2, 3), keepdims=True).astype(f32)
    mx = x.max((2, 3), keepdims=True)
    return sigmoid(fc(avg) + fc(mx))


def _spatial_logits(x, w):
    s = np.concatenate((x.astype(np.float64).mean(1, keepdims=True).astype(f32), x.max(1, keepdims=True)), 1)
    return conv2d(s, w, padding=w.shape[-1] // 2)


def prob_net(x, p, prefix='prob'):
    """ProbNet.forward :176-180 with ResCBAMBlock :100-137 (eval)."""
    h = relu(batchnorm_eval(conv2d(x, p[prefix + '.base_conv.0.weight'], p[prefix + '.base_conv.0.bias'], padding=1),
                            p, prefix + '.base_conv.1'))
    blk = prefix + '.prob_conv.0'
    out = relu(batchnorm_eval(conv2d(h, p[blk + '.conv1.weight'], padding=1), p, blk + '.bn1'))
    out = batchnorm_eval(conv2d(out, p[blk + '.conv2.weight'], padding=1), p, blk + '.bn2')
    out = _channel_attention(out, p, blk + '.ca') * out
    out = sigmoid(_spatial_logits(out, p[blk + '.sa.conv1.weight'])) * out
    out = relu(out + h)
    return conv2d(out, p[prefix + '.mask_net.weight'], p[prefix + '.mask_net.bias'])


def bev_geom_attention(x, bev_prob, p, prefix='geom_att'):
    """BEVGeomAttention :215-228."""
    return sigmoid(_spatial_logits(x, p[prefix + '.conv1.weight']) + bev_prob)


def learned_positional_encoding(p, prefix, B, H, W):
    """mmdet ``LearnedPositionalEncoding`` (not in the reference tree; mmdet 2.x behaviour, SURVEY
    A.6): cat(col_embed(x) over rows, row_embed(y) over columns) -> (B, 2F, H, W)."""
    col, row = p[prefix + '.col_embed.weight'][:W], p[prefix + '.row_embed.weight'][:H]
    pos = np.concatenate((np.broadcast_to(col[None], (H, W, col.shape[1])),
                          np.broadcast_to(row[:, None], (H, W, row.shape[1]))), -1)
    return np.broadcast_to(pos.transpose(2, 0, 1)[None], (B,) + (pos.shape[2], H, W)).astype(f32)
