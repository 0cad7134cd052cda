"""Python boundary of ``csrc/neck.hip``: the stages of ``view_transform_core`` between the poolings,
the render and HOA, eval mode.

Each function takes / returns CUDA tensors, calls the C ABI (``include/ocrf_hip.h``) on torch's
current stream and raises ``OcrfHipError`` on CPU tensors — there is no fallback.  The ``pack_*`` /
``compose_*`` helpers turn the reference's ``state_dict`` layout into the flat parameter blocks the
kernels read; they are pure torch and run anywhere.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib

__all__ = ['prefilter', 'pillar_sample_mean', 'retain_valid_pixels', 'gauss_heads', 'pack_gauss_head_params',
           'lift_coefficients', 'pack_gauss_head_params_autograd', 'gauss_heads_train', 'gauss_heads_backward',
           'TallLinear', 'tall_linear',
           'compose_nerf_maps', 'nerf_alpha', 'nerf_render', 'pack_fusion_params', 'dual_feat_fusion',
           'plane_bias_act_stats', 'channel_mlp', 'scaled_channel_stats', 'cbam_tail', 'pack_global_att',
           'pack_probnet', 'probnet_forward', 'global_att_vector']


def _f32c(t):
    return t.contiguous().float()


def prefilter(x, D, C, depth_threshold, semantic_threshold):
    """view_transformer_ocrf.py:1323-1331.  x (BN, D+2+C, H, W) ->
    depth (BN,D,H,W), filter_depth (BN,D,H,W), semantic (BN,2,H,W), filter_feat channels-last
    (BN,H,W,C) (the operand the poolings permute to, :875/:901)."""
    _lib.require_cuda(x)
    x = _f32c(x)
    BN, ch, H, W = x.shape
    if ch < D + 2 + C:
        raise ValueError(f'prefilter: {ch} channels < D + 2 + C = {D + 2 + C}')
    if ch != D + 2 + C:
        x = x[:, :D + 2 + C].contiguous()
    dev = x.device
    depth = torch.empty(BN, D, H, W, device=dev)
    fdepth = torch.empty_like(depth)
    sem = torch.empty(BN, 2, H, W, device=dev)
    feat = torch.empty(BN, H, W, C, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ocrf_prefilter(_lib.ptr(x), BN, D, C, H * W, ctypes.c_float(depth_threshold),
                                             ctypes.c_float(semantic_threshold), _lib.ptr(depth), _lib.ptr(fdepth),
                                             _lib.ptr(sem), _lib.ptr(feat), _lib.stream_ptr(dev)), 'ocrf_prefilter')
    return depth, fdepth, sem, feat


def _mask_bytes(mask, shape):
    m = mask.reshape(shape)
    if m.dtype != torch.bool and m.dtype != torch.uint8:
        m = m != 0
    return m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.contiguous()


def pillar_sample_mean(imgs, pix, mask, view_hw=None):
    """``lidar_points_to_image_values`` + ``color_voxels``'s ``avg_color``
    (view_transformer_ocrf.py:924-959).  imgs (B,N,C,H,W), pix (B,N,Zh,Q,2) pixel coordinates, mask
    (B,N,Zh,Q[,1]) -> (B,Zh,Q,C).  ``view_hw``: the (H,W) the caller's *view* of the image memory
    claims, for the reference's swapped view of the alpha volume (:1123)."""
    _lib.require_cuda(imgs, pix, mask)
    B, N, C, H, W = imgs.shape
    if view_hw is not None:
        if view_hw[0] * view_hw[1] != H * W:
            raise ValueError('view_hw must cover the same memory as the image')
        H, W = int(view_hw[0]), int(view_hw[1])
    Zh, Q = pix.shape[2], pix.shape[3]
    imgs, pix = _f32c(imgs), _f32c(pix)
    m = _mask_bytes(mask, (B, N, Zh * Q))
    avg = torch.empty(B, Zh, Q, C, device=imgs.device)
    with _lib.on_device(imgs.device):
        _lib.check(_lib.lib().ocrf_pillar_sample_mean(_lib.ptr(imgs), _lib.ptr(pix), _lib.ptr(m), _lib.ptr(avg), B, N, C,
                                                      H, W, Zh * Q, _lib.stream_ptr(imgs.device)),
                   'ocrf_pillar_sample_mean')
    return avg


def retain_valid_pixels(image_matrix, pseudo_point_cloud, mask, cam_sel=None):
    """``retain_valid_pixels`` (view_transformer_ocrf.py:1004-1024).  image_matrix (B,N,C,H,W),
    pseudo_point_cloud (B,N,Zh,...,2) pixel coordinates, mask (B,N,Zh,...[,1]).
    ``cam_sel`` None -> (B,N,C,H,W) like the reference; an int32 device vector (B) -> only camera
    ``cam_sel[b]`` of each sample, (B,C,H,W) (the reference consumes just that one, :1104)."""
    _lib.require_cuda(image_matrix, pseudo_point_cloud, mask)
    B, N, C, H, W = image_matrix.shape
    imgs = _f32c(image_matrix)
    pix = _f32c(pseudo_point_cloud).reshape(B, N, -1, 2)
    ZQ = pix.shape[2]
    m = _mask_bytes(mask, (B, N, ZQ))
    dev = imgs.device
    if cam_sel is not None:
        _lib.require_cuda(cam_sel)
        cam_sel = cam_sel.to(torch.int32).contiguous()
        out = torch.empty(B, C, H, W, device=dev)
    else:
        out = torch.empty(B, N, C, H, W, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ocrf_retain_valid_pixels(_lib.ptr(imgs), _lib.ptr(pix), _lib.ptr(m), _lib.ptr(cam_sel),
                                                       _lib.ptr(out), B, N, C, H, W, ZQ, _lib.stream_ptr(dev)),
                   'ocrf_retain_valid_pixels')
    return out


def pack_gauss_head_params(vfe, s_mlp, r_mlp, a_mlp, c_mlp):
    """Flat parameter block of ``ocrf_gauss_heads`` (layout: csrc/neck.hip) from the reference's
    modules: ``VoxelFeatureExtractor`` with its BatchNorm3d folded (eval statistics) and the four
    heads' ``fc1`` / ``fc2``."""
    conv, bn = vfe.conv[0], vfe.conv[1]
    with torch.no_grad():
        s = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
        a = conv.weight.double().reshape(-1) * s
        b = (conv.bias.double() - bn.running_mean.double()) * s + bn.bias.double()
        C = s_mlp.fc1.in_features
        w1 = torch.cat((s_mlp.fc1.weight, r_mlp.fc1.weight, a_mlp.fc1.weight, c_mlp.fc1.weight[:, :C]), 0)
        b1 = torch.cat((s_mlp.fc1.bias, r_mlp.fc1.bias, a_mlp.fc1.bias, c_mlp.fc1.bias))
        # first layer stored channel-major (C,16): one 64-byte scalar load per channel in the kernel
        parts = [a.float(), b.float(), w1.t().reshape(-1), c_mlp.fc1.weight[:, C:].reshape(-1), b1]
        for m in (s_mlp, r_mlp, a_mlp, c_mlp):
            parts += [m.fc2.weight.reshape(-1), m.fc2.bias]
        return torch.cat([p.detach().float().reshape(-1) for p in parts]).contiguous()


def gauss_heads(bev, rgb_avg, params, num_height):
    """VoxelFeatureExtractor + S/R/A/C_MLP (view_transformer_ocrf.py:1051, :1130-1133) without the
    (B,13,Y,X,80) voxel feature.  bev (B,C,Y,X), rgb_avg (B,Zh,Y*X,3) in 0..255 ->
    opacity (B,P,1), scales (B,P,3), rotations (B,P,4), color (B,P,3), P = Zh*Y*X."""
    _lib.require_cuda(bev, rgb_avg, params)
    B, C, Y, X = bev.shape
    bev, rgb_avg = _f32c(bev), _f32c(rgb_avg)
    L = _lib.lib()
    if params.numel() != L.ocrf_gauss_heads_params_len(C, num_height):
        raise ValueError('gauss_heads: parameter block does not match (C, num_height)')
    P = num_height * Y * X
    dev = bev.device
    op, sc = torch.empty(B, P, 1, device=dev), torch.empty(B, P, 3, device=dev)
    rot, col = torch.empty(B, P, 4, device=dev), torch.empty(B, P, 3, device=dev)
    with _lib.on_device(dev):
        _lib.check(L.ocrf_gauss_heads(_lib.ptr(bev), _lib.ptr(rgb_avg), _lib.ptr(params), B, C, num_height, Y * X,
                                      _lib.ptr(op), _lib.ptr(sc), _lib.ptr(rot), _lib.ptr(col), _lib.stream_ptr(dev)),
                   'ocrf_gauss_heads')
    return op, sc, rot, col


def lift_coefficients(vfe, x=None):
    """``VoxelFeatureExtractor`` (view_transformer_ocrf.py:520-531) as ``relu(a_h x + b_h)`` per height h, WITH autograd:
    channel h of the 1-channel ``Conv3d(k = 1)`` is ``w_h x + c_h``, so BatchNorm3d's statistics of it are
    ``w_h mean(x) + c_h`` and ``w_h^2 var(x)``.  ``x`` given and the BatchNorm in training mode: the batch statistics of
    ``x`` (differentiated through, and the running statistics take their momentum update); otherwise the running ones."""
    conv, bn = vfe.conv[0], vfe.conv[1]
    w = conv.weight.reshape(-1)
    c = conv.bias if conv.bias is not None else torch.zeros_like(w)
    if bn.training and x is not None:
        n = x.numel()
        var, mean = torch.var_mean(x.float(), unbiased=False)
        mean_h, var_h = w * mean + c, w * w * var
        with torch.no_grad():
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - mom).add_(mean_h, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var_h * (n / max(n - 1, 1)), alpha=mom)
    else:
        mean_h, var_h = bn.running_mean, bn.running_var
    s = bn.weight * torch.rsqrt(var_h + bn.eps)
    return s * w, s * (c - mean_h) + bn.bias


def pack_gauss_head_params_autograd(vfe, s_mlp, r_mlp, a_mlp, c_mlp, x=None):
    """``pack_gauss_head_params`` as differentiable torch ops (fp32): the gradient ``gauss_heads_train`` returns for the
    packed block flows on to the modules' parameters — and, through the batch statistics, to ``x``."""
    a, b = lift_coefficients(vfe, x)
    C = s_mlp.fc1.in_features
    w1 = torch.cat((s_mlp.fc1.weight, r_mlp.fc1.weight, a_mlp.fc1.weight, c_mlp.fc1.weight[:, :C]), 0)
    parts = [a, b, w1.t().reshape(-1), c_mlp.fc1.weight[:, C:].reshape(-1),
             s_mlp.fc1.bias, r_mlp.fc1.bias, a_mlp.fc1.bias, c_mlp.fc1.bias]
    for m in (s_mlp, r_mlp, a_mlp, c_mlp):
        parts += [m.fc2.weight.reshape(-1), m.fc2.bias]
    return torch.cat([p.float().reshape(-1) for p in parts])


class _GaussHeadsTrain(torch.autograd.Function):
    """``ocrf_gauss_heads`` forward, ``ocrf_gauss_heads_backward`` backward: nothing is kept between the two but the inputs
    (the (B,Zh,Y,X,C) voxel feature of the reference, 333 MB at cfg2, exists in neither)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, bev, rgb_avg, params, num_height):
        bev, rgb_avg, params = _f32c(bev), _f32c(rgb_avg), _f32c(params)
        ctx.save_for_backward(bev, rgb_avg, params)
        ctx.num_height = num_height
        return gauss_heads(bev, rgb_avg, params, num_height)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g_op, g_sc, g_rot, g_col):
        bev, rgb_avg, params = ctx.saved_tensors
        d_bev, d_params = gauss_heads_backward(bev, rgb_avg, params, ctx.num_height, g_op, g_sc, g_rot, g_col)
        return d_bev, None, d_params, None


def gauss_heads_train(bev, rgb_avg, params, num_height):
    """``gauss_heads`` under autograd (training mode of view_transformer_ocrf.py:1051, :1130-1133): gradients for ``bev``
    and for the packed ``params`` (``pack_gauss_head_params_autograd``); none for ``rgb_avg``, which is sampled from the
    camera images."""
    return _GaussHeadsTrain.apply(bev, rgb_avg, params, num_height)


def gauss_heads_backward(bev, rgb_avg, params, num_height, g_opacity=None, g_scales=None, g_rotations=None, g_color=None):
    """-> (d_bev (B,C,Y,X), d_params (like params)) from the gradients of the four outputs (None = zero)."""
    _lib.require_cuda(bev, rgb_avg, params)
    B, C, Y, X = bev.shape
    L = _lib.lib()
    if params.numel() != L.ocrf_gauss_heads_params_len(C, num_height):
        raise ValueError('gauss_heads_backward: parameter block does not match (C, num_height)')
    dev, P = bev.device, num_height * Y * X
    grads = []
    for g, k in ((g_opacity, 1), (g_scales, 3), (g_rotations, 4), (g_color, 3)):
        if g is not None:
            if g.numel() != B * P * k:
                raise ValueError('gauss_heads_backward: an output gradient does not match its output')
            g = _f32c(g)
        grads.append(g)
    need = L.ocrf_gauss_heads_backward_workspace_bytes(B, C, num_height, Y * X)
    if not need:
        raise ValueError(f'gauss_heads_backward: num_height {num_height} has no register tile')
    scratch = _lib.workspace.get(dev, need, 'gauss_heads_backward')
    d_bev, d_params = torch.empty_like(bev), torch.empty_like(params)
    with _lib.on_device(dev):
        _lib.check(L.ocrf_gauss_heads_backward(
            _lib.ptr(bev), _lib.ptr(rgb_avg), _lib.ptr(params), B, C, num_height, Y * X,
            *[_lib.ptr(g) if g is not None else None for g in grads], _lib.ptr(d_bev), _lib.ptr(d_params),
            _lib.ptr(scratch), scratch.numel(), _lib.stream_ptr(dev)), 'ocrf_gauss_heads_backward')
    return d_bev, d_params


class TallLinear(torch.autograd.Function):
    """``F.linear`` whose weight gradient is a split-K batched GEMM.  The heads run on 10^6 rows with 4..83
    columns; hipBLASLt computes ``dW = dY^T X`` (K = rows) as ONE tall reduction at 1-5 ms per layer (20 ms of an
    84 ms neck forward + backward).  Chunks of 4096 rows through ``bmm`` + a sum are ~30x faster."""
    CHUNK = 4096

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda')
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = g.matmul(weight)
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        if ctx.needs_input_grad[1]:
            r = TallLinear.CHUNK
            n = g2.shape[0] // r
            gw = torch.bmm(g2[:n * r].view(n, r, -1).transpose(1, 2), x2[:n * r].view(n, r, -1)).sum(0)
            if n * r < g2.shape[0]:
                gw = gw + g2[n * r:].t().mm(x2[n * r:])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


def tall_linear(lin, x):
    """``lin(x)`` for an ``nn.Linear``; many-row inputs under autograd take the split-K weight gradient."""
    if x.is_cuda and torch.is_grad_enabled() and x.numel() // x.shape[-1] >= 8 * TallLinear.CHUNK:
        return TallLinear.apply(x, lin.weight, lin.bias)
    return lin(x)


def compose_nerf_maps(resize, sigma, c_mlp_nerf, img_feat_resize1, img_feat_resize2):
    """ResizeNetwork has no non-linearity (view_transformer_ocrf.py:534-554), so for every consumer
    whose first layer is a Linear ``L`` (weights w (k,80), bias c (k)):
        L(upsample3(upsample2(z)))[y, x] = z[:, y//8, x//8] . M[:, pos] + const[pos],  pos = (y%8)*8 + x%8
    with M (k,32,64).  Composed in float64.  -> (w_sigma (32,64), c_sigma (64),
    render parameter block of ``ocrf_nerf_render``) as float32 tensors on the modules' device."""
    with torch.no_grad():
        W2, b2 = resize.upsample2.weight.double(), resize.upsample2.bias.double()      # (32,80,2,2), (80)
        W3, b3 = resize.upsample3.weight.double(), resize.upsample3.bias.double()      # (80,80,4,4), (80)

        def compose(w, c):
            """w (k,80), c (k) -> M (k,32,8,8) flattened to (k,32,64), const (k,64)."""
            t = torch.einsum('ofrs,kf->kors', W3, w)                 # (k, 80 mid channels, 4, 4)
            M = torch.einsum('iopq,kors->kiprqs', W2, t)             # y%8 = 4p + r, x%8 = 4q + s
            const = torch.einsum('o,kors->krs', b2, t) + (w @ b3 + c)[:, None, None]    # (k,4,4)
            const = const[:, None, :, None, :].expand(-1, 2, -1, 2, -1)                 # (k,p,r,q,s)
            k = w.shape[0]
            return M.reshape(k, 32, 64), const.reshape(k, 64)
        ws = sigma[1].weight.double() @ sigma[0].weight.double()                         # (1,80)
        cs = sigma[1].weight.double() @ sigma[0].bias.double() + sigma[1].bias.double()
        Ms, consts = compose(ws, cs)
        heads = (c_mlp_nerf, img_feat_resize1, img_feat_resize2)
        w1 = torch.cat([h.fc1.weight.double()[:, :80] for h in heads], 0)               # (12,80)
        b1 = torch.cat([h.fc1.bias.double() for h in heads])
        M12, c12 = compose(w1, b1)
        rgb = torch.cat([h.fc1.weight.double()[:, 80:] for h in heads], 0)              # (12,3)
        parts = [M12, c12, rgb]
        for h in heads:
            parts += [h.fc2.weight.double(), h.fc2.bias.double()]
        block = torch.cat([p.reshape(-1) for p in parts]).float().contiguous()
        return Ms[0].float().contiguous(), consts[0].float().contiguous(), block


def nerf_alpha(z, w_sigma, c_sigma):
    """alpha = 1 - exp(-softplus(sigma(feat))) for every camera image (view_transformer_ocrf.py:
    1096-1102) from z (M,32,h2,w2) = ResizeNetwork.conv2's output.  -> (M, 8*h2, 8*w2)."""
    _lib.require_cuda(z, w_sigma, c_sigma)
    z = _f32c(z)
    M, ci, h2, w2 = z.shape
    if ci != 32 or w_sigma.numel() != 32 * 64 or c_sigma.numel() != 64:
        raise ValueError('nerf_alpha: z must have 32 channels and the maps 32x64 / 64 entries')
    alpha = torch.empty(M, 8 * h2, 8 * w2, device=z.device)
    with _lib.on_device(z.device):
        _lib.check(_lib.lib().ocrf_nerf_alpha(_lib.ptr(z), _lib.ptr(w_sigma), _lib.ptr(c_sigma), _lib.ptr(alpha), M,
                                              h2, w2, _lib.stream_ptr(z.device)), 'ocrf_nerf_alpha')
    return alpha


def nerf_render(z, cam_sel, alpha, sparse_rgb, params, n_cams):
    """NeRF-branch image / depth of the selected camera of each sample (view_transformer_ocrf.py:
    1104-1121).  z (B*N,32,h2,w2), cam_sel (B) int32, alpha (B*N,H,W), sparse_rgb (B,3,H,W) in
    0..255 -> render_image_N (B,3,H,W), render_depth_N (B,1,H,W)."""
    _lib.require_cuda(z, cam_sel, alpha, sparse_rgb, params)
    z, alpha, sparse_rgb = _f32c(z), _f32c(alpha), _f32c(sparse_rgb)
    M, _, h2, w2 = z.shape
    B = M // n_cams
    L = _lib.lib()
    if params.numel() != L.ocrf_nerf_render_params_len() or tuple(sparse_rgb.shape) != (B, 3, 8 * h2, 8 * w2):
        raise ValueError('nerf_render: parameter block or sparse_rgb shape mismatch')
    cam_sel = cam_sel.to(torch.int32).contiguous()
    img = torch.empty(B, 3, 8 * h2, 8 * w2, device=z.device)
    dep = torch.empty(B, 1, 8 * h2, 8 * w2, device=z.device)
    with _lib.on_device(z.device):
        _lib.check(L.ocrf_nerf_render(_lib.ptr(z), _lib.ptr(cam_sel), _lib.ptr(alpha), _lib.ptr(sparse_rgb),
                                      _lib.ptr(params), B, n_cams, h2, w2, _lib.ptr(img), _lib.ptr(dep),
                                      _lib.stream_ptr(z.device)), 'ocrf_nerf_render')
    return img, dep


def _fold_conv_bn(conv, bn):
    """1x1 conv + eval BatchNorm -> (W (out,in), b (out)) in float64."""
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    w = conv.weight.double().reshape(conv.out_channels, conv.in_channels) * s[:, None]
    b = (conv.bias.double() - bn.running_mean.double()) * s + bn.bias.double()
    return w, b


def _fold_conv_bn_kxk(conv, bn):
    """k x k conv (bias optional) + eval BatchNorm -> (W (out,in,k,k) f32, b (out) f32), folded in float64."""
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    w = conv.weight.double() * s[:, None, None, None]
    cb = conv.bias.double() if conv.bias is not None else torch.zeros_like(s)
    b = (cb - bn.running_mean.double()) * s + bn.bias.double()
    return w.float().contiguous(), b.float().contiguous()


def pack_fusion_params(ms_cam):
    """Parameter block of ``ocrf_dual_feat_fusion`` from an ``MS_CAM``'s ``local_att`` (conv, bn, relu,
    conv, bn; view_transformer_ocrf.py:42-48): W1t[2C][M] | b1[M] | W2[C][M] | b2[C]."""
    with torch.no_grad():
        la = ms_cam.local_att
        w1, b1 = _fold_conv_bn(la[0], la[1])          # (M, 2C)
        w2, b2 = _fold_conv_bn(la[3], la[4])          # (C, M)
        return torch.cat((w1.t().reshape(-1), b1, w2.reshape(-1), b2)).float().contiguous()


def dual_feat_fusion(x1, x2, params, global_vec, hidden, addend=None):
    """``DualFeatFusion.forward`` (view_transformer_ocrf.py:203-213), eval mode, in one pass.
    x1, x2 (B,C,Y,X); ``global_vec`` (B,C) = MS_CAM's global branch; -> (B,C,Y,X).  With ``addend`` (B,C,Y,X):
    -> (out, addend + out), both written by the same pass."""
    _lib.require_cuda(x1, x2, params, global_vec)
    B, C, Y, X = x1.shape
    x1, x2, gv = _f32c(x1), _f32c(x2), _f32c(global_vec).reshape(B, C)
    if params.numel() != 2 * C * hidden + hidden + C * hidden + C:
        raise ValueError('dual_feat_fusion: parameter block does not match (C, hidden)')
    out = torch.empty_like(x1)
    with _lib.on_device(x1.device):
        if addend is not None:
            if tuple(addend.shape) != tuple(x1.shape):
                raise ValueError('dual_feat_fusion: addend must have the shape of the inputs')
            add, plus = _f32c(addend), torch.empty_like(x1)
            _lib.check(_lib.lib().ocrf_dual_feat_fusion_plus(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(params), _lib.ptr(gv),
                                                             _lib.ptr(out), _lib.ptr(add), _lib.ptr(plus), B, C, hidden,
                                                             Y * X, _lib.stream_ptr(x1.device)),
                       'ocrf_dual_feat_fusion_plus')
            return out, plus
        _lib.check(_lib.lib().ocrf_dual_feat_fusion(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(params), _lib.ptr(gv),
                                                    _lib.ptr(out), B, C, hidden, Y * X, _lib.stream_ptr(x1.device)),
                   'ocrf_dual_feat_fusion')
    return out


# ------------------------------------------------------------------------------------------------
# CBAM / ProbNet tail (ocrf_plane_bias_act_stats, ocrf_channel_mlp, ocrf_scaled_channel_stats, ocrf_cbam_tail)
# ------------------------------------------------------------------------------------------------
_SPLITS = 8


def plane_bias_act_stats(y, bias=None, relu=False, write=True, stats=None, c_off=0):
    """In place on y (B,C,Y,X): ``y += bias[c]`` then ReLU (``write``); with ``stats=(psum, pmax)`` (each
    (B, out_C, S)) also the S partial sums / maxima of every plane at channel offset ``c_off``."""
    _lib.require_cuda(y)
    if not (y.is_contiguous() and y.dtype == torch.float32):
        raise ValueError('plane_bias_act_stats works in place on a contiguous float32 tensor')
    B, C, Y, X = y.shape
    psum, pmax = stats if stats is not None else (None, None)
    out_C, S = (psum.shape[1], psum.shape[2]) if psum is not None else (C, _SPLITS)
    with _lib.on_device(y.device):
        _lib.check(_lib.lib().ocrf_plane_bias_act_stats(_lib.ptr(y), _lib.ptr(bias), B, C, Y * X, int(relu), int(write),
                                                        S, out_C, c_off, _lib.ptr(psum), _lib.ptr(pmax),
                                                        _lib.stream_ptr(y.device)), 'ocrf_plane_bias_act_stats')
    return y


def channel_mlp(psum, pmax, inv_n, w1, b1, w2, b2, use_max, sigmoid):
    """Pooled vectors (from the partials) through W2.relu(W1.v + b1) + b2, summed over the mean and max vectors
    when ``use_max``; -> (B, N)."""
    _lib.require_cuda(psum, w1, w2)
    B, K, S = psum.shape
    M, N = w1.shape[0], w2.shape[0]
    out = torch.empty(B, N, device=psum.device)
    with _lib.on_device(psum.device):
        _lib.check(_lib.lib().ocrf_channel_mlp(_lib.ptr(psum), _lib.ptr(pmax), B, K, S, ctypes.c_float(inv_n),
                                               _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), M, N,
                                               int(use_max), int(sigmoid), _lib.ptr(out),
                                               _lib.stream_ptr(psum.device)), 'ocrf_channel_mlp')
    return out


def scaled_channel_stats(x, scale):
    """(B,2,Y,X): mean and max over channels of ``scale[b,c] * x[b,c]``."""
    _lib.require_cuda(x, scale)
    B, C, Y, X = x.shape
    stats = torch.empty(B, 2, Y, X, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().ocrf_scaled_channel_stats(_lib.ptr(x), _lib.ptr(scale), B, C, Y * X, _lib.ptr(stats),
                                                        _lib.stream_ptr(x.device)), 'ocrf_scaled_channel_stats')
    return stats


def cbam_tail(y, scale, stats, sa_weight, res, wm, bm, want_block_out=False):
    """ResCBAMBlock's tail + a 1x1 head: -> logit (B,1,Y,X) [, block output (B,C,Y,X)]."""
    _lib.require_cuda(y, scale, stats, sa_weight, res, wm)
    B, C, Y, X = y.shape
    k = sa_weight.shape[-1]
    logit = torch.empty(B, 1, Y, X, device=y.device)
    block = torch.empty_like(y) if want_block_out else None
    with _lib.on_device(y.device):
        _lib.check(_lib.lib().ocrf_cbam_tail(_lib.ptr(y), _lib.ptr(scale), _lib.ptr(stats), _lib.ptr(sa_weight), k,
                                             _lib.ptr(res), _lib.ptr(wm), ctypes.c_float(bm), B, C, Y, X,
                                             _lib.ptr(logit), _lib.ptr(block), _lib.stream_ptr(y.device)),
                   'ocrf_cbam_tail')
    return (logit, block) if want_block_out else logit


def pack_global_att(ms_cam):
    """MS_CAM.global_att (pool, conv, bn, relu, conv, bn; :50-58) with both BatchNorms folded."""
    with torch.no_grad():
        ga = ms_cam.global_att
        w1, b1 = _fold_conv_bn(ga[1], ga[2])
        w2, b2 = _fold_conv_bn(ga[4], ga[5])
        return tuple(t.float().contiguous() for t in (w1, b1, w2, b2))


def global_att_vector(x1, x2, packed):
    """MS_CAM's global branch of cat(x1, x2): (B, C) — one read of each input + one tiny launch instead of
    two means, a cat, two 1x1 convolutions with bias, two BatchNorms and a ReLU."""
    _lib.require_cuda(x1, x2)
    B, C, Y, X = x1.shape
    x1, x2 = _f32c(x1), _f32c(x2)
    psum = torch.empty(B, 2 * C, _SPLITS, device=x1.device)
    pmax = torch.empty_like(psum)
    with _lib.on_device(x1.device):          # both inputs in ONE launch (no cat: the kernel reads the second tensor itself)
        _lib.check(_lib.lib().ocrf_plane_stats_pair(_lib.ptr(x1), _lib.ptr(x2), B, C, C, Y * X, _SPLITS, _lib.ptr(psum),
                                                    _lib.ptr(pmax), _lib.stream_ptr(x1.device)), 'ocrf_plane_stats_pair')
    w1, b1, w2, b2 = packed
    return channel_mlp(psum, None, 1.0 / (Y * X), w1, b1, w2, b2, use_max=False, sigmoid=False)


def pack_probnet(prob):
    """Everything ``probnet_forward`` needs from a ``ProbNet`` (:139-201) whose ``prob_conv`` is one
    ``ResCBAMBlock`` without downsample: the three 3x3 convolutions with their BatchNorms folded, the
    channel-attention MLP, the spatial-attention kernel and the 1x1 mask head."""
    with torch.no_grad():
        blk = prob.prob_conv[0]
        w0, b0 = _fold_conv_bn_kxk(prob.base_conv[0], prob.base_conv[1])
        w1, b1 = _fold_conv_bn_kxk(blk.conv1, blk.bn1)
        w2, b2 = _fold_conv_bn_kxk(blk.conv2, blk.bn2)
        fc = blk.ca.fc
        return {'w0': w0, 'b0': b0, 'w1': w1, 'b1': b1, 'w2': w2, 'b2': b2,
                'ca1': fc[0].weight.float().reshape(fc[0].out_channels, -1).contiguous(),
                'ca2': fc[2].weight.float().reshape(fc[2].out_channels, -1).contiguous(),
                'sa': blk.sa.conv1.weight.float().reshape(-1).contiguous(), 'k': blk.sa.conv1.kernel_size[0],
                'wm': prob.mask_net.weight.float().reshape(-1).contiguous(), 'bm': float(prob.mask_net.bias.item())}


def probnet_forward(x, pk):
    """``ProbNet.forward`` (:171-172 of the module here; reference :195-201), eval mode: three MIOpen
    convolutions + six HIP launches."""
    import torch.nn.functional as F
    _lib.require_cuda(x)
    B, _, Y, X = x.shape
    y0 = plane_bias_act_stats(F.conv2d(_f32c(x), pk['w0'], None, padding=1).contiguous(), pk['b0'], relu=True)
    y1 = plane_bias_act_stats(F.conv2d(y0, pk['w1'], None, padding=1).contiguous(), pk['b1'], relu=True)
    y2 = F.conv2d(y1, pk['w2'], None, padding=1).contiguous()
    C = y2.shape[1]
    psum = torch.empty(B, C, _SPLITS, device=x.device)
    pmax = torch.empty_like(psum)
    plane_bias_act_stats(y2, pk['b2'], relu=False, stats=(psum, pmax))
    scale = channel_mlp(psum, pmax, 1.0 / (Y * X), pk['ca1'], None, pk['ca2'], None, use_max=True, sigmoid=True)
    stats = scaled_channel_stats(y2, scale)
    sa = pk['sa'].view(1, 2, pk['k'], pk['k'])
    return cbam_tail(y2, scale, stats, sa, y0, pk['wm'], pk['bm'])
