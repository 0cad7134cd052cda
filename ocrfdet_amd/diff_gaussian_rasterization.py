"""Gaussian rasteriser with depth output — drop-in for the ``diff_gaussian_rasterization`` module
OcRFDet imports (the JonathonLuiten *w-depth* fork; call site
``mmdet3d/models/necks/MVSGaussian/lib/gaussian_renderer/__init__.py:14,39-70``).

Surface kept (names, argument order, return arity of the fork):

``GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
viewmatrix, projmatrix, sh_degree, campos, prefiltered[, debug])``
    the fork's 11 fields; ``debug`` (the stock 12th field,
    ``.../diff_gaussian_rasterization/__init__.py:157-169``) is accepted and defaults to False.
``GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
scales=None, rotations=None, cov3D_precomp=None) -> (color (3,H,W), radii (P,) int32,
depth (1,H,W))``
    exactly one of shs / colors_precomp and one of (scales, rotations) / cov3D_precomp, else the
    reference's ``Exception`` (``__init__.py:191-195``).  ``viewmatrix`` / ``projmatrix`` are the
    transposed (row-vector) 4x4 matrices.
``rasterize_gaussians(...)`` and ``GaussianRasterizer.markVisible``.

Beyond the reference: ``rasterize_views`` renders a batch of cameras over one Gaussian set in a
single launch sequence and also returns ``final_T`` (1 - accumulated opacity) and ``n_contrib``.

Compute: ``csrc/rasterize.hip`` through the C ABI (``ocrf_rasterize_forward`` /
``ocrf_rasterize_backward``).  The backward covers what the fork's does — the colour output w.r.t.
means3D, means2D (screen-space, for densification statistics), colours, opacities, scales, rotations;
depth has no backward in the fork either (diff-gaussian-rasterization-w-depth/README.md:13).
``cov3D_precomp`` has its own backward (gradient w.r.t. the six covariance entries).  SH colours (``shs`` (P,M,3) with
``raster_settings.sh_degree`` / ``campos``; OcRFDet itself passes ``shs=None``) are evaluated by ``ocrf_sh_to_rgb`` into the
(P,3) colours of the ordinary pipeline, forward and backward (``forward.cu:20-71``, ``backward.cu:20-140``).
"""
import ctypes
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib

__all__ = ['GaussianRasterizationSettings', 'GaussianRasterizer', 'rasterize_gaussians', 'rasterize_views', 'pack_cameras',
           'rasterize_views_backward', 'rasterize_views_autograd', 'rasterize_packed_autograd', 'sh_to_rgb',
           'sh_to_rgb_backward']


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool = False


def _f32c(t):
    return t.detach().contiguous().float()


def pack_cameras(viewmatrices, projmatrices, tanfovx, tanfovy, H, W, dev):
    """(V,36) float rows: view16 | proj16 | tanfovx | tanfovy | focal_x | focal_y — the camera block of
    the C ABI.  Costs a few small host->device copies: callers with a fixed rig build it once and pass it
    as ``packed_cameras=``."""
    vm = _f32c(viewmatrices).reshape(-1, 16)
    pm = _f32c(projmatrices).reshape(-1, 16)
    V = vm.size(0)
    tfx = torch.as_tensor(tanfovx, dtype=torch.float64).reshape(-1).expand(V) if not torch.is_tensor(tanfovx) \
        else tanfovx.detach().double().reshape(-1).cpu().expand(V)
    tfy = torch.as_tensor(tanfovy, dtype=torch.float64).reshape(-1).expand(V) if not torch.is_tensor(tanfovy) \
        else tanfovy.detach().double().reshape(-1).cpu().expand(V)
    # focal = size / (2 * tan) evaluated in float32 like rasterizer_impl.cu:222-223
    tf = torch.stack((tfx, tfy), 1).float()
    focal = torch.stack((torch.tensor(float(W)) / (2.0 * tf[:, 0]), torch.tensor(float(H)) / (2.0 * tf[:, 1])), 1)
    return torch.cat((vm, pm, tf.to(dev), focal.float().to(dev)), 1).contiguous()


def sh_to_rgb(means3D, campos, shs, degree):
    """View-dependent colours from spherical harmonics (``forward.cu:20-71``): means3D (P,3), campos (3,) the camera
    centre, shs (P,M,3), ``degree`` 0..3 with M >= (degree+1)^2 -> (colors (P,3) float32, clamped (P,3) uint8)."""
    _lib.require_cuda(means3D, campos, shs)
    if shs.dim() != 3 or shs.size(2) != 3 or shs.size(0) != means3D.size(0):
        raise RuntimeError('shs must have dimensions (num_points, num_coefficients, 3)')
    P, M, degree = means3D.size(0), shs.size(1), int(degree)
    if not 0 <= degree <= 3 or M < (degree + 1) ** 2:
        raise RuntimeError(f'sh_degree {degree} needs 0 <= degree <= 3 and at least {(degree + 1) ** 2} coefficients, got {M}')
    means3D, campos, shs = _f32c(means3D), _f32c(campos).reshape(3), _f32c(shs)
    dev = means3D.device
    colors = torch.empty(P, 3, device=dev)
    clamped = torch.empty(P, 3, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ocrf_sh_to_rgb(P, degree, M, _lib.ptr(means3D), _lib.ptr(campos), _lib.ptr(shs),
                                             _lib.ptr(colors), _lib.ptr(clamped), _lib.stream_ptr(dev)), 'ocrf_sh_to_rgb')
    return colors, clamped


def sh_to_rgb_backward(means3D, campos, shs, degree, clamped, dL_dcolors, dL_dmeans3D=None):
    """Backward of ``sh_to_rgb`` (``backward.cu:20-140``) -> (dL_dmeans3D (P,3), dL_dshs (P,M,3)).  The means' part (the
    view direction depends on the mean) is ADDED to ``dL_dmeans3D`` when one is given (in place), else returned alone."""
    _lib.require_cuda(means3D, campos, shs, clamped, dL_dcolors)
    P, M = means3D.size(0), shs.size(1)
    means3D, campos, shs, g = _f32c(means3D), _f32c(campos).reshape(3), _f32c(shs), _f32c(dL_dcolors).reshape(P, 3)
    dev = means3D.device
    if dL_dmeans3D is None:
        dL_dmeans3D = torch.zeros(P, 3, device=dev)
    elif not (dL_dmeans3D.is_contiguous() and dL_dmeans3D.dtype == torch.float32 and dL_dmeans3D.numel() == 3 * P):
        raise RuntimeError('dL_dmeans3D must be a contiguous float32 (P,3) tensor')
    d_sh = torch.empty(P, M, 3, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ocrf_sh_to_rgb_backward(
            P, int(degree), M, _lib.ptr(means3D), _lib.ptr(campos), _lib.ptr(shs), _lib.ptr(clamped.contiguous()),
            _lib.ptr(g), _lib.ptr(dL_dmeans3D), _lib.ptr(d_sh), _lib.stream_ptr(dev)), 'ocrf_sh_to_rgb_backward')
    return dL_dmeans3D, d_sh


def rasterize_views_backward(grad_color, fwd, means3D, colors, opacities, scales, rotations, viewmatrices,
                             projmatrices, tanfovx, tanfovy, image_height, image_width, bg, scale_modifier=1.0,
                             want_means2D=False, packed_cameras=None, cov3D_precomp=None):
    """Backward of ``rasterize_views``' colour output, summed over the views.  ``fwd`` is the dict
    ``rasterize_views`` returned for the same inputs.  Returns a dict ``means3D`` (P,3), ``colors``
    (P,3), ``opacities`` (P,1), ``scales`` (P,3), ``rotations`` (P,4) [, ``means2D`` (V,P,3)]; with
    ``cov3D_precomp`` (P,6) (``scales`` / ``rotations`` None) ``cov3D`` (P,6) instead of the last two
    (backward.cu:346-396 without computeCov3D's backward)."""
    if cov3D_precomp is not None:
        return _backward_cov3d(grad_color, fwd, means3D, colors, opacities, cov3D_precomp, viewmatrices, projmatrices,
                               tanfovx, tanfovy, image_height, image_width, bg, want_means2D, packed_cameras)
    _lib.require_cuda(means3D, colors, opacities, scales, rotations, grad_color)
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(image_height), int(image_width)
    cams = packed_cameras if packed_cameras is not None else \
        pack_cameras(viewmatrices, projmatrices, tanfovx, tanfovy, H, W, dev)
    V = cams.size(0)
    g = _f32c(grad_color).reshape(V, 3, H, W)
    means3D, colors, opac = _f32c(means3D), _f32c(colors), _f32c(opacities).reshape(-1)
    sc, rot, bg = _f32c(scales), _f32c(rotations), _f32c(bg).reshape(3)
    fc, ft, fn = fwd['color'].contiguous(), fwd['final_T'].contiguous(), fwd['n_contrib'].contiguous()
    if tuple(fc.shape) != (V, 3, H, W) or tuple(ft.shape) != (V, H, W) or tuple(fn.shape) != (V, H, W):
        raise RuntimeError('rasterize_views_backward: forward outputs do not match the view batch')
    out = dict(means3D=torch.empty(P, 3, device=dev), colors=torch.empty(P, 3, device=dev),
               opacities=torch.empty(P, 1, device=dev), scales=torch.empty(P, 3, device=dev),
               rotations=torch.empty(P, 4, device=dev))
    m2d = torch.empty(V, P, 3, device=dev) if want_means2D else None
    if P == 0:
        if m2d is not None:
            out['means2D'] = m2d
        return out
    L = _lib.lib()
    with _lib.on_device(dev):
        need = L.ocrf_rasterize_backward_workspace_bytes(P, V)
        ws = _lib.workspace.get(dev, need, 'raster')
        _lib.check(L.ocrf_rasterize_backward(
            P, V, H, W, _lib.ptr(means3D), _lib.ptr(colors), _lib.ptr(opac), _lib.ptr(sc),
            ctypes.c_float(scale_modifier), _lib.ptr(rot), _lib.ptr(cams), _lib.ptr(bg), _lib.ptr(fc), _lib.ptr(ft),
            _lib.ptr(fn), _lib.ptr(g), _lib.ptr(out['means3D']), _lib.ptr(out['colors']), _lib.ptr(out['opacities']),
            _lib.ptr(out['scales']), _lib.ptr(out['rotations']), _lib.ptr(m2d), _lib.ptr(ws),
            ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev)), 'ocrf_rasterize_backward')
    if m2d is not None:
        out['means2D'] = m2d
    return out


def rasterize_views(means3D, colors, opacities, scales, rotations, viewmatrices, projmatrices,
                    tanfovx, tanfovy, image_height, image_width, bg, scale_modifier=1.0,
                    cov3D_precomp=None, depth_mode='median', want_tiles_touched=False, packed_cameras=None,
                    workspace_tag='raster', want_n_contrib=True):
    """Render ``V`` cameras over the same ``P`` Gaussians.

    ``viewmatrices`` / ``projmatrices``: (V,4,4) transposed matrices as the reference passes them;
    ``tanfovx`` / ``tanfovy``: length-V sequences (or scalars).  Returns a dict of fresh tensors:
    ``color`` (V,3,H,W), ``depth`` (V,1,H,W), ``final_T`` (V,H,W), ``n_contrib`` (V,H,W) int32,
    ``radii`` (V,P) int32 [, ``tiles_touched`` (V,P) int32].  Calls that may run concurrently on
    different streams must use different ``workspace_tag``s (the scratch buffer is per tag).
    ``want_n_contrib=False`` (inference: only the backward reads the per-pixel contributor index) drops
    ``n_contrib`` and the two selects per pixel-record that track it."""
    _lib.require_cuda(means3D, colors, opacities, bg)
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError('means3D must have dimensions (num_points, 3)')     # rasterize_points.cu:57-59
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(image_height), int(image_width)
    cams = packed_cameras if packed_cameras is not None else \
        pack_cameras(viewmatrices, projmatrices, tanfovx, tanfovy, H, W, dev)              # (V,36)
    _lib.require_cuda(cams)
    V = cams.size(0)
    means3D, colors, opac = _f32c(means3D), _f32c(colors), _f32c(opacities).reshape(-1)
    if cov3D_precomp is not None and cov3D_precomp.numel() > 0:
        cov, sc, rot = _f32c(cov3D_precomp), None, None
    else:
        cov, sc, rot = None, _f32c(scales), _f32c(rotations)
    bg = _f32c(bg).reshape(3)
    out = dict(color=torch.empty(V, 3, H, W, device=dev), depth=torch.empty(V, 1, H, W, device=dev),
               final_T=torch.empty(V, H, W, device=dev),
               radii=torch.empty(V, max(P, 0), dtype=torch.int32, device=dev))
    if want_n_contrib:
        out['n_contrib'] = torch.empty(V, H, W, dtype=torch.int32, device=dev)
    tt = torch.empty(V, P, dtype=torch.int32, device=dev) if want_tiles_touched else None
    status = torch.empty(1, dtype=torch.int32, device=dev)      # zeroed on the device by the bucket scan kernel
    L = _lib.lib()
    with _lib.on_device(dev):
        need = L.ocrf_rasterize_workspace_bytes(P, V)
        ws = _lib.workspace.get(dev, need, workspace_tag)
        _lib.check(L.ocrf_rasterize_forward(
            P, V, H, W, _lib.ptr(means3D), _lib.ptr(colors), _lib.ptr(opac), _lib.ptr(sc),
            ctypes.c_float(scale_modifier), _lib.ptr(rot), _lib.ptr(cov), _lib.ptr(cams), _lib.ptr(bg),
            {'median': 0, 'mean': 1}[depth_mode], _lib.ptr(out['color']), _lib.ptr(out['depth']),
            _lib.ptr(out['final_T']), _lib.ptr(out.get('n_contrib')), _lib.ptr(out['radii']), _lib.ptr(tt),
            _lib.ptr(status), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev)), 'ocrf_rasterize_forward')
    if tt is not None:
        out['tiles_touched'] = tt
    out['status'] = status     # device int, informational: 2 = the exact streaming path ran for some tile
    return out


def rasterize_sets(means3D, colors, opacities, scales, rotations, packed_cameras, image_height, image_width, bg,
                   scale_modifier=1.0, depth_mode='median', workspace_tag='raster', want_n_contrib=False):
    """``S`` Gaussian sets x ``V`` views each in one set of launches (C ABI ``ocrf_rasterize_forward_sets``):
    means3D (S,P,3), colors (S,P,3), opacities (S,P[,1]), scales (S,P,3), rotations (S,P,4),
    ``packed_cameras`` (S*V,36) from ``pack_cameras`` (view v renders set v // V).  Forward only, so the
    contributor index is not tracked unless ``want_n_contrib``.
    -> dict like ``rasterize_views`` with a leading S*V view axis."""
    _lib.require_cuda(means3D, colors, opacities, scales, rotations, packed_cameras, bg)
    if means3D.dim() != 3 or means3D.size(2) != 3:
        raise RuntimeError('means3D must have dimensions (num_sets, num_points, 3)')
    dev = means3D.device
    S, P = means3D.size(0), means3D.size(1)
    NV = packed_cameras.size(0)
    if NV % S:
        raise RuntimeError('packed_cameras rows must be a multiple of the number of sets')
    H, W = int(image_height), int(image_width)
    means3D, colors, opac = _f32c(means3D), _f32c(colors), _f32c(opacities).reshape(S, P)
    sc, rot, bg, cams = _f32c(scales), _f32c(rotations), _f32c(bg).reshape(3), _f32c(packed_cameras)
    out = dict(color=torch.empty(NV, 3, H, W, device=dev), depth=torch.empty(NV, 1, H, W, device=dev),
               final_T=torch.empty(NV, H, W, device=dev), radii=torch.empty(NV, P, dtype=torch.int32, device=dev))
    if want_n_contrib:
        out['n_contrib'] = torch.empty(NV, H, W, dtype=torch.int32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    L = _lib.lib()
    with _lib.on_device(dev):
        need = L.ocrf_rasterize_workspace_bytes(P, NV)
        ws = _lib.workspace.get(dev, need, workspace_tag)
        _lib.check(L.ocrf_rasterize_forward_sets(
            P, S, NV // S, H, W, _lib.ptr(means3D), _lib.ptr(colors), _lib.ptr(opac), _lib.ptr(sc),
            ctypes.c_float(scale_modifier), _lib.ptr(rot), _lib.ptr(None), _lib.ptr(cams), _lib.ptr(bg),
            {'median': 0, 'mean': 1}[depth_mode], _lib.ptr(out['color']), _lib.ptr(out['depth']), _lib.ptr(out['final_T']),
            _lib.ptr(out.get('n_contrib')), _lib.ptr(out['radii']), _lib.ptr(None), _lib.ptr(status), _lib.ptr(ws),
            ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev)), 'ocrf_rasterize_forward_sets')
    out['status'] = status
    return out


def _backward_cov3d(grad_color, fwd, means3D, colors, opacities, cov3D, viewmatrices, projmatrices, tanfovx, tanfovy,
                    image_height, image_width, bg, want_means2D, packed_cameras):
    _lib.require_cuda(means3D, colors, opacities, cov3D, grad_color)
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(image_height), int(image_width)
    cams = packed_cameras if packed_cameras is not None else \
        pack_cameras(viewmatrices, projmatrices, tanfovx, tanfovy, H, W, dev)
    V = cams.size(0)
    g = _f32c(grad_color).reshape(V, 3, H, W)
    means3D, colors, opac = _f32c(means3D), _f32c(colors), _f32c(opacities).reshape(-1)
    cov, bg = _f32c(cov3D).reshape(P, 6), _f32c(bg).reshape(3)
    fc, ft, fn = fwd['color'].contiguous(), fwd['final_T'].contiguous(), fwd['n_contrib'].contiguous()
    if tuple(fc.shape) != (V, 3, H, W) or tuple(ft.shape) != (V, H, W) or tuple(fn.shape) != (V, H, W):
        raise RuntimeError('rasterize_views_backward: forward outputs do not match the view batch')
    out = dict(means3D=torch.empty(P, 3, device=dev), colors=torch.empty(P, 3, device=dev),
               opacities=torch.empty(P, 1, device=dev), cov3D=torch.empty(P, 6, device=dev))
    m2d = torch.empty(V, P, 3, device=dev) if want_means2D else None
    if P:
        L = _lib.lib()
        with _lib.on_device(dev):
            ws = _lib.workspace.get(dev, L.ocrf_rasterize_backward_workspace_bytes(P, V), 'raster')
            _lib.check(L.ocrf_rasterize_backward_cov3d(
                P, V, H, W, _lib.ptr(means3D), _lib.ptr(colors), _lib.ptr(opac), _lib.ptr(cov), _lib.ptr(cams),
                _lib.ptr(bg), _lib.ptr(fc), _lib.ptr(ft), _lib.ptr(fn), _lib.ptr(g), _lib.ptr(out['means3D']),
                _lib.ptr(out['colors']), _lib.ptr(out['opacities']), _lib.ptr(out['cov3D']), _lib.ptr(m2d),
                _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev)), 'ocrf_rasterize_backward_cov3d')
    if m2d is not None:
        out['means2D'] = m2d
    return out


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings):
        rs = raster_settings
        ctx.has_sh = sh is not None and sh.numel() > 0
        if ctx.has_sh:
            # colours of THIS view from the SH coefficients (forward.cu:240-247); everything after is the ordinary pipeline
            colors_precomp, sh_clamped = sh_to_rgb(means3D, rs.campos, sh, rs.sh_degree)
        out = rasterize_views(means3D, colors_precomp, opacities, scales, rotations,
                              rs.viewmatrix.reshape(1, 4, 4), rs.projmatrix.reshape(1, 4, 4),
                              float(rs.tanfovx), float(rs.tanfovy), rs.image_height, rs.image_width,
                              rs.bg, float(rs.scale_modifier),
                              cov3Ds_precomp if cov3Ds_precomp is not None and cov3Ds_precomp.numel() else None,
                              want_n_contrib=any(ctx.needs_input_grad))      # only the backward reads it
        color, radii, depth = out['color'][0], out['radii'][0], out['depth'][0]
        ctx.mark_non_differentiable(radii, depth)
        ctx.raster_settings = rs
        ctx.has_cov = cov3Ds_precomp is not None and cov3Ds_precomp.numel() > 0
        # the forward's outputs are saved through autograd (not as plain attributes): the backward
        # rebuilds S = C_out - T_final * bg from them, so an in-place edit of the rendered image
        # (clamp_, mul_) must raise instead of silently changing the gradients
        if ctx.has_cov:
            ctx.save_for_backward(means3D, colors_precomp, opacities, cov3Ds_precomp, color, out['final_T'],
                                  out.get('n_contrib'))
        else:
            ctx.save_for_backward(means3D, colors_precomp, opacities, scales, rotations, color, out['final_T'],
                                  out.get('n_contrib'))
        ctx.sh_state = (sh, sh_clamped) if ctx.has_sh else None
        return color, radii, depth

    @staticmethod
    def _sh_grads(ctx, means3D, g):
        """With SH colours: dL/dcolors goes on to the coefficients and, through the view direction, to the means."""
        if not ctx.has_sh:
            return g['means3D'], None, g['colors']
        sh, clamped = ctx.sh_state
        rs = ctx.raster_settings
        d_means, d_sh = sh_to_rgb_backward(means3D, rs.campos, sh, rs.sh_degree, clamped, g['colors'], g['means3D'])
        return d_means, d_sh.to(sh.dtype), None

    @staticmethod
    def backward(ctx, grad_color, _r, _d):
        """Gradient tuple in the reference's order (diff_gaussian_rasterization/__init__.py:96-149):
        means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings."""
        rs = ctx.raster_settings
        if ctx.has_cov:
            means3D, colors, opacities, cov, fwd_color, fwd_T, fwd_n = ctx.saved_tensors
            fwd = dict(color=fwd_color.unsqueeze(0), final_T=fwd_T, n_contrib=fwd_n)
            g = rasterize_views_backward(grad_color.unsqueeze(0), fwd, means3D, colors, opacities, None, None,
                                         rs.viewmatrix.reshape(1, 4, 4), rs.projmatrix.reshape(1, 4, 4),
                                         float(rs.tanfovx), float(rs.tanfovy), rs.image_height, rs.image_width, rs.bg,
                                         want_means2D=True, cov3D_precomp=cov)
            d_means, d_sh, d_col = _RasterizeGaussians._sh_grads(ctx, means3D, g)
            grads = (d_means, g['means2D'][0], d_sh, d_col, g['opacities'].reshape(opacities.shape),
                     None, None, g['cov3D'].reshape(cov.shape), None)
            return tuple(v if need else None for v, need in zip(grads, ctx.needs_input_grad))
        means3D, colors, opacities, scales, rotations, fwd_color, fwd_T, fwd_n = ctx.saved_tensors
        fwd = dict(color=fwd_color.unsqueeze(0), final_T=fwd_T, n_contrib=fwd_n)
        g = rasterize_views_backward(grad_color.unsqueeze(0), fwd, means3D, colors, opacities, scales, rotations,
                                     rs.viewmatrix.reshape(1, 4, 4), rs.projmatrix.reshape(1, 4, 4),
                                     float(rs.tanfovx), float(rs.tanfovy), rs.image_height, rs.image_width, rs.bg,
                                     float(rs.scale_modifier), want_means2D=True)
        d_means, d_sh, d_col = _RasterizeGaussians._sh_grads(ctx, means3D, g)
        grads = (d_means, g['means2D'][0], d_sh, d_col, g['opacities'].reshape(opacities.shape),
                 g['scales'], g['rotations'], None, None)
        # inputs that were not tensors needing a gradient (means2D=None from render()) must get None
        return tuple(v if need else None for v, need in zip(grads, ctx.needs_input_grad))


class _RasterizeViews(torch.autograd.Function):
    """Differentiable batched render: colour (V,3,H,W) w.r.t. the Gaussian parameters, gradients
    summed over the views in one backward launch sequence."""

    @staticmethod
    def forward(ctx, means3D, colors, opacities, scales, rotations, cam):
        out = rasterize_views(means3D, colors, opacities, scales, rotations, *cam)
        ctx.cam = cam
        ctx.save_for_backward(means3D, colors, opacities, scales, rotations, out['color'], out['final_T'], out['n_contrib'])
        ctx.mark_non_differentiable(out['depth'], out['radii'], out['final_T'])
        return out['color'], out['depth'], out['final_T'], out['radii']

    @staticmethod
    def backward(ctx, grad_color, _d, _t, _r):
        means3D, colors, opacities, scales, rotations, fwd_color, fwd_T, fwd_n = ctx.saved_tensors
        fwd = dict(color=fwd_color, final_T=fwd_T, n_contrib=fwd_n)
        g = rasterize_views_backward(grad_color, fwd, means3D, colors, opacities, scales, rotations, *ctx.cam)
        return g['means3D'], g['colors'], g['opacities'].reshape(opacities.shape), g['scales'], g['rotations'], None


def rasterize_views_autograd(means3D, colors, opacities, scales, rotations, viewmatrices, projmatrices, tanfovx,
                             tanfovy, image_height, image_width, bg, scale_modifier=1.0):
    """``rasterize_views`` with autograd through the colour output: returns (color (V,3,H,W),
    depth (V,1,H,W), final_T (V,H,W), radii (V,P))."""
    cam = (viewmatrices, projmatrices, tanfovx, tanfovy, image_height, image_width, bg, scale_modifier)
    return _RasterizeViews.apply(means3D, colors, opacities, scales, rotations, cam)


class _RasterizePacked(torch.autograd.Function):
    """``_RasterizeViews`` for cameras already packed on the device ((V,36) rows of ``pack_cameras``): nothing of the call
    touches the host's copy of the calibration (no matrix uploads, no focal arithmetic per call)."""

    @staticmethod
    def forward(ctx, means3D, colors, opacities, scales, rotations, packed, H, W, bg):
        out = rasterize_views(means3D, colors, opacities, scales, rotations, None, None, None, None, H, W, bg,
                              packed_cameras=packed)
        ctx.cam = (packed, H, W, bg)
        ctx.save_for_backward(means3D, colors, opacities, scales, rotations, out['color'], out['final_T'], out['n_contrib'])
        ctx.mark_non_differentiable(out['depth'], out['radii'], out['final_T'])
        return out['color'], out['depth'], out['final_T'], out['radii']

    @staticmethod
    def backward(ctx, grad_color, _d, _t, _r):
        means3D, colors, opacities, scales, rotations, fwd_color, fwd_T, fwd_n = ctx.saved_tensors
        packed, H, W, bg = ctx.cam
        fwd = dict(color=fwd_color, final_T=fwd_T, n_contrib=fwd_n)
        g = rasterize_views_backward(grad_color, fwd, means3D, colors, opacities, scales, rotations, None, None, None, None,
                                     H, W, bg, packed_cameras=packed)
        need = ctx.needs_input_grad
        grads = (g['means3D'], g['colors'], g['opacities'].reshape(opacities.shape), g['scales'], g['rotations'])
        return tuple(v if n else None for v, n in zip(grads, need)) + (None, None, None, None)


def rasterize_packed_autograd(means3D, colors, opacities, scales, rotations, packed_cameras, image_height, image_width, bg):
    """``rasterize_views_autograd`` with the cameras given as packed device rows (V,36):
    -> (color (V,3,H,W), depth (V,1,H,W), final_T (V,H,W), radii (V,P)), autograd through the colour."""
    return _RasterizePacked.apply(means3D, colors, opacities, scales, rotations, packed_cameras, int(image_height),
                                  int(image_width), bg)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales,
                                     rotations, cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Near-plane test of ``in_frustum`` (auxiliary.h:139-164): view-space z > 0.2."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix.reshape(4, 4).to(positions)
            z = positions[:, 0] * vm[0, 2] + positions[:, 1] * vm[1, 2] + positions[:, 2] * vm[2, 2] + vm[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)
