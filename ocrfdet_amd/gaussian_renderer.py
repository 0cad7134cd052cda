"""``render`` — drop-in for ``mmdet3d/models/necks/MVSGaussian/lib/gaussian_renderer/__init__.py:17-75``
(called at ``view_transformer_ocrf.py:1153``), plus the camera set-up that precedes it
(``view_transformer_ocrf.py:1135-1152`` with ``MVSGaussian/lib/utils/data_utils.py:703-733``)."""
import math

import numpy as np
import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views

__all__ = ['render', 'render_views', 'getWorld2View2', 'getProjectionMatrix', 'camera_from_calibration']


def render(data, idx, pts_xyz, pts_rgb, rotations, scales, opacity, bg_color):
    """data: dict(FovX, FovY, height, width, world_view_transform, full_proj_transform,
    camera_center) -> (image (3,H,W), depth (1,H,W)).  ``bg_color``: list of 3 floats."""
    bg = torch.tensor(bg_color, dtype=torch.float32, device=pts_xyz.device)
    settings = GaussianRasterizationSettings(
        image_height=int(data['height']), image_width=int(data['width']),
        tanfovx=math.tan(data['FovX'] * 0.5), tanfovy=math.tan(data['FovY'] * 0.5),
        bg=bg, scale_modifier=1.0, viewmatrix=data['world_view_transform'],
        projmatrix=data['full_proj_transform'], sh_degree=3, campos=data['camera_center'],
        prefiltered=False)
    image, _, depth = GaussianRasterizer(raster_settings=settings)(
        means3D=pts_xyz, means2D=None, shs=None, colors_precomp=pts_rgb, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=None)
    return image, depth


def render_views(cameras, pts_xyz, pts_rgb, rotations, scales, opacity, bg_color, height, width,
                 depth_mode='median'):
    """All cameras of a frame over one Gaussian set in one call.  ``cameras``: list of the dicts
    ``render`` takes.  -> dict with color (V,3,H,W), depth (V,1,H,W), final_T (V,H,W), ..."""
    dev = pts_xyz.device
    vm = torch.stack([c['world_view_transform'].to(dev) for c in cameras])
    pm = torch.stack([c['full_proj_transform'].to(dev) for c in cameras])
    tfx = [math.tan(float(c['FovX']) * 0.5) for c in cameras]
    tfy = [math.tan(float(c['FovY']) * 0.5) for c in cameras]
    bg = torch.tensor(bg_color, dtype=torch.float32, device=dev)
    return rasterize_views(pts_xyz, pts_rgb, opacity, scales, rotations, vm, pm, tfx, tfy, height, width,
                           bg, depth_mode=depth_mode)


def getWorld2View2(R, t, translate=np.array([.0, .0, .0]), scale=1.0):
    """World->view 4x4 (float32) from a rotation whose TRANSPOSE is stored and a translation, with
    the optional re-centring of the camera centre (data_utils.py:703-714)."""
    w2c = np.zeros((4, 4))
    w2c[:3, :3] = np.asarray(R).transpose()
    w2c[:3, 3] = t
    w2c[3, 3] = 1.0
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(c2w))


def getProjectionMatrix(znear, zfar, K, h, w):
    """OpenGL-style projection from pinhole intrinsics (data_utils.py:716-733), float32 torch 4x4.
    K is taken in float64, which is what numpy-1.x promotion made of the reference's
    python-float x np.float32 products."""
    K = np.asarray(K, dtype=np.float64)
    nfx, nfy = znear / K[0, 0], znear / K[1, 1]
    left, right = -(w - K[0, 2]) * nfx, K[0, 2] * nfx
    bottom, top = (K[1, 2] - h) * nfy, K[1, 2] * nfy
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_from_calibration(K, c2w, height, width, znear=0.01, zfar=999.9, device=None):
    """The ``data`` dict of ``render`` built as view_transformer_ocrf.py:1135-1152 builds it,
    quirks included: FoV from the intrinsics as given (the reference passes the unscaled 1600x900
    K with the 704x256 viewport, :1079,1143-1146), ``c2w[:3,:3]`` / ``c2w[:3,3]`` fed where a
    world->view rotation / translation are expected (:1140-1141)."""
    K = torch.as_tensor(K).detach().cpu()
    c2w = torch.as_tensor(c2w).detach().cpu()
    fov_x = 2 * torch.atan(torch.tensor(width).float() / (2 * K[0, 0]))
    fov_y = 2 * torch.atan(torch.tensor(height).float() / (2 * K[1, 1]))
    proj = getProjectionMatrix(znear, zfar, K.numpy(), height, width).transpose(0, 1).clone()
    w2v = torch.tensor(getWorld2View2(c2w[:3, :3].numpy(), c2w[:3, 3].numpy(), np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    full = w2v.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    center = w2v.inverse()[3, :3]
    d = dict(FovX=fov_x, FovY=fov_y, height=height, width=width, world_view_transform=w2v,
             full_proj_transform=full, camera_center=center)
    if device is not None:
        d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in d.items()}
    return d
