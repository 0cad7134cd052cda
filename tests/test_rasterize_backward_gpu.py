"""GPU parity: HIP rasteriser backward (C ABI ocrf_rasterize_backward through the reference-shaped
autograd surface) vs the C oracle's restatement of cuda_rasterizer/backward.cu.

Bar: float gradients within 1e-4 relative to the gradient scale of each output (north_star's float
tolerance; the reference itself is not bit-reproducible here — float atomics, backward.cu:509-541).
A pixel whose alpha sits within an ulp of a threshold can take a different branch than the oracle
(see test_rasterize_gpu.py); the scenes are sized so that such flips stay below the bar, and the
few-Gaussian cases below have none."""
import numpy as np
import pytest
import torch

from ocrfdet_amd import diff_gaussian_rasterization as dgr
from tests import helpers

pytestmark = pytest.mark.gpu


def _t(a, cuda, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda).requires_grad_(grad)


def _check(got, want, name, tol=1e-4):
    scale = max(float(np.abs(want).max()), 1e-6)
    err = float(np.abs(got - want).max()) / scale
    assert err <= tol, f'{name}: max err {err:.3e} of the gradient scale {scale:.3e}'


def _run(oracle_lib, cuda, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, bg, seed=0, tol=1e-4):
    rng = np.random.default_rng(seed)
    gcol = rng.standard_normal((3, H, W)).astype(np.float32)
    want = oracle_lib.rasterize_backward(gcol, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, np.float32(bg))
    settings = dgr.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy, bg=_t(np.float32(bg), cuda), scale_modifier=1.0,
        viewmatrix=_t(view, cuda), projmatrix=_t(full, cuda), sh_degree=0, campos=torch.zeros(3, device=cuda),
        prefiltered=False)
    m3, m2 = _t(xyz, cuda, True), torch.zeros(xyz.shape[0], 3, device=cuda, requires_grad=True)
    c, o, s, r = _t(rgb, cuda, True), _t(opac, cuda, True), _t(sc, cuda, True), _t(rot, cuda, True)
    color, radii, depth = dgr.GaussianRasterizer(settings)(m3, m2, o, colors_precomp=c, scales=s, rotations=r)
    (color * _t(gcol, cuda)).sum().backward()
    torch.cuda.synchronize()
    _check(c.grad.cpu().numpy(), want['colors'], 'dL_dcolors', tol)
    _check(o.grad.cpu().numpy(), want['opacities'], 'dL_dopacity', tol)
    _check(m2.grad.cpu().numpy(), want['means2D'], 'dL_dmeans2D', tol)
    _check(m3.grad.cpu().numpy(), want['means3D'], 'dL_dmeans3D', tol)
    _check(s.grad.cpu().numpy(), want['scales'], 'dL_dscales', tol)
    _check(r.grad.cpu().numpy(), want['rotations'], 'dL_drotations', tol)
    return want


def test_single_gaussian_backward(cuda, oracle_lib):
    W, H = 64, 48
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    _run(oracle_lib, cuda, np.float32([[0.3, -0.2, 5.0]]), np.float32([[0.2, 0.5, 0.9]]), np.float32([[0.7]]),
         np.float32([[0.3, 0.2, 0.25]]), np.float32([[0.9, 0.1, -0.3, 0.2]]), view, full, tfx, tfy, H, W,
         (0.1, 0.2, 0.3))


def test_few_overlapping_gaussians_backward(cuda, oracle_lib):
    W, H = 80, 48
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    rng = np.random.default_rng(3)
    n = 12
    xyz = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.6, 0.6, n), rng.uniform(3, 9, n)], 1).astype(np.float32)
    _run(oracle_lib, cuda, xyz, rng.uniform(0, 1, (n, 3)).astype(np.float32),
         rng.uniform(0.2, 0.9, (n, 1)).astype(np.float32), rng.uniform(0.1, 0.5, (n, 3)).astype(np.float32),
         rng.standard_normal((n, 4)).astype(np.float32), view, full, tfx, tfy, H, W, (0.3, 0.1, 0.6))


@pytest.mark.parametrize('n,seed', [(300, 0), (5000, 1)])
def test_random_scene_backward(cuda, oracle_lib, n, seed):
    rng = np.random.default_rng(seed)
    W, H = 176, 64
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, n)
    _run(oracle_lib, cuda, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, (0.2, 0.4, 0.1), seed, tol=2e-3)


def test_multi_view_backward_sums_views(cuda, oracle_lib):
    """rasterize_views_autograd: one backward over V cameras == the sum of the per-view oracle backwards."""
    rng = np.random.default_rng(5)
    W, H = 96, 64
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    view2, full2, _, _ = helpers.simple_camera(W, H, cam_pos=(-0.4, 0.1, 0.0))
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, 400)
    bg = np.float32([0.1, 0.1, 0.1])
    gcol = rng.standard_normal((2, 3, H, W)).astype(np.float32)
    w0 = oracle_lib.rasterize_backward(gcol[0], xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, bg)
    w1 = oracle_lib.rasterize_backward(gcol[1], xyz, rgb, opac, sc, rot, view2, full2, tfx, tfy, H, W, bg)
    m3, c, o = _t(xyz, cuda, True), _t(rgb, cuda, True), _t(opac, cuda, True)
    s, r = _t(sc, cuda, True), _t(rot, cuda, True)
    vms = torch.stack((_t(view, cuda), _t(view2, cuda)))
    pms = torch.stack((_t(full, cuda), _t(full2, cuda)))
    color, depth, final_T, radii = dgr.rasterize_views_autograd(m3, c, o, s, r, vms, pms, [tfx, tfx], [tfy, tfy], H, W,
                                                                _t(bg, cuda))
    (color * _t(gcol, cuda)).sum().backward()
    torch.cuda.synchronize()
    for name, got in (('means3D', m3.grad), ('colors', c.grad), ('opacities', o.grad), ('scales', s.grad),
                      ('rotations', r.grad)):
        _check(got.cpu().numpy(), w0[name] + w1[name], name, 2e-3)


def test_backward_of_empty_view_is_zero(cuda):
    W, H = 32, 32
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    xyz = np.float32([[0, 0, -5.0], [100.0, 0, 5.0]])       # behind the camera / far off screen
    m3 = _t(xyz, cuda, True)
    c = _t(np.float32([[1, 1, 1], [1, 1, 1]]), cuda, True)
    color, _, _, _ = dgr.rasterize_views_autograd(m3, c, _t(np.float32([[0.5], [0.5]]), cuda),
                                                  _t(np.float32([[0.1] * 3] * 2), cuda),
                                                  _t(np.float32([[1, 0, 0, 0]] * 2), cuda), _t(view, cuda).view(1, 4, 4),
                                                  _t(full, cuda).view(1, 4, 4), [tfx], [tfy], H, W,
                                                  torch.zeros(3, device=cuda))
    color.sum().backward()
    assert float(m3.grad.abs().max()) == 0.0 and float(c.grad.abs().max()) == 0.0


def _cov3d_of(sc, rot):
    """computeCov3D (forward.cu:118-121 layout xx xy xz yy yz zz) in float32, Sigma = (S R)^T (S R)."""
    r, x, y, z = (rot[:, i].astype(np.float32) for i in range(4))
    one, two = np.float32(1), np.float32(2)
    R = np.stack([np.stack([one - two * (y * y + z * z), two * (x * y - r * z), two * (x * z + r * y)], 1),
                  np.stack([two * (x * y + r * z), one - two * (x * x + z * z), two * (y * z - r * x)], 1),
                  np.stack([two * (x * z - r * y), two * (y * z + r * x), one - two * (x * x + y * y)], 1)], 1)   # (P,3,3)
    M = sc[:, :, None].astype(np.float32) * np.transpose(R, (0, 2, 1))     # M[k][i] = s_k R[i][k]
    S = np.einsum('pki,pkj->pij', M, M).astype(np.float32)
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)


@pytest.mark.parametrize('n,seed,tol', [(12, 3, 1e-4), (2000, 7, 2e-3)])
def test_backward_with_cov3d_precomp(cuda, oracle_lib, n, seed, tol):
    """cov3D_precomp path (backward.cu:346-396 stops at dL/dcov3D): the oracle returns dL/dcov3D of the same scene
    built from (scales, rotations); the HIP op gets the covariance itself and must return that gradient, and the
    same gradients for everything upstream of the covariance."""
    rng = np.random.default_rng(seed)
    W, H = 112, 64
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, n)
    rot = (rot / np.linalg.norm(rot, axis=1, keepdims=True)).astype(np.float32)
    bg = np.float32([0.2, 0.1, 0.3])
    gcol = rng.standard_normal((3, H, W)).astype(np.float32)
    want = oracle_lib.rasterize_backward(gcol, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, bg)
    settings = dgr.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy, bg=_t(bg, cuda), scale_modifier=1.0,
        viewmatrix=_t(view, cuda), projmatrix=_t(full, cuda), sh_degree=0, campos=torch.zeros(3, device=cuda),
        prefiltered=False)
    m3, m2 = _t(xyz, cuda, True), torch.zeros(n, 3, device=cuda, requires_grad=True)
    c, o, cov = _t(rgb, cuda, True), _t(opac, cuda, True), _t(_cov3d_of(sc, rot), cuda, True)
    color, radii, depth = dgr.GaussianRasterizer(settings)(m3, m2, o, colors_precomp=c, cov3D_precomp=cov)
    (color * _t(gcol, cuda)).sum().backward()
    torch.cuda.synchronize()
    _check(c.grad.cpu().numpy(), want['colors'], 'dL_dcolors', tol)
    _check(o.grad.cpu().numpy(), want['opacities'], 'dL_dopacity', tol)
    _check(m2.grad.cpu().numpy(), want['means2D'], 'dL_dmeans2D', tol)
    _check(m3.grad.cpu().numpy(), want['means3D'], 'dL_dmeans3D', tol)
    _check(cov.grad.cpu().numpy(), want['cov3D'], 'dL_dcov3D', tol)
    # and the two forward paths render the same image
    s, r = _t(sc, cuda), _t(rot, cuda)
    with torch.no_grad():
        color2, _, _ = dgr.GaussianRasterizer(settings)(_t(xyz, cuda), None, _t(opac, cuda), colors_precomp=_t(rgb, cuda),
                                                        scales=s, rotations=r)
    assert float((color2 - color.detach()).abs().max()) <= 1e-4
