"""GPU parity of csrc/raster_sh.hip: spherical-harmonics colours (the ``shs`` argument of the reference's rasteriser,
forward.cu:20-71 / backward.cu:20-140) against the numpy oracle (oracle/sh.py) and, end to end, against the same render
fed with the oracle's colours as ``colors_precomp``.  Tolerances: 2e-6 on colours (fp32, another summation grouping; values
O(1)), 1e-4 of the largest entry on gradients (north_star's bar), clamp masks equal except where the unclamped colour is
within 2e-6 of zero."""
import math

import numpy as np
import pytest
import torch

from ocrfdet_amd import diff_gaussian_rasterization as dgr
from oracle import sh as osh
from tests import helpers

pytestmark = pytest.mark.gpu


def _t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.mark.parametrize('deg,M', [(0, 1), (0, 16), (1, 4), (1, 9), (2, 9), (2, 16), (3, 16), (3, 20)])
def test_sh_colours_and_their_backward_vs_oracle(cuda, deg, M):
    rng = np.random.default_rng(10 * deg + M)
    P = 3001                                             # not a multiple of the 256-thread block
    means = (rng.standard_normal((P, 3)) * 6).astype(np.float32)
    campos = np.array([0.4, -1.1, 0.9], np.float32)
    shs = (rng.standard_normal((P, M, 3)) * 0.7).astype(np.float32)
    g = rng.standard_normal((P, 3)).astype(np.float32)
    want, want_clamped = osh.sh_to_rgb(means, campos, shs, deg)
    raw = osh.sh_to_rgb(means, campos, shs + np.float32(0), deg, dtype=np.float64)[0]
    col, clamped = dgr.sh_to_rgb(_t(means, cuda), _t(campos, cuda), _t(shs, cuda), deg)
    assert np.abs(col.cpu().numpy() - want).max() <= 2e-6
    differ = clamped.cpu().numpy().astype(bool) != want_clamped
    assert want_clamped.any() and (np.abs(raw[differ]) <= 2e-6).all()     # a clamp may flip only where the colour is ~0
    # backward: the oracle with the kernel's own clamp mask (so a flipped bit is not counted twice)
    mask = clamped.cpu().numpy().astype(bool)
    wm, ws = osh.sh_to_rgb_backward(means, campos, shs, deg, mask, g)
    base = (rng.standard_normal((P, 3))).astype(np.float32)                # the rasteriser's part, already in the buffer
    acc = _t(base.copy(), cuda)
    dm, ds = dgr.sh_to_rgb_backward(_t(means, cuda), _t(campos, cuda), _t(shs, cuda), deg, clamped, _t(g, cuda), acc)
    assert dm.data_ptr() == acc.data_ptr()                                 # added in place
    wm64, ws64 = osh.sh_to_rgb_backward(means, campos, shs, deg, mask, g, dtype=np.float64)
    assert np.abs(ds.cpu().numpy() - ws64).max() <= 1e-4 * np.abs(ws64).max()
    if deg > 0:
        assert np.abs(dm.cpu().numpy() - base - wm64).max() <= 1e-4 * np.abs(wm64).max() + 1e-6
    else:
        assert torch.equal(dm.cpu(), torch.from_numpy(base))               # degree 0 does not see the direction
    assert not ds[:, (deg + 1) ** 2:].any()
    # (the float32 oracle agrees with its float64 self to the same bar: the comparison above is not looser than the oracle)
    assert np.abs(ws - ws64).max() <= 1e-4 * np.abs(ws64).max() and np.abs(wm - wm64).max() <= 1e-4 * np.abs(wm64).max() + 1e-6


def _camera(cuda, W, H, pos):
    view, full, tfx, tfy = helpers.simple_camera(W, H, cam_pos=pos)
    return view, full, tfx, tfy, dgr.GaussianRasterizationSettings(
        H, W, tfx, tfy, torch.zeros(3, device=cuda), 1.0, _t(view, cuda), _t(full, cuda), 3, _t(np.float32(pos), cuda), False)


def test_render_with_shs_is_the_render_of_the_sh_colours(cuda):
    """``GaussianRasterizer(...)(means3D, means2D, opacities, shs=...)`` (the reference's call with SH,
    diff_gaussian_rasterization/__init__.py:171-221): the image equals, bit for bit, the render of the colours ``sh_to_rgb``
    gives for that camera, and the gradients are the chain of the two backwards."""
    rng = np.random.default_rng(5)
    W, H, P = 96, 64, 1500
    xyz, _, opac, sc, rot = helpers.random_gaussians(rng, P)
    shs = (rng.standard_normal((P, 16, 3)) * 0.5).astype(np.float32)
    pos = (0.3, -0.2, 0.1)
    _, _, _, _, settings = _camera(cuda, W, H, pos)
    for deg in (0, 2, 3):
        s = settings._replace(sh_degree=deg)
        t = {k: _t(v, cuda).requires_grad_(True) for k, v in dict(xyz=xyz, shs=shs, opac=opac, sc=sc, rot=rot).items()}
        img, radii, depth = dgr.GaussianRasterizer(s)(t['xyz'], None, t['opac'], shs=t['shs'], scales=t['sc'],
                                                      rotations=t['rot'])
        w = torch.rand_like(img)
        (img * w).sum().backward()
        # the same through precomputed colours
        u = {k: _t(v, cuda).requires_grad_(True) for k, v in dict(xyz=xyz, opac=opac, sc=sc, rot=rot).items()}
        col, clamped = dgr.sh_to_rgb(u['xyz'].detach(), s.campos, _t(shs, cuda), deg)
        col.requires_grad_(True)
        img2, radii2, depth2 = dgr.GaussianRasterizer(s)(u['xyz'], None, u['opac'], colors_precomp=col, scales=u['sc'],
                                                         rotations=u['rot'])
        assert torch.equal(img, img2) and torch.equal(radii, radii2) and torch.equal(depth, depth2)
        (img2 * w).sum().backward()
        d_means, d_sh = dgr.sh_to_rgb_backward(u['xyz'].detach(), s.campos, _t(shs, cuda), deg, clamped, col.grad,
                                               u['xyz'].grad.clone())
        # float atomics in the rasteriser's backward: the two runs sum in different orders
        for name, got, want in (('shs', t['shs'].grad, d_sh), ('means3D', t['xyz'].grad, d_means),
                                ('opacities', t['opac'].grad, u['opac'].grad), ('scales', t['sc'].grad, u['sc'].grad)):
            assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-7, (deg, name)
        assert float(t['shs'].grad[:, (deg + 1) ** 2:].abs().max()) == 0 if deg < 3 else True


def test_sh_gradients_vs_float64_autograd_of_the_oracle_formulation(cuda):
    """The whole SH stage against torch autograd in float64 (the transcription tests/test_oracle_sh.py pins the oracle
    with): d colour / d coefficients and d colour / d mean for a random upstream gradient, degree 3."""
    from tests.test_oracle_sh import _torch_forward
    rng = np.random.default_rng(11)
    P = 777
    means = rng.standard_normal((P, 3)) * 4
    shs = rng.standard_normal((P, 16, 3)) * 0.6
    campos = np.array([1.0, 0.5, -0.25])
    g = rng.standard_normal((P, 3))
    tm, ts = torch.tensor(means, requires_grad=True), torch.tensor(shs, requires_grad=True)
    (_torch_forward(tm, torch.tensor(campos), ts, 3) * torch.tensor(g)).sum().backward()
    f32 = lambda a: _t(np.asarray(a, np.float32), cuda)                                     # noqa: E731
    col, clamped = dgr.sh_to_rgb(f32(means), f32(campos), f32(shs), 3)
    dm, ds = dgr.sh_to_rgb_backward(f32(means), f32(campos), f32(shs), 3, clamped, f32(g))
    assert float((ds.cpu().double() - ts.grad).abs().max()) <= 1e-4 * float(ts.grad.abs().max())
    assert float((dm.cpu().double() - tm.grad).abs().max()) <= 1e-4 * float(tm.grad.abs().max())


def test_sh_argument_errors(cuda):
    x = torch.zeros(4, 3, device=cuda)
    c = torch.zeros(3, device=cuda)
    with pytest.raises(RuntimeError, match='coefficients'):
        dgr.sh_to_rgb(x, c, torch.zeros(4, 8, 3, device=cuda), 3)            # degree 3 needs 16
    with pytest.raises(RuntimeError, match='coefficients'):
        dgr.sh_to_rgb(x, c, torch.zeros(4, 16, 3, device=cuda), 4)
    with pytest.raises(RuntimeError, match='num_points'):
        dgr.sh_to_rgb(x, c, torch.zeros(5, 16, 3, device=cuda), 1)
    col, cl = dgr.sh_to_rgb(torch.zeros(0, 3, device=cuda), c, torch.zeros(0, 16, 3, device=cuda), 3)
    assert col.shape == (0, 3) and cl.shape == (0, 3)
    assert math.isclose(float(dgr.sh_to_rgb(x + 1, c, torch.zeros(4, 1, 3, device=cuda), 0)[0][0, 0]), 0.5)
