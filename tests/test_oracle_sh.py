"""The SH oracle (oracle/sh.py) pinned the only way it can be without reference vectors: its backward against autograd
through a torch transcription of its own forward in float64, and its forward against the closed forms of the real SH basis
that can be checked by hand (degree 0 is a constant, degree 1 is linear in the direction, every band integrates to zero over
the sphere)."""
import numpy as np
import pytest
import torch

from oracle import sh as osh


def _torch_forward(means, campos, shs, deg):
    v = means - campos.reshape(1, 3)
    d = v / v.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    C1, C2, C3 = osh.C1, osh.C2, osh.C3
    polys = [torch.full_like(x, osh.C0)]
    if deg > 0:
        polys += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        polys += [C2[0] * x * y, C2[1] * y * z, C2[2] * (2 * z * z - x * x - y * y), C2[3] * x * z, C2[4] * (x * x - y * y)]
    if deg > 2:
        polys += [C3[0] * y * (3 * x * x - y * y), C3[1] * x * y * z, C3[2] * y * (4 * z * z - x * x - y * y),
                  C3[3] * z * (2 * z * z - 3 * x * x - 3 * y * y), C3[4] * x * (4 * z * z - x * x - y * y),
                  C3[5] * z * (x * x - y * y), C3[6] * x * (x * x - 3 * y * y)]
    res = sum(p * shs[:, i] for i, p in enumerate(polys)) + 0.5
    return res.clamp(min=0)


@pytest.mark.parametrize('deg', [0, 1, 2, 3])
def test_backward_is_the_derivative_of_the_forward(deg):
    rng = np.random.default_rng(deg)
    P, M = 200, 16
    means = rng.standard_normal((P, 3)) * 5
    campos = np.array([0.3, -1.2, 0.7])
    shs = rng.standard_normal((P, M, 3)) * 0.6
    g = rng.standard_normal((P, 3))
    col, clamped = osh.sh_to_rgb(means, campos, shs, deg, dtype=np.float64)
    assert clamped.any() and not clamped.all()
    d_means, d_sh = osh.sh_to_rgb_backward(means, campos, shs, deg, clamped, g, dtype=np.float64)
    tm = torch.tensor(means, requires_grad=True)
    ts = torch.tensor(shs, requires_grad=True)
    out = _torch_forward(tm, torch.tensor(campos), ts, deg)
    np.testing.assert_allclose(out.detach().numpy(), col, rtol=0, atol=1e-12)
    (out * torch.tensor(g)).sum().backward()
    want_means = tm.grad.numpy() if tm.grad is not None else np.zeros_like(means)     # (degree 0: no direction involved)
    np.testing.assert_allclose(d_means, want_means, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(d_sh, ts.grad.numpy(), rtol=1e-9, atol=1e-12)
    n = (deg + 1) ** 2
    assert not d_sh[:, n:].any()                      # coefficients above the active degree get no gradient


def test_basis_properties():
    rng = np.random.default_rng(7)
    P = 20000
    means = rng.standard_normal((P, 3))               # isotropic directions
    campos = np.zeros(3)
    # one coefficient at a time: the colour minus 0.5 IS the basis function (keep it unclamped with a large offset in sh0)
    base = np.zeros((P, 16, 3))
    base[:, 0] = 10.0
    ref, _ = osh.sh_to_rgb(means, campos, base, 3, dtype=np.float64)
    assert np.allclose(ref, 10.0 * osh.C0 + 0.5)      # degree 0: constant
    for i in range(1, 16):
        shs = base.copy()
        shs[:, i] = 1.0
        b = osh.sh_to_rgb(means, campos, shs, 3, dtype=np.float64)[0][:, 0] - ref[:, 0]
        assert abs(b.mean()) < 0.02, i                # every band above 0 integrates to zero over the sphere
        assert abs((b * b).mean() - 1.0 / (4 * np.pi)) < 0.01, i      # and is normalised: <b^2> = 1 / (4 pi)
    d = means / np.linalg.norm(means, axis=1, keepdims=True)
    shs = base.copy()
    shs[:, 1:4] = np.array([2.0, -3.0, 5.0])[None, :, None]
    lin = osh.sh_to_rgb(means, campos, shs, 1, dtype=np.float64)[0][:, 0] - ref[:, 0]
    assert np.allclose(lin, osh.C1 * (-2.0 * d[:, 1] - 3.0 * d[:, 2] - 5.0 * d[:, 0]))


def test_float32_forward_close_to_float64():
    rng = np.random.default_rng(3)
    means, campos = rng.standard_normal((500, 3)).astype(np.float32) * 4, np.array([1, 2, 3], np.float32)
    shs = rng.standard_normal((500, 16, 3)).astype(np.float32)
    a, ca = osh.sh_to_rgb(means, campos, shs, 3)
    b, cb = osh.sh_to_rgb(means, campos, shs, 3, dtype=np.float64)
    assert a.dtype == np.float32 and np.abs(a - b).max() < 5e-6
    assert (ca != cb).sum() <= 1
