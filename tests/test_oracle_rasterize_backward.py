"""CPU: the oracle's rasteriser backward (oracle/rasterize_ref.c, restating cuda_rasterizer/backward.cu)
against autograd of an independent float64 torch re-evaluation of the same forward, with the forward's
discrete decisions (tile coverage, 1/255 skip, T < 1e-4 stop, 0.99 clamp) frozen — which is what the
reference's analytic backward differentiates too."""
import numpy as np
import torch

from tests import helpers


def _torch_forward(oracle_out, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, bg):
    """Differentiable float64 colour image from the same inputs.  Uses the oracle's radii / means2D
    only to freeze tile coverage."""
    P = xyz.shape[0]
    V, Pm = view.reshape(4, 4), full.reshape(4, 4)            # transposed (row-vector) matrices
    ones = torch.ones(P, 1, dtype=torch.float64)
    ph = torch.cat((xyz, ones), 1)
    p_view = ph @ V
    p_hom = ph @ Pm
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    pix = torch.stack((((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5), 1)
    r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    R = torch.stack((torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)), 1),
                     torch.stack((2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)), 1),
                     torch.stack((2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), 1)), 1)
    Sigma = R @ torch.diag_embed(sc * sc) @ R.transpose(1, 2)
    fx, fy = W / (2 * tfx), H / (2 * tfy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tfx, 1.3 * tfy
    # the reference's backward treats a clamped t.x / t.y as a constant (x_grad_mul / y_grad_mul = 0,
    # backward.cu:174-175,259-261) — it does not propagate d(lim * t.z)/d t.z; mirrored here
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    tx = torch.where(txtz.abs() <= limx, p_view[:, 0], (torch.clamp(txtz, -limx, limx) * tz).detach())
    ty = torch.where(tytz.abs() <= limy, p_view[:, 1], (torch.clamp(tytz, -limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack((torch.stack((fx / tz, zero, -fx * tx / tz ** 2), 1),
                     torch.stack((zero, fy / tz, -fy * ty / tz ** 2), 1)), 1)          # (P,2,3)
    Rv = V[:3, :3].T                                                                  # world->view rotation
    M = J @ Rv
    cov = M @ Sigma @ M.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    con = torch.stack((c / det, -b / det, a / det), 1)
    radii = torch.from_numpy(oracle_out['radii'])
    vis = radii > 0
    order = torch.from_numpy(np.lexsort((np.arange(P), oracle_out['depths'].view(np.uint32))))
    order = order[vis[order]]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    m2 = torch.from_numpy(oracle_out['means2D']).double()
    x0 = torch.clamp(((m2[:, 0] - radii) / 16).trunc(), 0, gx)
    x1 = torch.clamp(((m2[:, 0] + radii + 15) / 16).trunc(), 0, gx)
    y0 = torch.clamp(((m2[:, 1] - radii) / 16).trunc(), 0, gy)
    y1 = torch.clamp(((m2[:, 1] + radii + 15) / 16).trunc(), 0, gy)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing='ij')
    T = torch.ones(H, W, dtype=torch.float64)
    alive = torch.ones(H, W, dtype=torch.bool)
    C = torch.zeros(3, H, W, dtype=torch.float64)
    for i in order.tolist():
        cover = (xs // 16 >= x0[i]) & (xs // 16 < x1[i]) & (ys // 16 >= y0[i]) & (ys // 16 < y1[i])
        dx, dy = pix[i, 0] - xs, pix[i, 1] - ys
        power = -0.5 * (con[i, 0] * dx * dx + con[i, 2] * dy * dy) - con[i, 1] * dx * dy
        G = torch.exp(power)
        raw = opac[i, 0] * G
        alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()              # clamp value, unclamped derivative
        ok = cover & alive & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        stop = ok & (test_T.detach() < 1e-4)
        alive = alive & ~stop
        use = ok & ~stop
        w = torch.where(use, alpha * T, torch.zeros_like(T))
        C = C + rgb[i][:, None, None] * w[None]
        T = torch.where(use, test_T, T)
    return C + T[None] * bg[:, None, None]


def test_backward_matches_float64_autograd(oracle_lib):
    rng = np.random.default_rng(5)
    W, H, n = 48, 32, 60
    view, full, tfx, tfy = helpers.simple_camera(W, H, cam_pos=(0.3, -0.2, -1.0))
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, n, z_range=(2.0, 12.0), xy_extent=4.0, scale=(0.1, 0.6))
    rot = rot * rng.uniform(0.8, 1.2, (n, 1)).astype(np.float32)       # not unit: the kernel does not normalise
    bg = np.float32([0.2, 0.5, 0.1])
    gcol = rng.standard_normal((3, H, W)).astype(np.float32)
    fwd = oracle_lib.rasterize_forward(xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, bg)
    got = oracle_lib.rasterize_backward(gcol, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, bg)
    t = [torch.from_numpy(a).double().requires_grad_() for a in (xyz, rgb, opac, sc, rot)]
    img = _torch_forward(fwd, *t, torch.from_numpy(view).double(), torch.from_numpy(full).double(), tfx, tfy, H, W,
                         torch.from_numpy(bg).double())
    np.testing.assert_allclose(img.detach().numpy(), fwd['color'], atol=2e-5)        # the two forwards agree
    (img * torch.from_numpy(gcol).double()).sum().backward()
    for name, ten, tol in (('means3D', t[0], 1e-4), ('colors', t[1], 1e-5), ('opacities', t[2], 1e-5),
                           ('scales', t[3], 1e-4), ('rotations', t[4], 1e-4)):
        want = ten.grad.numpy().reshape(got[name].shape)
        scale = np.abs(want).max()
        assert scale > 0
        err = np.abs(got[name] - want).max() / scale
        assert err < tol, f'{name}: relative error {err:.2e}'
