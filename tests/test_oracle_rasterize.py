"""CPU: the C oracle of the rasteriser forward (oracle/rasterize_ref.c).  The reference holds no
golden vectors for the rasteriser and its CUDA sources cannot be built here, so the oracle is
checked against closed-form renders and an independent per-pixel numpy evaluation (PARITY
UNPINNED against a reference binary — DESIGN.md "Oracle")."""
import numpy as np

from tests import helpers


def _render(oracle_lib, xyz, rgb, opac, scales, rot, W=64, H=48, **kw):
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    return oracle_lib.rasterize_forward(xyz, rgb, opac, scales, rot, view, full, tfx, tfy, H, W,
                                        np.zeros(3, np.float32), **kw), (W, H, 0.8 * W)


def test_single_isotropic_gaussian_closed_form(oracle_lib):
    z0, s, o = 5.0, 0.25, 0.7
    xyz = np.float32([[0.0, 0.0, z0]])        # on the optical axis: the EWA Jacobian is diagonal
    out, (W, H, f) = _render(oracle_lib, xyz, np.float32([[0.2, 0.5, 0.9]]), np.float32([[o]]),
                             np.float32([[s, s, s]]), np.float32([[1, 0, 0, 0]]))
    var = (f * s / z0) ** 2 + 0.3                       # EWA + 0.3 px low-pass, forward.cu:110-111
    mx = f * xyz[0, 0] / z0 + W / 2.0 - 0.5             # ndc2Pix: pixel centres at integers
    my = f * xyz[0, 1] / z0 + H / 2.0 - 0.5
    np.testing.assert_allclose(out['means2D'][0], [mx, my], rtol=0, atol=2e-4)
    # eigenvalues: mid +- sqrt(max(0.1, mid^2 - det)) (forward.cu:228-230); isotropic -> floor 0.1
    assert out['radii'][0] == int(np.ceil(3 * np.sqrt(var + np.sqrt(0.1))))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    alpha = np.minimum(0.99, o * np.exp(-0.5 * ((xx - mx) ** 2 + (yy - my) ** 2) / var))
    # only tiles inside the radius-derived rectangle are touched (auxiliary.h:46-56)
    r = out['radii'][0]
    tx0, tx1 = int((mx - r) / 16), int((mx + r + 15) / 16)
    ty0, ty1 = int((my - r) / 16), int((my + r + 15) / 16)
    cover = np.zeros((H, W), bool)
    cover[ty0 * 16:ty1 * 16, tx0 * 16:tx1 * 16] = True
    alpha = np.where(cover & (alpha >= 1 / 255), alpha, 0.0)
    np.testing.assert_allclose(out['final_T'], 1 - alpha, atol=3e-6)
    for ch, c in enumerate((0.2, 0.5, 0.9)):
        np.testing.assert_allclose(out['color'][ch], c * alpha, atol=3e-6)
    depth = np.where(1 - alpha < 0.5, z0, 15.0)          # median depth, default 15 (README:5-11)
    np.testing.assert_allclose(out['depth'][0], depth, atol=1e-6)
    assert out['num_rendered'] == (tx1 - tx0) * (ty1 - ty0) == out['tiles_touched'][0]
    assert out['n_contrib'].max() == 1


def test_two_gaussians_composite_front_to_back(oracle_lib):
    xyz = np.float32([[0, 0, 8.0], [0, 0, 4.0]])        # index 1 is nearer: must be blended first
    rgb = np.float32([[1, 0, 0], [0, 1, 0]])
    opac = np.float32([[0.9], [0.6]])
    sc = np.full((2, 3), 0.5, np.float32)
    rot = np.float32([[1, 0, 0, 0]] * 2)
    out, (W, H, f) = _render(oracle_lib, xyz, rgb, opac, sc, rot)
    cy, cx = H // 2, W // 2
    a_near = min(0.99, 0.6 * np.exp(-0.5 * (0.5 ** 2 + 0.5 ** 2) / ((f * 0.5 / 4) ** 2 + 0.3)))
    a_far = min(0.99, 0.9 * np.exp(-0.5 * (0.5 ** 2 + 0.5 ** 2) / ((f * 0.5 / 8) ** 2 + 0.3)))
    np.testing.assert_allclose(out['color'][:, cy, cx], [a_far * (1 - a_near), a_near, 0], atol=2e-5)
    np.testing.assert_allclose(out['final_T'][cy, cx], (1 - a_near) * (1 - a_far), atol=2e-5)
    assert out['depth'][0, cy, cx] == np.float32(4.0)    # T crosses 0.5 at the near one (a=0.6)
    mean = _render(oracle_lib, xyz, rgb, opac, sc, rot, depth_mode='mean')[0]
    np.testing.assert_allclose(mean['depth'][0, cy, cx], 4.0 * a_near + 8.0 * a_far * (1 - a_near), rtol=1e-5)


def test_culling_and_empty(oracle_lib):
    xyz = np.float32([[0, 0, 0.1], [0, 0, -3.0], [500.0, 0, 5.0]])     # near-culled, behind, off-screen
    out, _ = _render(oracle_lib, xyz, np.ones((3, 3), np.float32), np.full((3, 1), 0.9, np.float32),
                     np.full((3, 3), 0.3, np.float32), np.float32([[1, 0, 0, 0]] * 3))
    assert out['num_rendered'] == 0 and not out['radii'].any()
    assert (out['final_T'] == 1).all() and (out['depth'] == 15).all() and not out['color'].any()
    empty, _ = _render(oracle_lib, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32),
                       np.zeros((0, 1), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 4), np.float32))
    assert not empty['color'].any() and not empty['final_T'].any()      # rasterize_points.cu:68-69


def test_random_scene_against_independent_numpy(oracle_lib):
    """Per-pixel re-evaluation in float64 from the oracle's own per-Gaussian state: checks the
    sort order, tile coverage, thresholds and compositing independently of the C loop."""
    rng = np.random.default_rng(0)
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, 300)
    out, (W, H, _) = _render(oracle_lib, xyz, rgb, opac, sc, rot)
    vis = np.nonzero(out['radii'] > 0)[0]
    order = vis[np.lexsort((vis, out['depths'][vis].view(np.uint32)))]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    bad = 0
    for py in range(0, H, 3):
        for px in range(0, W, 3):
            T, C, D = 1.0, np.zeros(3), 15.0
            for i in order:
                mx, my = out['means2D'][i]
                r = int(out['radii'][i])
                x0, x1 = min(gx, max(0, int((mx - r) / 16))), min(gx, max(0, int((mx + r + 15) / 16)))
                y0, y1 = min(gy, max(0, int((my - r) / 16))), min(gy, max(0, int((my + r + 15) / 16)))
                if not (x0 <= px // 16 < x1 and y0 <= py // 16 < y1):
                    continue
                a, b, c, o = out['conic_opacity'][i].astype(np.float64)
                dx, dy = mx - px, my - py
                power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
                if power > 0:
                    continue
                alpha = min(0.99, o * np.exp(power))
                if alpha < 1 / 255:
                    continue
                if T * (1 - alpha) < 1e-4:
                    break
                C += rgb[i] * alpha * T
                if T > 0.5 and T * (1 - alpha) < 0.5:
                    D = out['depths'][i]
                T *= 1 - alpha
            ok = abs(out['final_T'][py, px] - T) < 1e-5 and np.abs(out['color'][:, py, px] - C).max() < 1e-5 \
                and out['depth'][0, py, px] == np.float32(D)
            bad += not ok
    assert bad <= 1        # a float32/float64 threshold flip on one sampled pixel is tolerated
