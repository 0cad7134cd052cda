"""GPU parity: HIP rasteriser forward (C ABI ocrf_rasterize_forward, via the reference-shaped
Python surface) vs the C oracle on identical inputs.

Bars: per-Gaussian integer state (radii, tiles touched) bit-exact; per-Gaussian float state
(means2D, conic, depth) bit-exact (same expression order, fp-contract off on both sides);
rendered colour / final_T (opacity = 1 - final_T) / mean depth within 1e-4 (north_star).  The
blend evaluates exp() with the GPU's instruction where the oracle uses libm's expf, so a pixel
whose alpha sits within an ulp of one of the algorithm's thresholds (1/255 skip, T<1e-4 stop,
T crossing 0.5 for the median depth) may decide differently; such pixels are bounded explicitly
below instead of being hidden by a loose tolerance."""
import math

import numpy as np
import pytest
import torch

from ocrfdet_amd import diff_gaussian_rasterization as dgr
from ocrfdet_amd import gaussian_renderer as gr
from ocrfdet_amd import synthetic
from tests import helpers

pytestmark = pytest.mark.gpu
TOL = 1e-4
# Relative distance to a decision threshold below which another exp implementation may decide the other way: the alpha
# cut / clamp see one exp (v_exp_f32: 1 ulp, + the log2(e) multiply: ~4e-6 relative); T is a product of up to a few
# hundred (1 - alpha) factors, each off by that much (2e-4 relative).  The oracle marks such pixels ("ambiguous").
AMBIGUITY = (4e-6, 2e-4)
MEASURED = []            # (label, pixels, pixels off by > TOL, ambiguous pixels) of every comparison, printed at the end


def _t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _both(oracle_lib, cuda, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, bg=(0, 0, 0), depth_mode='median'):
    want = oracle_lib.rasterize_forward(xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W,
                                        np.float32(bg), depth_mode=depth_mode, ambiguity=AMBIGUITY)
    got = dgr.rasterize_views(_t(xyz, cuda), _t(rgb, cuda), _t(opac, cuda), _t(sc, cuda), _t(rot, cuda),
                              _t(view, cuda).view(1, 4, 4), _t(full, cuda).view(1, 4, 4), [tfx], [tfy], H, W,
                              _t(np.float32(bg), cuda), depth_mode=depth_mode, want_tiles_touched=True)
    torch.cuda.synchronize()
    assert int(got['status'].item()) & 1 == 0
    return want, {k: v.cpu().numpy() for k, v in got.items()}


def _compare(want, got, H, W, max_outlier_frac=2e-5, label=''):
    """Per-Gaussian integer state bit-exact; images within TOL except at pixels where the ORACLE ITSELF sits within
    rounding of a decision threshold (``want['ambiguous']``, oracle/rasterize_ref.c): every pixel that differs by more
    than TOL must be one of those, their number stays under ``max_outlier_frac`` of the image, and a flip changes a
    pixel by at most one skipped contribution."""
    np.testing.assert_array_equal(got['radii'][0], want['radii'])
    np.testing.assert_array_equal(got['tiles_touched'][0].astype(np.uint32), want['tiles_touched'])
    dc = np.abs(got['color'][0] - want['color']).max(0)
    dt = np.abs(got['final_T'][0] - want['final_T'])
    bad = (dc > TOL) | (dt > TOL)
    n_bad = int(bad.sum())
    assert n_bad <= max(2, max_outlier_frac * H * W), f'{n_bad} pixels off by more than {TOL}'
    # a threshold flip changes a pixel by at most one skipped contribution: alpha <= 1/255 + ulp
    assert dc.max() < 5e-3 and dt.max() < 5e-3
    dd = np.abs(got['depth'][0, 0] - want['depth'][0])
    assert int((dd > TOL).sum()) <= max(2, max_outlier_frac * H * W)
    nc = got['n_contrib'][0].astype(np.int64) != want['n_contrib'].astype(np.int64)
    assert int(nc.sum()) <= max(2, max_outlier_frac * H * W)
    if 'ambiguous' in want:
        amb = want['ambiguous'].astype(bool)
        off = bad | (dd > TOL) | nc
        stray = off & ~amb
        assert int(stray.sum()) == 0, (f'{int(stray.sum())} pixels differ from the oracle without any of their decisions '
                                       'lying at a threshold')
        MEASURED.append((label, H * W, int(off.sum()), int(amb.sum())))
        print(f'[rasteriser parity] {label or "scene"}: {int(off.sum())} of {H * W} pixels off by > {TOL} '
              f'({int(bad.sum())} colour / T, {int((dd > TOL).sum())} depth, {int(nc.sum())} n_contrib), all among the '
              f'{int(amb.sum())} pixels the oracle marks as threshold-ambiguous')
    return n_bad


def test_single_gaussian_matches_oracle_exactly_in_state(cuda, oracle_lib):
    W, H = 64, 48
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    xyz = np.float32([[0.3, -0.2, 5.0]])
    want, got = _both(oracle_lib, cuda, xyz, np.float32([[0.2, 0.5, 0.9]]), np.float32([[0.7]]),
                      np.float32([[0.3, 0.2, 0.25]]), np.float32([[0.9, 0.1, -0.3, 0.2]]), view, full, tfx, tfy, H, W,
                      bg=(0.1, 0.2, 0.3))
    _compare(want, got, H, W, 0)
    assert got['color'][0][:, 0, 0] == pytest.approx([0.1, 0.2, 0.3])      # background where T = 1


@pytest.mark.parametrize('opacity,n_stack', [(0.99, 3), (0.995, 4), (0.97, 6), (0.9, 12)])
def test_stacked_opaque_gaussians_stop_where_the_reference_stops(cuda, oracle_lib, opacity, n_stack):
    """Consecutive records of opacity ~0.99 centred on one pixel: T goes 1 -> 0.01 -> 1e-4 -> ... inside ONE
    trip of the blend loop, so the stop test (forward.cu:340-345: T (1 - alpha) < 1e-4 -> done, record not blended) must
    be decided per record.  final_T and colour against the oracle, which loops like the reference."""
    W, H = 64, 48
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    rng = np.random.default_rng(17)
    # a stack on the optical axis (pixel centre 32, 24), 2 cm apart in depth, then a bright wall behind it
    xyz = [[0.0, 0.0, 4.0 + 0.02 * i] for i in range(n_stack)] + [[0.0, 0.0, 6.0]]
    # ... beside another stack whose opacities alternate low / high (pairs of a trip with very different bounds)
    xyz += [[0.8, 0.3, 3.0 + 0.02 * i] for i in range(2 * n_stack)]
    n = len(xyz)
    opac = [opacity] * n_stack + [0.9] + [0.2 if i % 2 else opacity for i in range(2 * n_stack)]
    sc = np.full((n, 3), 0.25, np.float32)
    sc[n_stack] = 2.0
    q = np.tile(np.float32([1, 0, 0, 0]), (n, 1))
    rgb = rng.uniform(0.2, 1.0, (n, 3)).astype(np.float32)
    for mode in ('median', 'mean'):
        want, got = _both(oracle_lib, cuda, np.float32(xyz), rgb, np.float32(opac).reshape(-1, 1), sc, q, view, full, tfx,
                          tfy, H, W, bg=(0.3, 0.6, 0.9), depth_mode=mode)
        _compare(want, got, H, W, label=f'stack of {n_stack} x {opacity}')
        # the centre pixel of the stack: the transmittance the reference leaves, not 100 x less
        np.testing.assert_allclose(got['final_T'][0][24, 32], want['final_T'][24, 32], rtol=1e-3, atol=1e-7)
        # ... and the planned render (its own loop over the same shared arithmetic) agrees bit for bit
        from ocrfdet_amd import raster_plan as rp
        cams = dgr.pack_cameras(_t(view, cuda).view(1, 4, 4), _t(full, cuda).view(1, 4, 4), [tfx], [tfy], H, W, cuda)
        plan = rp.RasterPlan(_t(np.float32(xyz), cuda), cams, H, W, extent_bound=4.0)
        pl = plan.render(_t(rgb, cuda), _t(np.float32(opac).reshape(-1, 1), cuda), _t(sc, cuda), _t(q, cuda),
                         _t(np.float32((0.3, 0.6, 0.9)), cuda), depth_mode=mode)
        torch.cuda.synchronize()
        for key in ('color', 'depth', 'final_T'):
            np.testing.assert_array_equal(pl[key].cpu().numpy().reshape(got[key].shape), got[key])


@pytest.mark.parametrize('n,seed', [(300, 0), (5000, 1), (20000, 2)])
def test_random_scene_parity(cuda, oracle_lib, n, seed):
    rng = np.random.default_rng(seed)
    W, H = 176, 64           # 11 x 4 tiles
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, n)
    for mode in ('median', 'mean'):
        want, got = _both(oracle_lib, cuda, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, depth_mode=mode)
        _compare(want, got, H, W)


def test_reference_shape_ocrf_grid_and_camera_convention(cuda, oracle_lib):
    """The OcRF voxel-grid Gaussians (13 x 128 x 128) seen through the reference's own camera
    set-up (view_transformer_ocrf.py:1135-1152, quirks included) at 256 x 704."""
    from oracle import index_prep as oip
    cfg = synthetic.CONFIGS['ref_6cam_256x704_bev128x128x1']
    r = synthetic.rig(6, cfg.input_size, 1)
    X, Y, _ = cfg.bev_xyz
    ref = oip.get_reference_points_3d(Y, X, bs=1, num_points_in_pillar=13)
    l2i, aug = oip.get_projection(r['rots'], r['trans'], r['intrins'], r['post_rots'], r['post_trans'], r['bda'])
    _, _, voxel = oip.get_sampling_point(ref, cfg.pc_range, cfg.grid['depth'], l2i, aug, cfg.input_size)
    xyz = voxel.reshape(-1, 3).astype(np.float32)
    rng = np.random.default_rng(0)
    P = xyz.shape[0]
    sc = rng.uniform(0.69, 0.84, (P, 3)).astype(np.float32)       # seeded-init MLP ranges
    q = rng.standard_normal((P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.3, 0.5, (P, 1)).astype(np.float32)
    rgb = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    H, W = cfg.input_size
    cam = gr.camera_from_calibration(r['intrins'][0, 1], r['c2w'][0, 1], H, W)
    view, full = cam['world_view_transform'].numpy(), cam['full_proj_transform'].numpy()
    tfx, tfy = math.tan(float(cam['FovX']) * 0.5), math.tan(float(cam['FovY']) * 0.5)
    want, got = _both(oracle_lib, cuda, xyz, rgb, opac, sc, q, view, full, tfx, tfy, H, W)
    assert want['num_rendered'] > 1000
    # Round 5: the blend evaluates the exponent of a SIMPLE record as two packed FMAs over a conic pre-multiplied by
    # log2(e) with log2(opacity) as the constant term (csrc/raster_blend_math.h) instead of in forward.cu's order.  The
    # quadratic form cancels (its three terms are several times its value), so BOTH orders carry ~1e-6..1e-5 of relative
    # rounding in alpha — they just carry different ones (as does nvcc's own FMA contraction of forward.cu:320-323), and
    # a few more of the pixels whose T lands within that distance of 0.5 / 1e-4 decide the other way: 6 of 180 224 here
    # (3.3e-5), every one of them inside the oracle's threshold-ambiguity map.  Cap: 5e-5 of the image.
    _compare(want, got, H, W, max_outlier_frac=5e-5, label='reference shape 256x704')


def test_one_depth_bucket_larger_than_the_lds_sort(cuda, oracle_lib):
    """8 000 Gaussians at EXACTLY the same depth over the same tiles: more than the in-LDS sort holds
    for one 0.2 % depth bucket, so the blend kernel must take its exact streaming-selection path
    (status bit 1) and still reproduce the reference order (ties by Gaussian id)."""
    rng = np.random.default_rng(9)
    W, H = 64, 48
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    n = 8000
    xyz = np.stack([rng.uniform(-0.4, 0.4, n), rng.uniform(-0.3, 0.3, n), np.full(n, 5.0)], 1).astype(np.float32)
    extra = helpers.random_gaussians(rng, 300)
    xyz = np.concatenate((xyz, extra[0]))
    rgb = np.concatenate((rng.uniform(0, 1, (n, 3)).astype(np.float32), extra[1]))
    opac = np.concatenate((rng.uniform(0.004, 0.02, (n, 1)).astype(np.float32), extra[2]))
    sc = np.concatenate((rng.uniform(0.2, 0.5, (n, 3)).astype(np.float32), extra[3]))
    rot = np.concatenate((np.tile(np.float32([[1, 0, 0, 0]]), (n, 1)), extra[4]))
    perm = rng.permutation(len(xyz))                      # ids of the tied Gaussians are not contiguous
    xyz, rgb, opac, sc, rot = xyz[perm], rgb[perm], opac[perm], sc[perm], rot[perm]
    want = oracle_lib.rasterize_forward(xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, np.zeros(3, np.float32))
    got = dgr.rasterize_views(_t(xyz, cuda), _t(rgb, cuda), _t(opac, cuda), _t(sc, cuda), _t(rot, cuda),
                              _t(view, cuda).view(1, 4, 4), _t(full, cuda).view(1, 4, 4), [tfx], [tfy], H, W,
                              torch.zeros(3, device=cuda), want_tiles_touched=True)
    torch.cuda.synchronize()
    assert int(got['status'].item()) == 2
    assert want['n_contrib'].max() > 4096                  # pixels really consume more than one LDS load
    _compare(want, {k: v.cpu().numpy() for k, v in got.items()}, H, W)


def test_render_api_matches_reference_call_shape(cuda, oracle_lib):
    """render(data, idx, xyz, rgb, rot, scales, opacity, bg) -> (image (3,H,W), depth (1,H,W))
    (gaussian_renderer/__init__.py:17-75) and the multi-view batch agree with single calls."""
    rng = np.random.default_rng(3)
    W, H = 96, 64
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, 2000)
    cams = []
    for pos in ((0, 0, 0), (1.0, 0.2, -1.0), (-2.0, 0.0, 0.5)):
        view, full, tfx, tfy = helpers.simple_camera(W, H, cam_pos=pos)
        cams.append(dict(FovX=2 * math.atan(tfx), FovY=2 * math.atan(tfy), height=H, width=W,
                         world_view_transform=_t(view, cuda), full_proj_transform=_t(full, cuda),
                         camera_center=_t(np.float32(pos), cuda)))
    args = [_t(a, cuda) for a in (xyz, rgb, rot, sc, opac)]
    batch = gr.render_views(cams, *args, [0, 0, 0], H, W)
    for i, cam in enumerate(cams):
        img, dep = gr.render(cam, 0, *args, bg_color=[0, 0, 0])
        assert img.shape == (3, H, W) and dep.shape == (1, H, W)
        torch.testing.assert_close(img, batch['color'][i], rtol=0, atol=0)
        torch.testing.assert_close(dep, batch['depth'][i], rtol=0, atol=0)
        want = oracle_lib.rasterize_forward(xyz, rgb, opac, sc, rot, cam['world_view_transform'].cpu().numpy(),
                                            cam['full_proj_transform'].cpu().numpy(), math.tan(cam['FovX'] * 0.5),
                                            math.tan(cam['FovY'] * 0.5), H, W, np.zeros(3, np.float32))
        assert np.abs(img.cpu().numpy() - want['color']).max() < 5e-3
        assert (np.abs(img.cpu().numpy() - want['color']).max(0) > TOL).sum() <= 2


def test_argument_errors_and_empty(cuda):
    s = dgr.GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3, device=cuda), 1.0,
                                          torch.eye(4, device=cuda), torch.eye(4, device=cuda), 3,
                                          torch.zeros(3, device=cuda), False)
    rast = dgr.GaussianRasterizer(s)
    x = torch.zeros(4, 3, device=cuda)
    with pytest.raises(Exception, match='SHs or precomputed colors'):
        rast(x, None, torch.ones(4, 1, device=cuda), scales=x, rotations=torch.zeros(4, 4, device=cuda))
    with pytest.raises(Exception, match='scale/rotation pair or precomputed 3D covariance'):
        rast(x, None, torch.ones(4, 1, device=cuda), colors_precomp=x)
    color, radii, depth = rast(torch.zeros(0, 3, device=cuda), None, torch.zeros(0, 1, device=cuda),
                               colors_precomp=torch.zeros(0, 3, device=cuda), scales=torch.zeros(0, 3, device=cuda),
                               rotations=torch.zeros(0, 4, device=cuda))
    assert color.shape == (3, 8, 8) and float(color.abs().sum()) == 0 and radii.numel() == 0


def test_gaussian_sets_match_per_set_calls(cuda):
    """ocrf_rasterize_forward_sets: S sets x V views in one call == S separate calls, bit for bit."""
    import torch
    from ocrfdet_amd.diff_gaussian_rasterization import pack_cameras, rasterize_sets, rasterize_views
    rng = np.random.default_rng(11)
    S, V, P, H, W = 3, 2, 3000, 80, 112
    sets = [helpers.random_gaussians(rng, P) for _ in range(S)]
    cams = []
    for v in range(V):
        vm, pm, tfx, tfy = helpers.simple_camera(W, H, cam_pos=(0.5 * v, 0.0, -1.0 * v))
        cams.append((vm, pm, tfx, tfy))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)   # noqa: E731
    packed = pack_cameras(t(np.stack([c[0] for c in cams])), t(np.stack([c[1] for c in cams])),
                          [c[2] for c in cams], [c[3] for c in cams], H, W, cuda)
    bg = torch.tensor([0.1, 0.2, 0.3], device=cuda)
    st = [t(np.stack([s[i] for s in sets])) for i in range(5)]             # xyz, rgb, opac, scales, q
    got = rasterize_sets(st[0], st[1], st[2], st[3], st[4], packed.repeat(S, 1), H, W, bg, want_n_contrib=True)
    for s in range(S):
        want = rasterize_views(st[0][s], st[1][s], st[2][s], st[3][s], st[4][s], None, None, None, None, H, W, bg,
                               packed_cameras=packed)
        for k in ('color', 'depth', 'final_T', 'n_contrib', 'radii'):
            assert torch.equal(got[k][s * V:(s + 1) * V], want[k]), (s, k)


def test_full_size_properties_of_the_bench_scene(cuda):
    """BASELINE configs[2] at full size (520 000 Gaussians, 6 views of 256x704 — more than the C oracle renders in
    test time): properties that hold bit for bit whatever the size.  (1) with a zero background the colour is
    linear in the Gaussians' colours, and a factor of two is exact in binary floating point: image(2 c) == 2
    image(c), depth / final_T / n_contrib / radii unchanged; (2) a view's image does not depend on which other
    views share the launch or on their order; (3) two launches of the same inputs agree bitwise."""
    from ocrfdet_amd import hotpath
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    hp = hotpath.HotPath(cfg, cuda)
    g, rc = hp.gauss, hp.render_cams
    H, W = cfg.input_size
    xyz = hp.voxel_xyz[0].reshape(-1, 3)
    assert xyz.shape[0] == 13 * 200 * 200
    zero = torch.zeros(3, device=cuda)

    def run(rgb, packed):
        return dgr.rasterize_views(xyz, rgb, g['opacity'], g['scales'], g['rotations'], None, None, None, None, H, W,
                                   zero, packed_cameras=packed)
    a = run(g['rgb'], rc['packed'])
    b = run(g['rgb'], rc['packed'])
    for k in ('color', 'depth', 'final_T', 'n_contrib', 'radii'):
        assert torch.equal(a[k], b[k]), f'{k} differs between two identical launches'
    assert int(a['status'].item()) & 1 == 0
    assert float(a['final_T'].min()) < 0.5 and float(a['color'].max()) > 0.0          # the scene is not empty
    c = run(g['rgb'] * 2.0, rc['packed'])
    assert torch.equal(c['color'], a['color'] * 2.0)
    for k in ('depth', 'final_T', 'n_contrib', 'radii'):
        assert torch.equal(c[k], a[k]), k
    perm = torch.tensor([4, 2, 5, 0, 3, 1], device=cuda)
    d = run(g['rgb'], rc['packed'][perm].contiguous())
    for k in ('color', 'depth', 'final_T', 'n_contrib', 'radii'):
        assert torch.equal(d[k], a[k][perm]), f'{k} depends on the order of the views'
    e = run(g['rgb'], rc['packed'][3:4].contiguous())
    for k in ('color', 'depth', 'final_T', 'n_contrib', 'radii'):
        assert torch.equal(e[k][0], a[k][3]), f'{k} depends on the other views of the launch'
    # (4) the inference variant that does not track the contributor index renders the same images
    f = dgr.rasterize_views(xyz, g['rgb'], g['opacity'], g['scales'], g['rotations'], None, None, None, None, H, W, zero,
                            packed_cameras=rc['packed'], want_n_contrib=False)
    assert 'n_contrib' not in f
    for k in ('color', 'depth', 'final_T', 'radii'):
        assert torch.equal(f[k], a[k]), k


def test_opacity_edge_values(cuda, oracle_lib):
    """Opacities at the edges of what the per-record skip threshold of the blend assumes: exactly 0, far below and
    right around the 1/255 alpha cut, 1, and above 1 (the op accepts any float; alpha is then clamped at 0.99) — both
    kernel variants (with and without the contributor index) against the oracle."""
    rng = np.random.default_rng(12)
    n = 6000
    W, H = 176, 96
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, n)
    edge = np.float32([0.0, 1e-6, 1e-3, 1.0 / 255.0 * 0.999, 1.0 / 255.0, 1.0 / 255.0 * 1.001, 0.01, 1.0, 1.5, 3.0])
    opac = edge[rng.integers(0, edge.size, n)].reshape(n, 1).astype(np.float32)
    want, got = _both(oracle_lib, cuda, xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, bg=(0.05, 0.1, 0.15))
    _compare(want, got, H, W)
    inf = dgr.rasterize_views(_t(xyz, cuda), _t(rgb, cuda), _t(opac, cuda), _t(sc, cuda), _t(rot, cuda),
                              _t(view, cuda).view(1, 4, 4), _t(full, cuda).view(1, 4, 4), [tfx], [tfy], H, W,
                              _t(np.float32((0.05, 0.1, 0.15)), cuda), want_n_contrib=False)
    torch.cuda.synchronize()
    for k in ('color', 'depth', 'final_T'):                 # the inference variant: bit-identical to the other one
        np.testing.assert_array_equal(inf[k].cpu().numpy(), got[k])
