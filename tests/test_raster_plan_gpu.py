"""GPU parity of the static render plans (csrc/raster_plan.hip, ocrfdet_amd/raster_plan.py).

The planned render must reproduce the per-call pipeline (``rasterize_views``, itself checked against the C
oracle in tests/test_rasterize_gpu.py) BIT FOR BIT in colour, depth and final_T, and in radii: it blends the same
records in the same order with the same arithmetic; what it leaves out (Gaussians culled statically, records a
wave's pixel block cannot reach) contributes exactly nothing in the reference either
(forward.cu:166-171,236-238,331-333)."""
import math

import numpy as np
import pytest
import torch

from ocrfdet_amd import diff_gaussian_rasterization as dgr
from ocrfdet_amd import gaussian_renderer as gr
from ocrfdet_amd import raster_plan as rp
from ocrfdet_amd import synthetic
from tests import helpers
from tests.test_rasterize_gpu import _compare

pytestmark = pytest.mark.gpu


def _t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _scene(rng, n, cuda, **kw):
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, n, **kw)
    return tuple(_t(a, cuda) for a in (xyz, rgb, opac, sc, rot))


def _cams(cuda, W, H, positions):
    vms, pms, tfx, tfy = [], [], [], []
    for p in positions:
        view, full, tx, ty = helpers.simple_camera(W, H, cam_pos=p)
        vms.append(torch.from_numpy(view)), pms.append(torch.from_numpy(full))
        tfx.append(tx), tfy.append(ty)
    return dgr.pack_cameras(torch.stack(vms).to(cuda), torch.stack(pms).to(cuda), tfx, tfy, H, W, cuda)


def _same(a, b, keys=('color', 'depth', 'final_T')):
    for k in keys:
        assert torch.equal(a[k], b[k]), f'{k} differs: max |d| = {(a[k].float() - b[k].float()).abs().max().item()}'


@pytest.mark.parametrize('n,seed', [(1, 0), (300, 1), (5000, 2), (40000, 3)])
@pytest.mark.parametrize('mode', ['median', 'mean'])
def test_planned_render_is_bit_identical_to_the_per_call_pipeline(cuda, n, seed, mode):
    rng = np.random.default_rng(seed)
    W, H = 176, 80                      # 11 x 5 tiles: the last tile pair has no lower tile
    xyz, rgb, opac, sc, rot = _scene(rng, n, cuda)
    cams = _cams(cuda, W, H, [(0, 0, 0), (1.5, -0.5, 2.0), (-3.0, 0.4, -1.0)])
    bg = torch.tensor([0.1, 0.2, 0.3], device=cuda)
    want = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, depth_mode=mode,
                               packed_cameras=cams, want_n_contrib=False)
    plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot)
    got = plan.render(rgb, opac, sc, rot, bg, depth_mode=mode, want_radii=True)
    torch.cuda.synchronize()
    assert plan.check()
    _same(want, got, ('color', 'depth', 'final_T', 'radii'))
    assert sum(plan.kept) <= 3 * n


def _head_mode(n):
    """ocrf_tune_set(13, n): list entries per view the head kernel prepares — n > 0 fixed, n < 0 none (every record is
    computed by the tile pair that scans to it), 0 adaptive (the product default)."""
    from ocrfdet_amd import _lib
    _lib.check(_lib.lib().ocrf_tune_set(13, int(n)), 'ocrf_tune_set')


@pytest.mark.parametrize('n,seed', [(300, 1), (5000, 2), (40000, 3)])
@pytest.mark.parametrize('mode', ['median', 'mean'])
def test_head_of_the_list_however_long_gives_the_same_image(cuda, n, seed, mode):
    """Without radii a render prepares only the HEAD of every view's list in front of the blend (raster_plan_head_kernel);
    a tile pair that scans further extends the arrays itself.  Whatever the split — nothing prepared, 256 entries, the
    adaptive default (twice: the second call uses what the first one reached), everything (the full update, which
    `want_radii` takes) — the images are the per-call pipeline's, bit for bit."""
    rng = np.random.default_rng(seed)
    W, H = 176, 80
    xyz, rgb, opac, sc, rot = _scene(rng, n, cuda)
    cams = _cams(cuda, W, H, [(0, 0, 0), (1.5, -0.5, 2.0), (-3.0, 0.4, -1.0)])
    bg = torch.tensor([0.1, 0.2, 0.3], device=cuda)
    want = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, depth_mode=mode,
                               packed_cameras=cams, want_n_contrib=False)
    plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot)
    try:
        for head in (-1, 256, 0, 0, 1 << 20):
            _head_mode(head)
            got = plan.render(rgb, opac, sc, rot, bg, depth_mode=mode)
            _same(want, got)
    finally:
        _head_mode(0)
    _same(want, plan.render(rgb, opac, sc, rot, bg, depth_mode=mode, want_radii=True))
    torch.cuda.synchronize()
    assert plan.check()


def test_head_path_checks_extent_cameras_and_views(cuda):
    """The head kernel carries the call's checks when no radii are asked: extent bound (status bit 4), the call's cameras
    against the plan's (bit 16), a view named twice (bit 8)."""
    rng = np.random.default_rng(13)
    W, H = 128, 96
    xyz, rgb, opac, sc, rot = _scene(rng, 3000, cuda, scale=(0.05, 0.3))
    cams = _cams(cuda, W, H, [(0, 0, 0), (2.0, 0.0, 0.0)])
    bg = torch.zeros(3, device=cuda)
    plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot, margin=1.0)
    plan.render(rgb, opac, sc, rot, bg)
    assert not plan.exceeded()
    big = sc.clone()
    big[5] *= 50.0
    plan.render(rgb, opac, big, rot, bg)
    assert plan.exceeded()
    other = _cams(cuda, W, H, [(0, 0, 0), (2.5, 0.0, 0.0)])
    plan.render(rgb, opac, sc, rot, bg, cameras=other)
    assert plan.exceeded()
    plan.render(rgb, opac, sc, rot, bg, item_view=torch.tensor([1, 1], dtype=torch.int32, device=cuda))
    with pytest.raises(Exception):
        plan.check()


def test_plan_matches_the_oracle(cuda, oracle_lib):
    rng = np.random.default_rng(5)
    W, H = 176, 64
    view, full, tfx, tfy = helpers.simple_camera(W, H)
    xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, 5000)
    want = oracle_lib.rasterize_forward(xyz, rgb, opac, sc, rot, view, full, tfx, tfy, H, W, np.float32([0, 0, 0]))
    cams = dgr.pack_cameras(_t(view, cuda).view(1, 4, 4), _t(full, cuda).view(1, 4, 4), [tfx], [tfy], H, W, cuda)
    plan = rp.RasterPlan(_t(xyz, cuda), cams, H, W, scales=_t(sc, cuda), rotations=_t(rot, cuda))
    got = plan.render(_t(rgb, cuda), _t(opac, cuda), _t(sc, cuda), _t(rot, cuda), torch.zeros(3, device=cuda),
                      want_radii=True)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in got.items()}
    # the shared comparison wants n_contrib / tiles_touched too: the plan path has neither, feed the oracle's own
    g['n_contrib'] = want['n_contrib'][None].astype(np.int32)
    g['tiles_touched'] = want['tiles_touched'][None].astype(np.int32)
    _compare(want, g, H, W)


def test_a_plan_outlives_its_parameters(cuda):
    """One plan, many parameter sets within its bound — and every one equals the per-call render."""
    rng = np.random.default_rng(7)
    W, H = 128, 96
    xyz, rgb, opac, sc, rot = _scene(rng, 8000, cuda)
    cams = _cams(cuda, W, H, [(0, 0, 0), (0.5, 0.2, 1.0)])
    bg = torch.zeros(3, device=cuda)
    plan = rp.RasterPlan(xyz, cams, H, W, extent_bound=1.0)
    for k in range(4):
        g = torch.Generator(device='cpu').manual_seed(k)
        sc_k = (torch.rand(8000, 3, generator=g) * 0.9 + 0.02).to(cuda)           # |s| <= 0.92 < bound
        q = torch.randn(8000, 4, generator=g)
        rot_k = (q / q.norm(dim=1, keepdim=True)).to(cuda)
        opac_k = torch.rand(8000, 1, generator=g).to(cuda)
        rgb_k = torch.rand(8000, 3, generator=g).to(cuda)
        want = dgr.rasterize_views(xyz, rgb_k, opac_k, sc_k, rot_k, None, None, None, None, H, W, bg,
                                   packed_cameras=cams, want_n_contrib=False)
        got = plan.render(rgb_k, opac_k, sc_k, rot_k, bg, want_radii=True)
        _same(want, got, ('color', 'depth', 'final_T', 'radii'))
    torch.cuda.synchronize()
    assert plan.check()


def test_extent_bound_guards(cuda):
    rng = np.random.default_rng(11)
    W, H = 128, 96
    xyz, rgb, opac, sc, rot = _scene(rng, 6000, cuda, scale=(0.05, 0.3))
    cams = _cams(cuda, W, H, [(0, 0, 0), (2.0, 0.0, 0.0)])
    bg = torch.tensor([0.3, 0.1, 0.0], device=cuda)
    plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot, margin=1.0)     # bound = the example's own extent
    big = sc.clone()
    big[::7] *= 9.0                                                                  # far beyond the bound
    want = dgr.rasterize_views(xyz, rgb, opac, big, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                               want_n_contrib=False)
    # host guard: the call is flagged
    plan.render(rgb, opac, big, rot, bg)
    assert plan.exceeded()
    # device guard: the armed per-call chain renders the call; results exact, flag raised
    got = plan.render(rgb, opac, big, rot, bg, guard='device')
    _same(want, got, ('color', 'depth', 'final_T', 'radii'))
    assert plan.exceeded()
    # within the bound the armed chain retires at once and the planned result stands
    want0 = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                                want_n_contrib=False)
    got0 = plan.render(rgb, opac, sc, rot, bg, guard='device')
    _same(want0, got0, ('color', 'depth', 'final_T', 'radii'))
    assert not plan.exceeded()
    # the flag is lowered by the call that fired (no memset per call): fired, fired, quiet, fired, quiet — each exact
    for scales, ref, fired in ((big, want, True), (big, want, True), (sc, want0, False), (big, want, True),
                               (sc, want0, False), (sc, want0, False)):
        got = plan.render(rgb, opac, scales, rot, bg, guard='device')
        _same(ref, got, ('color', 'depth', 'final_T', 'radii'))
        assert plan.exceeded() == fired
    # a host-guarded call that exceeds the bound leaves the flag alone (it is raised only for guarded calls)
    plan.render(rgb, opac, big, rot, bg)
    assert plan.exceeded()
    _same(want0, plan.render(rgb, opac, sc, rot, bg, guard='device'), ('color', 'depth', 'final_T', 'radii'))
    # the device guard WITHOUT the radii as an output (the neck module's call): the short front end — extent check, head of
    # every list — and the armed chain behind the blend; exact when the bound is exceeded, the planned images when not
    for scales, ref, fired in ((big, want, True), (sc, want0, False), (big, want, True), (sc, want0, False)):
        got = plan.render(rgb, opac, scales, rot, bg, guard='device', want_radii=False)
        assert 'radii' not in got
        _same(ref, got)
        assert plan.exceeded() == fired
    # a NaN scale counts as a violation
    bad = sc.clone()
    bad[5, 1] = float('nan')
    plan.render(rgb, opac, bad, rot, bg)
    assert plan.exceeded()


def test_item_views_and_sets(cuda):
    """The neck's shape: S samples with their own parameters, one (device-chosen) camera each."""
    rng = np.random.default_rng(13)
    W, H = 112, 64
    S, n = 3, 4000
    xyz = _t(helpers.random_gaussians(rng, n)[0], cuda)
    cams = _cams(cuda, W, H, [(0, 0, 0), (1.0, 0.0, 0.5), (-1.0, 0.2, 0.0), (0.0, -0.5, 2.0)])
    par = [_scene(rng, n, cuda)[1:] for _ in range(S)]
    rgb, opac, sc, rot = (torch.stack([p[i] for p in par]) for i in range(4))
    bg = torch.zeros(3, device=cuda)
    plan = rp.RasterPlan(xyz, cams, H, W, extent_bound=1.0)
    choice = torch.tensor([2, 0, 3], dtype=torch.int32, device=cuda)
    got = plan.render(rgb, opac, sc, rot, bg, item_view=choice, want_radii=True)
    for s in range(S):
        v = int(choice[s])
        want = dgr.rasterize_views(xyz, rgb[s], opac[s], sc[s], rot[s], None, None, None, None, H, W, bg,
                                   packed_cameras=cams[v:v + 1], want_n_contrib=False)
        for k in ('color', 'depth', 'final_T', 'radii'):
            assert torch.equal(want[k][0], got[k][s]), (s, k)
    # every set through every view, and the device guard with an item_view
    got_all = plan.render(rgb, opac, sc, rot, bg)
    assert got_all['color'].shape[0] == S * 4
    want = dgr.rasterize_views(xyz, rgb[1], opac[1], sc[1], rot[1], None, None, None, None, H, W, bg,
                               packed_cameras=cams, want_n_contrib=False)
    assert torch.equal(got_all['color'][4:8], want['color'])
    got_g = plan.render(rgb, opac, sc * 5.0, rot, bg, item_view=choice, guard='device')
    for s in range(S):
        v = int(choice[s])
        want = dgr.rasterize_views(xyz, rgb[s], opac[s], sc[s] * 5.0, rot[s], None, None, None, None, H, W, bg,
                                   packed_cameras=cams[v:v + 1], want_n_contrib=False)
        assert torch.equal(want['color'][0], got_g['color'][s])
    assert plan.exceeded()
    with pytest.raises(Exception):
        plan.render(rgb, opac, sc, rot, bg, item_view=torch.tensor([9, 0, 0], dtype=torch.int32, device=cuda))
        plan.check()


def test_more_items_than_the_blends_item_table(cuda):
    """The blend keeps the view of the first 64 rendered items in LDS (one read per workgroup instead of one per tile
    pair); items beyond that are looked up per tile pair: 9 sets x 8 views = 72 items, every one against the per-call
    pipeline, on the head path (no radii)."""
    rng = np.random.default_rng(23)
    W, H = 64, 48
    S, n = 9, 700
    xyz = _t(helpers.random_gaussians(rng, n)[0], cuda)
    cams = _cams(cuda, W, H, [(0.3 * k, 0.1 * (k % 3), 0.2 * k) for k in range(8)])
    par = [_scene(rng, n, cuda)[1:] for _ in range(S)]
    rgb, opac, sc, rot = (torch.stack([p[i] for p in par]) for i in range(4))
    bg = torch.tensor([0.2, 0.1, 0.0], device=cuda)
    plan = rp.RasterPlan(xyz, cams, H, W, extent_bound=1.0)
    got = plan.render(rgb, opac, sc, rot, bg, want_radii=False)
    torch.cuda.synchronize()
    assert plan.check() and got['color'].shape[0] == S * 8
    for s in range(S):
        want = dgr.rasterize_views(xyz, rgb[s], opac[s], sc[s], rot[s], None, None, None, None, H, W, bg,
                                   packed_cameras=cams, want_n_contrib=False)
        for k in ('color', 'depth', 'final_T'):
            assert torch.equal(want[k], got[k][8 * s:8 * s + 8]), (s, k)


def test_opacity_edge_values_and_degenerate_inputs(cuda):
    rng = np.random.default_rng(17)
    W, H = 96, 64
    xyz, rgb, opac, sc, rot = _scene(rng, 3000, cuda)
    opac[::5] = 1.0 / 255.0
    opac[1::5] = 0.0039
    opac[2::5] = 0.0
    opac[3::5] = 1.0
    opac[7] = -0.3
    opac[9] = float('nan')
    sc[11] = 0.0                              # a point: cov2D = 0.3 I
    rot[13] = 0.0                             # zero quaternion: R = I (unnormalised formula)
    cams = _cams(cuda, W, H, [(0, 0, 0)])
    bg = torch.tensor([1.0, 1.0, 1.0], device=cuda)
    want = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                               want_n_contrib=False)
    plan = rp.RasterPlan(xyz, cams, H, W, extent_bound=2.0)
    got = plan.render(rgb, opac, sc, rot, bg, want_radii=True)
    _same(want, got, ('color', 'depth', 'final_T', 'radii'))
    # without radii the update kernel skips the covariance work of Gaussians no pixel can blend (opacity < 1/255)
    _same(want, plan.render(rgb, opac, sc, rot, bg))
    assert plan.check()


def test_blend_grid_and_yield_hint_do_not_change_the_image(cuda):
    """The sorted blend is a persistent grid over a ticket queue: any number of workgroups renders the same image, and
    the ``yield_if`` hint (workgroups beyond ``blend_workgroups`` leave while the word is up) only changes who does
    the work.  The word is written in stream order without a launch (ocrf_stream_write_value32)."""
    from ocrfdet_amd import _lib
    rng = np.random.default_rng(29)
    W, H = 176, 80
    xyz, rgb, opac, sc, rot = _scene(rng, 20000, cuda)
    cams = _cams(cuda, W, H, [(0, 0, 0), (1.5, -0.5, 2.0), (-3.0, 0.4, -1.0)])
    bg = torch.tensor([0.0, 0.3, 0.6], device=cuda)
    plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot)
    want = {k: v.clone() for k, v in plan.render(rgb, opac, sc, rot, bg).items()}
    word = torch.zeros(1, dtype=torch.int32, device=cuda)
    for grid in (1, 7, 64, 4096):
        _same(want, plan.render(rgb, opac, sc, rot, bg, blend_workgroups=grid))
        for busy in (1, 0, 5):
            _lib.check(_lib.lib().ocrf_stream_write_value32(_lib.ptr(word), busy, _lib.stream_ptr(cuda)), 'write_value32')
            got = plan.render(rgb, opac, sc, rot, bg, blend_workgroups=grid, yield_if=word)
            assert int(word.item()) == busy
            _same(want, got)
    assert plan.check()


def test_mostly_transparent_scene(cuda):
    """Free space: 90 % of the Gaussians under the 1/255 opacity threshold (their rects are emptied by the update
    kernel, so no tile pair scans into them) — same images as the per-call pipeline, with and without radii."""
    rng = np.random.default_rng(23)
    W, H = 176, 80
    xyz, rgb, opac, sc, rot = _scene(rng, 20000, cuda)
    faint = torch.from_numpy(rng.random(20000) < 0.9).to(cuda)
    opac[faint] = opac[faint] * 0.0039
    cams = _cams(cuda, W, H, [(0, 0, 0), (1.5, -0.5, 2.0)])
    bg = torch.tensor([0.2, 0.1, 0.0], device=cuda)
    want = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                               want_n_contrib=False)
    plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot)
    _same(want, plan.render(rgb, opac, sc, rot, bg))
    _same(want, plan.render(rgb, opac, sc, rot, bg, want_radii=True), ('color', 'depth', 'final_T', 'radii'))
    assert plan.check()


def _heads(plan):
    """head[v] control words of the plan's per-call scratch (raster_plan.hip kCtlHead)."""
    ctl = plan._dyn[plan._dyn.numel() - 1024:].view(torch.int32).cpu().numpy()
    return [int(x) for x in ctl[32:32 + plan.V]]


@pytest.mark.parametrize('bins', [(4, 2), (1, 1), (2, 1), (3, 5), None])
def test_candidate_lists_change_nothing(cuda, bins):
    """The plan-time candidate lists (what a tile may see: rasterizer_impl.cu:70-138, here per bin and for every extent up
    to the plan's bound) only shorten what a tile pair scans: the images are the per-call pipeline's bit for bit, whatever
    the bin shape, for a scene that saturates and for one that is mostly transparent, with and without radii."""
    rng = np.random.default_rng(31)
    W, H = 176, 80
    cams = _cams(cuda, W, H, [(0, 0, 0), (1.5, -0.5, 2.0), (-3.0, 0.4, -1.0)])
    bg = torch.tensor([0.1, 0.2, 0.3], device=cuda)
    for faint_frac in (0.0, 0.9):
        xyz, rgb, opac, sc, rot = _scene(rng, 30000, cuda)
        if faint_frac:
            faint = torch.from_numpy(rng.random(30000) < faint_frac).to(cuda)
            opac[faint] = opac[faint] * 0.012              # around the 1/255 threshold: tight alpha ellipses, empty rects
        want = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                                   want_n_contrib=False)
        plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot, bins=bins)
        try:
            for head in (0, -1, 256, 0):
                _head_mode(head)
                _same(want, plan.render(rgb, opac, sc, rot, bg))
        finally:
            _head_mode(0)
        _same(want, plan.render(rgb, opac, sc, rot, bg, want_radii=True), ('color', 'depth', 'final_T', 'radii'))
        assert plan.check()
        if bins is not None:
            assert plan.candidates > 0 and plan.cand_capacity >= plan.candidates


def test_candidate_lists_that_do_not_fit_are_ignored(cuda):
    """A candidate capacity too small for the lists (as after a ``rebuild`` for a pose that lists more): the build leaves
    the buffer without its magic word and the renders walk whole lists — slower, the same image."""
    rng = np.random.default_rng(41)
    W, H = 176, 80
    xyz, rgb, opac, sc, rot = _scene(rng, 20000, cuda)
    cams = _cams(cuda, W, H, [(0, 0, 0), (1.5, -0.5, 2.0)])
    bg = torch.tensor([0.1, 0.0, 0.3], device=cuda)
    want = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                               want_n_contrib=False)
    fits = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot)
    small = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot, cand_capacity=fits.candidates // 3)
    assert int(fits._bins_buf[:4].view(torch.int32).item()) != 0          # "OCRB"
    assert int(small._bins_buf[:4].view(torch.int32).item()) == 0         # no magic: ignored
    _same(want, fits.render(rgb, opac, sc, rot, bg))
    _same(want, small.render(rgb, opac, sc, rot, bg))
    assert small.check() and fits.check()


@pytest.mark.parametrize('bins', [(4, 2), (8, 2), (1, 1), (6, 3)])
def test_lists_that_never_saturate_are_prepared_once_and_whole(cuda, bins):
    """An object-centric opacity field: most Gaussians faint, so most pixels never reach T < 1e-4 and a tile pair walks its
    whole candidate list — far beyond round 5's 16 384-entry head.  The first call hands such tile pairs to the second pass
    (the rest of the lists is prepared once in between), the head then covers the whole list; every call gives the per-call
    pipeline's image bit for bit, and the head shrinks again when the scene saturates early.  From the second such call on the
    views are "deep": the host-visible hint is up, the call compacts the bins' candidates for this call's rects and the
    build of the blend that reads them renders (bins of up to 32 tiles; (6, 3) = 36 tiles: no compaction, same image)."""
    rng = np.random.default_rng(37)
    W, H = 176, 80
    n = 90000
    xyz, rgb, opac, sc, rot = _scene(rng, n, cuda)
    cams = _cams(cuda, W, H, [(0, 0, 0), (1.5, -0.5, 2.0)])
    bg = torch.tensor([0.3, 0.2, 0.1], device=cuda)
    faint = opac * 0.01                                    # every alpha a few times 1/255 at most
    plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot, bins=bins)
    assert min(plan.kept) > 2 * 16384
    want_faint = dgr.rasterize_views(xyz, rgb, faint, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                                     want_n_contrib=False)
    want_dense = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                                     want_n_contrib=False)
    assert float(want_faint['final_T'].max()) > 1e-3       # pixels that never saturate
    _same(want_dense, plan.render(rgb, opac, sc, rot, bg))
    dense_heads = _heads(plan)
    _same(want_faint, plan.render(rgb, faint, sc, rot, bg))        # lagging head: second pass
    grown = _heads(plan)
    assert all(h >= k for h, k in zip(grown, plan.kept)), (grown, plan.kept)
    torch.cuda.synchronize()
    assert int(plan._hint.item()) == 1                             # the close-out told the host: deep views
    _same(want_faint, plan.render(rgb, faint, sc, rot, bg))        # whole lists prepared by the head kernel, candidates compacted
    _same(want_faint, plan.render(0.5 * rgb, faint, sc, rot, bg), ('depth', 'final_T'))
    _same(want_dense, plan.render(rgb, opac, sc, rot, bg))         # (a deep-mode call on a scene that is not deep any more)
    torch.cuda.synchronize()
    assert int(plan._hint.item()) == 0
    _same(want_dense, plan.render(rgb, opac, sc, rot, bg))
    assert max(_heads(plan)) <= 4 * max(dense_heads) + 1024
    assert plan.check()


def test_full_size_ocrf_grid_both_conventions(cuda):
    """cfg2's own scene: the 13 x 200 x 200 voxel-grid Gaussians through the six cameras at 256 x 704."""
    from ocrfdet_amd import hotpath
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    hp = hotpath.HotPath(cfg, cuda)
    for conv in ('corrected', 'reference'):
        r = synthetic.rig(cfg.n_cams, cfg.input_size, hp.batch)
        hp._prepare_render(r, convention=conv)
        g, rc = hp.gauss, hp.render_cams
        H, W = cfg.input_size
        xyz = hp.voxel_xyz[0].reshape(-1, 3)
        want = dgr.rasterize_views(xyz, g['rgb'], g['opacity'], g['scales'], g['rotations'], None, None, None, None,
                                   H, W, hp.bg, packed_cameras=rc['packed'], want_n_contrib=False)
        plan = rp.RasterPlan(xyz, rc['packed'], H, W, scales=g['scales'], rotations=g['rotations'])
        got = plan.render(g['rgb'], g['opacity'], g['scales'], g['rotations'], hp.bg, want_radii=True)
        torch.cuda.synchronize()
        assert plan.check()
        _same(want, got, ('color', 'depth', 'final_T', 'radii'))
        assert sum(plan.kept) < 0.5 * xyz.shape[0] * 6


def test_rebuild_for_another_pose_without_a_host_read(cuda):
    """A plan per SAMPLE (the reference builds the render cameras from the dataloader's c2w per sample,
    view_transformer_ocrf.py:1140-1152): ``rebuild(cameras)`` between two renders gives the per-call image of the new
    pose bit for bit — enqueued while the device is still busy and captured into a hipGraph, i.e. without any host read
    or allocation inside — and a rebuild back gives the first pose's image again."""
    rng = np.random.default_rng(21)
    W, H = 176, 96
    xyz, rgb, opac, sc, rot = _scene(rng, 20000, cuda)
    bg = torch.tensor([0.0, 0.1, 0.2], device=cuda)
    poses = [[(0, 0, 0), (1.5, -0.5, 2.0)], [(0.4, 0.1, -0.3), (1.1, -0.2, 2.6)], [(-0.8, 0.0, 0.5), (2.0, 0.3, 1.0)]]
    cams = [_cams(cuda, W, H, p) for p in poses]
    want = [dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=c,
                                want_n_contrib=False) for c in cams]
    plan = rp.RasterPlan(xyz, cams[0], H, W, scales=sc, rotations=rot, headroom=1.5)
    _same(want[0], plan.render(rgb, opac, sc, rot, bg, want_radii=True), ('color', 'depth', 'final_T', 'radii'))
    for k in (1, 2, 0, 2):
        plan.rebuild(cams[k])
        got = plan.render(rgb, opac, sc, rot, bg, want_radii=True, cameras=cams[k])
        _same(want[k], got, ('color', 'depth', 'final_T', 'radii'))
    assert plan.check()
    # captured: rebuild + render as one hipGraph on a side stream, replayed for two poses through a static camera block
    static_cams = cams[1].clone()
    out = plan.render(rgb, opac, sc, rot, bg)
    side = torch.cuda.Stream(cuda)
    side.wait_stream(torch.cuda.current_stream(cuda))
    with torch.cuda.stream(side):
        plan.rebuild(static_cams)
        plan.render(rgb, opac, sc, rot, bg, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            plan.rebuild(static_cams)
            plan.render(rgb, opac, sc, rot, bg, out=out, cameras=static_cams)
    torch.cuda.current_stream(cuda).wait_stream(side)
    for k in (2, 1, 0):
        static_cams.copy_(cams[k])
        g.replay()
        _same(want[k], out)
    torch.cuda.synchronize()
    assert plan.check()


def test_a_plan_is_keyed_by_its_cameras_on_the_device(cuda):
    """``render(cameras=...)`` with cameras that are not the plan's: status bit 16; with ``guard='device'`` the per-call
    pipeline renders the call with THOSE cameras (exact), with ``guard='host'`` ``check()`` raises."""
    rng = np.random.default_rng(22)
    W, H = 128, 96
    xyz, rgb, opac, sc, rot = _scene(rng, 8000, cuda)
    bg = torch.zeros(3, device=cuda)
    a, b = _cams(cuda, W, H, [(0, 0, 0), (2.0, 0.0, 0.0)]), _cams(cuda, W, H, [(0.3, 0.1, 0.0), (2.0, 0.0, 0.4)])
    want_b = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=b,
                                 want_n_contrib=False)
    want_a = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=a,
                                 want_n_contrib=False)
    plan = rp.RasterPlan(xyz, a, H, W, scales=sc, rotations=rot)
    _same(want_a, plan.render(rgb, opac, sc, rot, bg, guard='device', cameras=a), ('color', 'depth', 'final_T', 'radii'))
    assert not plan.exceeded()
    got = plan.render(rgb, opac, sc, rot, bg, guard='device', cameras=b)       # a stale plan: the call is still exact
    _same(want_b, got, ('color', 'depth', 'final_T', 'radii'))
    assert plan.check()                                                         # ... and that is not an error
    _same(want_a, plan.render(rgb, opac, sc, rot, bg, guard='device', cameras=a), ('color', 'depth', 'final_T', 'radii'))
    plan.render(rgb, opac, sc, rot, bg, cameras=b)                              # host guard: flagged, check() raises
    with pytest.raises(Exception, match='cameras'):
        plan.check()


def test_a_plan_beyond_its_capacity_is_refused_on_the_device(cuda):
    rng = np.random.default_rng(23)
    W, H = 128, 96
    xyz, rgb, opac, sc, rot = _scene(rng, 8000, cuda)
    bg = torch.zeros(3, device=cuda)
    cams = _cams(cuda, W, H, [(0, 0, 0), (2.0, 0.0, 0.0)])
    full = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot)
    need = sum(full.kept)
    small = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot, capacity=need // 2)
    want = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams,
                               want_n_contrib=False)
    out = small.render(rgb, opac, sc, rot, bg)
    with pytest.raises(Exception, match='capacity'):
        small.check()
    # nothing was rendered from the unusable plan: the images are zeros, not whatever the allocation held
    assert all(float(out[k].abs().max()) == 0.0 for k in ('color', 'depth', 'final_T'))
    exact = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot, capacity=need)      # not one record to spare
    _same(want, exact.render(rgb, opac, sc, rot, bg))
    assert exact.check()
