"""GPU parity: HIP bev_pool_v2 (through the C ABI) vs the C oracle on the same inputs.

Tolerance: north_star states BEV features within 1e-4 fp32; a voxel whose points lie inside one 32-point
sub-chunk of its tile's point list is summed in list order exactly like the reference (bit-exact), the
others are re-associated (deterministically)."""
import ctypes

import numpy as np
import pytest
import torch

from ocrfdet_amd import _lib, bevpool, synthetic
from tests import helpers

pytestmark = pytest.mark.gpu
ATOL = RTOL = 1e-4
SUB = 32     # points per lane group and round in csrc/bev_pool.hip (kSub)
TV = 64      # voxels per tile (kTV)


def _in_one_subchunk(rb, st, ln, plane):
    """Intervals whose points lie inside ONE 32-point sub-chunk of the kernel's walk: a tile (64 consecutive
    voxels of one (b, z) plane of ``plane`` voxels) walks the points of its voxels in voxel order, cut into
    sub-chunks of 32 from the tile's first point."""
    vox = rb[st].astype(np.int64)
    tile = (vox // plane) * ((plane + TV - 1) // TV) + (vox % plane) // TV
    order = np.lexsort((vox, tile))
    t_s, l_s = tile[order], ln[order].astype(np.int64)
    cum = np.cumsum(l_s) - l_s                            # exclusive prefix over all intervals in (tile, voxel) order
    first = np.ones(len(t_s), bool)
    first[1:] = t_s[1:] != t_s[:-1]
    base = np.maximum.accumulate(np.where(first, cum, 0))
    off = cum - base                                      # first point of the interval in its tile's list
    ok = np.zeros(len(st), bool)
    ok[order] = (off // SUB) == ((off + l_s - 1) // SUB)
    return ok


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _run(cuda, depth, feat, rd, rf, rb, shape, st, ln):
    out = bevpool.bev_pool_v2(_dev(depth, cuda), _dev(feat, cuda), _dev(rd, cuda), _dev(rf, cuda),
                              _dev(rb, cuda), shape, _dev(st, cuda), _dev(ln, cuda))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_kat_forward_backward(cuda):
    """The reference's own known-answer test (bev_pool.py:145-176), verbatim numbers."""
    depth = torch.tensor([0.3, 0.4, 0.2, 0.1, 0.7, 0.6, 0.8, 0.9], device=cuda).view(1, 1, 2, 2, 2).requires_grad_()
    feat = torch.ones(1, 1, 2, 2, 2, device=cuda).requires_grad_()
    rd = torch.tensor([0, 4, 1, 6], dtype=torch.int32, device=cuda)
    rf = torch.tensor([0, 0, 1, 2], dtype=torch.int32, device=cuda)
    rb = torch.tensor([0, 0, 1, 1], dtype=torch.int32, device=cuda)
    st, ln = bevpool.runs_of(rb)
    bev = bevpool.bev_pool_v2(depth, feat, rd, rf, rb, (1, 1, 2, 2, 2), st, ln)
    loss = bev.sum()
    loss.backward()
    assert loss.item() == pytest.approx(4.4, abs=1e-6)
    assert torch.allclose(depth.grad.view(-1).cpu(), torch.tensor([2., 2., 0., 0., 2., 0., 2., 0.]))
    assert torch.allclose(feat.grad.view(-1).cpu(), torch.tensor([1., 1., .4, .4, .8, .8, 0., 0.]))


@pytest.mark.parametrize('cfg_name', ['cfg0_1cam_128x352_bev64x64x4', 'ref_6cam_256x704_bev128x128x1',
                                      'cfg1_6cam_256x704_bev128x128x8',
                                      'cfg2_6cam_2frame_bev200x200_render_hoa'])
@pytest.mark.parametrize('branch', ['lss', 'ht'])
def test_parity_reference_shapes(cuda, oracle_lib, cfg_name, branch):
    cfg = synthetic.CONFIGS[cfg_name]
    depth, feat = helpers.pool_inputs(cfg)
    X, Y, Z = cfg.bev_xyz
    if branch == 'lss':
        rb, rd, rf, st, ln = helpers.lss_ranks(cfg)
        shape = (cfg.batch, Z, Y, X, cfg.channels)
    else:
        rb, rd, rf, st, ln = helpers.ht_ranks(cfg)
        shape = (cfg.batch, 1, Y, X, cfg.channels)
    want = oracle_lib.bev_pool_v2(depth, feat, rd, rf, rb, shape, st, ln)
    got = _run(cuda, depth, feat, rd, rf, rb, shape, st, ln)
    assert got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    # voxels that do not straddle a 32-point sub-chunk border of their tile's walk are bit-exact
    short = _in_one_subchunk(rb, st, ln, Y * X)
    assert short.mean() > 0.3
    vox = rb[st[short]]
    g = got.transpose(0, 2, 3, 4, 1).reshape(-1, cfg.channels)[vox]
    w = want.transpose(0, 2, 3, 4, 1).reshape(-1, cfg.channels)[vox]
    np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize('c', [32, 64, 80, 128, 256, 3, 20, 512])
def test_parity_channel_counts_and_skew(cuda, oracle_lib, c):
    rng = np.random.default_rng(c)
    n_vox = 2000
    depth, feat, rd, rf, rb, st, ln = helpers.random_pool_problem(rng, 50000, n_vox, c)
    shape = (1, 1, 1, n_vox, c)
    d5, f5 = depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c)
    want = oracle_lib.bev_pool_v2(d5, f5, rd, rf, rb, shape, st, ln)
    got = _run(cuda, d5, f5, rd, rf, rb, shape, st, ln)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL * max(1.0, np.abs(want).max()))


def test_edge_cases(cuda, oracle_lib):
    rng = np.random.default_rng(7)
    c = 80
    # one interval covering every point (longest possible chain of partial rows)
    depth, feat, rd, rf, rb, st, ln = helpers.random_pool_problem(rng, 10000, 1, c)
    shape = (1, 1, 1, 4, c)
    want = oracle_lib.bev_pool_v2(depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c), rd, rf, rb, shape, st, ln)
    got = _run(cuda, depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c), rd, rf, rb, shape, st, ln)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-3)
    # every point its own interval; point counts around sub-chunk (32), round (384) and slice (768) borders
    for n in (1, 31, 32, 33, 63, 64, 65, 383, 384, 385, 767, 768, 769, 1537):
        rb = np.arange(n, dtype=np.int32)
        rd = rng.integers(0, 100, n).astype(np.int32)
        rf = rng.integers(0, 50, n).astype(np.int32)
        depth = rng.random(100, dtype=np.float32)
        feat = rng.standard_normal((50, c)).astype(np.float32)
        st, ln = np.arange(n, dtype=np.int32), np.ones(n, np.int32)
        shape = (1, 1, 1, n, c)
        want = oracle_lib.bev_pool_v2(depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c), rd, rf, rb, shape, st, ln)
        got = _run(cuda, depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c), rd, rf, rb, shape, st, ln)
        np.testing.assert_array_equal(got, want)
    # gaps: points outside every interval are ignored (intervals need not cover the list)
    depth, feat, rd, rf, rb, st, ln = helpers.random_pool_problem(rng, 5000, 300, c)
    keep = np.arange(len(st)) % 3 != 1
    want = oracle_lib.bev_pool_v2(depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c), rd, rf, rb, (1, 1, 1, 300, c), st[keep], ln[keep])
    got = _run(cuda, depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c), rd, rf, rb, (1, 1, 1, 300, c), st[keep], ln[keep])
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


def test_any_interval_layout_is_accepted(cuda, oracle_lib):
    """The tiled kernel reads intervals through a dense voxel table, so — like the reference's one-thread-per-
    interval kernel (bev_pool_cuda.cu:21-48) — it takes ANY layout: intervals in arbitrary order, empty
    intervals, intervals that share points, points in no interval.  (Two intervals naming the same voxel race
    in the reference; not exercised.)"""
    rng = np.random.default_rng(21)
    c = 80
    n_vox = 700
    depth, feat, rd, rf, rb, st, ln = helpers.random_pool_problem(rng, 30000, n_vox, c)
    perm = rng.permutation(len(st))
    st_p, ln_p = st[perm].copy(), ln[perm].copy()
    ln_p[::7] = 0                                    # empty intervals: their voxels pool to 0
    grow = np.arange(len(st_p)) % 5 == 1             # overlapping: run on into the next interval's points
    ln_p[grow] = np.minimum(ln_p[grow] + 3, len(rb) - st_p[grow])
    keep = np.arange(len(st_p)) % 11 != 3            # gaps: dropped intervals leave their points unused
    st_p, ln_p = np.ascontiguousarray(st_p[keep]), np.ascontiguousarray(ln_p[keep])
    shape = (1, 1, 7, 100, c)
    d5, f5 = depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c)
    want = oracle_lib.bev_pool_v2(d5, f5, rd, rf, rb, shape, st_p, ln_p)
    got = _run(cuda, d5, f5, rd, rf, rb, shape, st_p, ln_p)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL * max(1.0, np.abs(want).max()))
    # the reference-contract entry point (channel-last rows, QuickCumsumCuda) through the same kernel
    out = bevpool.QuickCumsumCuda.apply(_dev(d5, cuda), _dev(f5, cuda), _dev(rd, cuda), _dev(rf, cuda), _dev(rb, cuda),
                                        shape, _dev(st_p, cuda), _dev(ln_p, cuda))
    np.testing.assert_array_equal(out.permute(0, 4, 1, 2, 3).cpu().numpy(), got)


def test_heavy_tiles_are_cut_and_recombined_deterministically(cuda, oracle_lib):
    """A tile with more points than one slice (2 rounds x 384 points at C = 80) runs as several workgroups whose
    partial tiles are added in slice order by the last to arrive: exact same bits run after run, and the
    same bits whatever the slice length (csrc tuning knob) only up to re-association."""
    rng = np.random.default_rng(3)
    c = 80
    n_vox = 256                                       # 4 tiles; ~all points in a handful of voxels of tile 1
    w = np.full(n_vox, 1e-3)
    w[70:75] = [30, 5, 60, 1, 20]
    vox = np.sort(rng.choice(n_vox, size=40000, p=w / w.sum())).astype(np.int32)
    rd = rng.integers(0, 5000, vox.size).astype(np.int32)
    rf = rng.integers(0, 700, vox.size).astype(np.int32)
    depth = rng.random(5000, dtype=np.float32)
    feat = rng.standard_normal((700, c)).astype(np.float32)
    import oracle
    st, ln = oracle.intervals_from_sorted(vox)
    shape = (1, 1, 4, 64, c)
    d5, f5 = depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c)
    want = oracle_lib.bev_pool_v2(d5, f5, rd, rf, vox, shape, st, ln)
    a = _run(cuda, d5, f5, rd, rf, vox, shape, st, ln)
    np.testing.assert_allclose(a, want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
    for _ in range(4):
        np.testing.assert_array_equal(a, _run(cuda, d5, f5, rd, rf, vox, shape, st, ln))


def test_duplicate_intervals_cannot_overrun_the_slabs(cuda, oracle_lib):
    """Hundreds of copies of one long interval: the voxel table keeps one of them (they are identical, so any winner
    equals the reference), but every copy counts into its tile, which then "needs" far more slices than the slab buffer
    (sized for n_points) holds.  Such a tile is pooled as ONE long unit instead of being cut — no slab is written out
    of bounds — and the result still equals the reference."""
    rng = np.random.default_rng(33)
    c = 80
    n_pts = 3000
    rd = rng.integers(0, 400, n_pts).astype(np.int32)
    rf = rng.integers(0, 90, n_pts).astype(np.int32)
    rb = np.full(n_pts, 37, np.int32)
    rb[2000:] = 150                                                   # two voxels, in tiles 0 and 2
    depth = rng.random(400, dtype=np.float32)
    feat = rng.standard_normal((90, c)).astype(np.float32)
    st = np.concatenate((np.zeros(300, np.int32), np.full(5, 2000, np.int32)))
    ln = np.concatenate((np.full(300, 2000, np.int32), np.full(5, 1000, np.int32)))
    shape = (1, 1, 4, 64, c)
    d5, f5 = depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c)
    want = oracle_lib.bev_pool_v2(d5, f5, rd, rf, rb, shape, st[[0, 300]], ln[[0, 300]])
    got = _run(cuda, d5, f5, rd, rf, rb, shape, st, ln)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
    np.testing.assert_array_equal(got, _run(cuda, d5, f5, rd, rf, rb, shape, st, ln))


def test_empty_inputs(cuda):
    e = torch.zeros(0, dtype=torch.int32, device=cuda)
    out = bevpool.bev_pool_v2(torch.zeros(1, 1, 2, 2, 2, device=cuda), torch.zeros(1, 1, 2, 2, 80, device=cuda),
                              e, e, e, (1, 1, 4, 4, 80), e, e)
    assert out.shape == (1, 80, 1, 4, 4) and float(out.abs().sum()) == 0.0


def test_run_to_run_bitwise_reproducible(cuda):
    cfg = synthetic.CONFIGS['ref_6cam_256x704_bev128x128x1']
    depth, feat = helpers.pool_inputs(cfg)
    rb, rd, rf, st, ln = helpers.lss_ranks(cfg)
    X, Y, Z = cfg.bev_xyz
    a = _run(cuda, depth, feat, rd, rf, rb, (1, Z, Y, X, cfg.channels), st, ln)
    for _ in range(3):
        b = _run(cuda, depth, feat, rd, rf, rb, (1, Z, Y, X, cfg.channels), st, ln)
        np.testing.assert_array_equal(a, b)


def test_linearity_and_total_mass_full_size(cuda):
    """Size-independent properties at the headline size (6 cams, 200x200 BEV)."""
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    depth, feat = helpers.pool_inputs(cfg)
    rb, rd, rf, st, ln = helpers.lss_ranks(cfg)
    X, Y, Z = cfg.bev_xyz
    shape = (1, Z, Y, X, cfg.channels)
    a = _run(cuda, depth, feat, rd, rf, rb, shape, st, ln)
    b = _run(cuda, depth, 2.0 * feat, rd, rf, rb, shape, st, ln)
    np.testing.assert_array_equal(2.0 * a, b)                  # exact: scaling by 2 commutes with fp32 rounding
    # sum over voxels of out[:, c] == sum over kept points of depth*feat[:, c]
    tot = (depth.reshape(-1)[rd].astype(np.float64)[:, None] *
           feat.reshape(-1, cfg.channels)[rf].astype(np.float64)).sum(0)
    np.testing.assert_allclose(a.astype(np.float64).sum((0, 2, 3, 4)), tot, rtol=1e-5, atol=1e-3)
    # untouched voxels stay zero
    touched = np.zeros(Z * Y * X, bool)
    touched[rb] = True
    assert not a.transpose(0, 2, 3, 4, 1).reshape(-1, cfg.channels)[~touched].any()


def test_exact_signature_entry_and_interval_checker(cuda, oracle_lib):
    """`bev_pool_v2(c, n_intervals, ...)` with the reference's exact C signature
    (src/bev_pool.cpp:7-9) accepts unsorted / overlapping intervals."""
    rng = np.random.default_rng(11)
    c = 80
    depth, feat, rd, rf, rb, st, ln = helpers.random_pool_problem(rng, 6000, 200, c)
    perm = rng.permutation(len(st))
    st_p, ln_p = np.ascontiguousarray(st[perm]), np.ascontiguousarray(ln[perm])
    want = oracle_lib.bev_pool_v2_raw(depth.reshape(1, 1, -1, 1, 1), feat.reshape(1, 1, 1, -1, c), rd, rf, rb, (1, 1, 1, 200, c), st_p, ln_p)
    L = _lib.lib()
    t = [_dev(x, cuda) for x in (depth, feat, rd, rf, rb, st_p, ln_p)]
    out = torch.zeros(1, 1, 1, 200, c, device=cuda)
    torch.cuda.synchronize()
    L.bev_pool_v2(c, len(st_p), *[_lib.ptr(x) for x in t], _lib.ptr(out))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want)     # same sequential order -> bit-exact
    flag = torch.zeros(1, dtype=torch.int32, device=cuda)
    _lib.check(L.ocrf_bev_pool_v2_check_intervals(len(st_p), 6000, _lib.ptr(t[5]), _lib.ptr(t[6]), _lib.ptr(flag),
                                                   _lib.stream_ptr(cuda)), 'check')
    assert int(flag.item()) & 1
    st_d, ln_d = _dev(st, cuda), _dev(ln, cuda)       # keep alive: raw pointers cross the C ABI
    _lib.check(L.ocrf_bev_pool_v2_check_intervals(len(st), 6000, _lib.ptr(st_d), _lib.ptr(ln_d),
                                                   _lib.ptr(flag), _lib.stream_ptr(cuda)), 'check')
    assert int(flag.item()) == 0
    # bad arguments are reported, not launched
    assert L.ocrf_bev_pool_v2(c, 5, 100, 200, None, None, None, None, None, None, None, None, None,
                              ctypes.c_size_t(0), None) != 0


def test_backward_parity(cuda, oracle_lib):
    cfg = synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4']
    depth, feat = helpers.pool_inputs(cfg)
    rb, rd, rf, st, ln = helpers.lss_ranks(cfg)     # LSS: every depth cell appears once -> no store race
    X, Y, Z = cfg.bev_xyz
    shape = (1, Z, Y, X, cfg.channels)
    rng = np.random.default_rng(5)
    og = rng.standard_normal(shape).astype(np.float32)
    want_d, want_f = oracle_lib.bev_pool_v2_backward(og, depth, feat, rd, rf, rb)
    d = _dev(depth, cuda).requires_grad_()
    f = _dev(feat, cuda).requires_grad_()
    out = bevpool.QuickCumsumCuda.apply(d, f, _dev(rd, cuda), _dev(rf, cuda), _dev(rb, cuda), shape,
                                        _dev(st, cuda), _dev(ln, cuda))
    out.backward(_dev(og, cuda))
    np.testing.assert_allclose(d.grad.cpu().numpy(), want_d, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(f.grad.cpu().numpy(), want_f, rtol=1e-4, atol=1e-4)


def test_backward_parity_full_size_through_the_fused_op(cuda, oracle_lib):
    """cfg2 (6 cams x 2 frames, BEV 200 x 200, 852 k points): the gradients of the drop-in ``bev_pool_v2`` (fused
    forward writing (B,C,Z,Y,X) directly, HIP backward) against the C oracle's backward on the same ranks."""
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    depth, feat = helpers.pool_inputs(cfg)
    rb, rd, rf, st, ln = helpers.lss_ranks(cfg)
    X, Y, Z = cfg.bev_xyz
    B = depth.shape[0]
    shape = (B, Z, Y, X, cfg.channels)
    rng = np.random.default_rng(11)
    og = rng.standard_normal(shape).astype(np.float32)                      # gradient in the op's (B,Z,Y,X,C) frame
    want_d, want_f = oracle_lib.bev_pool_v2_backward(og, depth, feat, rd, rf, rb)
    d = _dev(depth, cuda).requires_grad_()
    f = _dev(feat, cuda).requires_grad_()
    out = bevpool.bev_pool_v2(d, f, _dev(rd, cuda), _dev(rf, cuda), _dev(rb, cuda), shape, _dev(st, cuda), _dev(ln, cuda))
    assert tuple(out.shape) == (B, cfg.channels, Z, Y, X)
    out.backward(_dev(og, cuda).permute(0, 4, 1, 2, 3).contiguous())       # the same gradient in (B,C,Z,Y,X)
    scale_d, scale_f = float(np.abs(want_d).max()), float(np.abs(want_f).max())
    assert float(np.abs(d.grad.cpu().numpy() - want_d).max()) <= 1e-4 * max(scale_d, 1.0)
    assert float(np.abs(f.grad.cpu().numpy() - want_f).max()) <= 1e-4 * max(scale_f, 1.0)


@pytest.mark.parametrize('cfg_name', ['cfg0_1cam_128x352_bev64x64x4', 'ref_6cam_256x704_bev128x128x1'])
def test_backward_ht_ranks_repeated_depth_cells(cuda, oracle_lib, cfg_name):
    """HT ranks: several pillar samples round to the same (camera, d, h, w) depth cell, so ``ranks_depth``
    repeats.  The reference stores ``depth_grad[ranks_depth[p]] = ...`` without accumulation
    (bev_pool_cuda.cu:103-104): a repeated cell ends with ONE contributor's value.  A depth cell fixes its
    feature pixel, so all of a cell's contributors lie in one ranks_feat run, i.e. in ONE reference thread,
    and the last of them in that run's order wins; the reference's order inside a run comes from an unstable
    argsort (bev_pool.py:47), so any contributor is a legal winner.  Here the sort is stable and a run is walked
    in order by one lane group: the winner is the LAST contributor in forward list order — deterministic, and
    the same as the C oracle's (which also sorts stably).  feat_grad is a sum and has no such freedom."""
    cfg = synthetic.CONFIGS[cfg_name]
    depth, feat = helpers.pool_inputs(cfg)
    rb, rd, rf, st, ln = helpers.ht_ranks(cfg)
    uniq, cnt = np.unique(rd, return_counts=True)
    assert (cnt > 1).sum() > 100, 'the HT ranks of this configuration should repeat depth cells'
    X, Y, _ = cfg.bev_xyz
    shape = (cfg.batch, 1, Y, X, cfg.channels)
    rng = np.random.default_rng(9)
    og = rng.standard_normal(shape).astype(np.float32)
    want_d, want_f = oracle_lib.bev_pool_v2_backward(og, depth, feat, rd, rf, rb)
    grads = []
    for _ in range(3):
        d = _dev(depth, cuda).requires_grad_()
        f = _dev(feat, cuda).requires_grad_()
        out = bevpool.QuickCumsumCuda.apply(d, f, _dev(rd, cuda), _dev(rf, cuda), _dev(rb, cuda), shape,
                                            _dev(st, cuda), _dev(ln, cuda))
        out.backward(_dev(og, cuda))
        grads.append((d.grad.cpu().numpy(), f.grad.cpu().numpy()))
    gd, gf = grads[0]
    np.testing.assert_allclose(gf, want_f, rtol=1e-4, atol=1e-4)
    # (1) every cell holds the value of one of its contributors (the reference's contract) ...
    per_point = np.einsum('pc,pc->p', og.reshape(-1, cfg.channels)[rb].astype(np.float64),
                          feat.reshape(-1, cfg.channels)[rf].astype(np.float64))
    flat = gd.reshape(-1)
    err = np.abs(flat[rd] - per_point)                      # distance of the cell's value to each contributor
    best = np.full(depth.size, np.inf)
    np.minimum.at(best, rd, err)
    assert best[uniq].max() <= 1e-4
    untouched = np.ones(depth.size, bool)
    untouched[rd] = False
    assert not flat[untouched].any()
    # (2) ... namely the last one in list order, like the oracle, and the same on every run
    np.testing.assert_allclose(gd, want_d, rtol=1e-4, atol=1e-4)
    last = np.zeros(depth.size, np.int64)
    last[rd] = np.arange(rd.size)                           # numpy fancy assignment: last write wins
    np.testing.assert_allclose(flat[uniq], per_point[last[uniq]], rtol=1e-4, atol=1e-4)
    for gd2, gf2 in grads[1:]:
        np.testing.assert_array_equal(gd, gd2)
        np.testing.assert_array_equal(gf, gf2)


def test_layouts_agree_with_reference_wrapper_ops(cuda, oracle_lib):
    """(B,C,Z,Y,X) from the fused kernel == QuickCumsumCuda + permute (the reference's wrapper,
    bev_pool.py:86-92); the collapsed form == cat(unbind(dim=2), 1) (view_transformer.py:194)."""
    cfg = synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4']
    depth, feat = helpers.pool_inputs(cfg)
    rb, rd, rf, st, ln = helpers.lss_ranks(cfg)
    X, Y, Z = cfg.bev_xyz
    shape = (1, Z, Y, X, cfg.channels)
    t = [_dev(x, cuda) for x in (depth, feat, rd, rf, rb)]
    st_d, ln_d = _dev(st, cuda), _dev(ln, cuda)
    fused = bevpool.bev_pool_v2(*t, shape, st_d, ln_d)
    legacy = bevpool.QuickCumsumCuda.apply(*t, shape, st_d, ln_d).permute(0, 4, 1, 2, 3).contiguous()
    assert fused.is_contiguous() and fused.shape == (1, cfg.channels, Z, Y, X)
    torch.testing.assert_close(fused, legacy, rtol=0, atol=0)
    coll = bevpool.bev_pool_v2_collapsed(*t, shape, st_d, ln_d)
    torch.testing.assert_close(coll, torch.cat(legacy.unbind(dim=2), 1), rtol=0, atol=0)
    want = oracle_lib.bev_pool_v2(depth, feat, rd, rf, rb, shape, st, ln)
    np.testing.assert_allclose(fused.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    # ragged x tiles (X not a multiple of 64) and several batches
    rng = np.random.default_rng(2)
    B, Zz, Yy, Xx, c = 2, 2, 5, 200, 80
    dep, fe, rd2, rf2, rb2, st2, ln2 = helpers.random_pool_problem(rng, 20000, B * Zz * Yy * Xx, c, skew=False)
    shape2 = (B, Zz, Yy, Xx, c)
    want = oracle_lib.bev_pool_v2(dep.reshape(1, 1, -1, 1, 1), fe.reshape(1, 1, 1, -1, c), rd2, rf2, rb2, shape2, st2, ln2)
    got = _run(cuda, dep.reshape(1, 1, -1, 1, 1), fe.reshape(1, 1, 1, -1, c), rd2, rf2, rb2, shape2, st2, ln2)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['skewed', 'uniform', 'one_interval', 'gaps', 'cfg2'])
def test_planned_pool_is_bit_identical(cuda, case):
    """bevpool.DevicePoolPlan + bev_pool_v2_planned (rank-only phases built once) == the unplanned
    kernel, bit for bit, in both output layouts, call after call."""
    import torch
    from ocrfdet_amd import bevpool, synthetic
    rng = np.random.default_rng(7)
    C = 80
    if case == 'cfg2':
        cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
        rb, rd, rf, st, ln = helpers.ht_ranks(cfg)
        depth, feat = helpers.pool_inputs(cfg)
        X, Y, _ = cfg.bev_xyz
        shape = (cfg.batch, 1, Y, X, C)
        depth, feat = depth.reshape(-1), feat.reshape(-1, C)
    else:
        n_vox = {'one_interval': 1}.get(case, 900)
        n_pts = 20000
        depth, feat, rd, rf, rb, st, ln = helpers.random_pool_problem(rng, n_pts, n_vox, C, skew=(case == 'skewed'))
        if case == 'gaps':                      # drop every third interval: uncovered points in between
            keep = np.ones(len(st), bool)
            keep[::3] = False
            st, ln = st[keep], ln[keep]
        shape = (1, 1, 30, 30, C) if n_vox > 1 else (1, 1, 1, 1, C)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)   # noqa: E731
    d, f = t(depth), t(feat)
    args = (t(rd), t(rf), t(rb), shape, t(st), t(ln))
    plan = bevpool.DevicePoolPlan(*args)
    for layout in (1, 0):
        want = (bevpool.bev_pool_v2_collapsed(d, f, *args) if layout == 1 else bevpool.bev_pool_v2(d, f, *args))
        for _ in range(2):
            got = bevpool.bev_pool_v2_planned(d, f, plan, layout=layout)
            assert torch.equal(got, want), (case, layout)


@pytest.mark.parametrize('which', ['lss', 'ht'])
def test_pooling_with_device_side_lengths_under_autograd(cuda, which):
    """``bev_pool_v2_device_counts_autograd`` (rank vectors at their CAPACITY, lengths on the device: what the index
    preparation hands over without a read-back) against ``bev_pool_v2_collapsed`` on the exact-size vectors, cfg2 at full
    size: the same pooled map and bit-identical gradients — the tail past the device-side count is uninitialised memory and
    must not matter (filled with out-of-range garbage here), and nothing in forward or backward reads the device."""
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    depth, feat = helpers.pool_inputs(cfg)
    rb, rd, rf, st, ln = helpers.lss_ranks(cfg) if which == 'lss' else helpers.ht_ranks(cfg)
    X, Y, Z = cfg.bev_xyz
    shape = (depth.shape[0], Z if which == 'lss' else 1, Y, X, cfg.channels)
    t = lambda a: _dev(np.asarray(a, np.int32), cuda)                                          # noqa: E731
    rng = np.random.default_rng(3)
    pad_p, pad_i = 4097, 513
    junk = lambda n: rng.integers(-2 ** 31, 2 ** 31 - 1, n, dtype=np.int64).astype(np.int32)   # noqa: E731
    caps = [t(np.concatenate((v, junk(pad)))) for v, pad in ((rb, pad_p), (rd, pad_p), (rf, pad_p), (st, pad_i), (ln, pad_i))]
    counts = t(np.array([len(rb), len(st)]))
    og = _dev(rng.standard_normal((shape[0], shape[1] * shape[4], Y, X)).astype(np.float32), cuda)

    d1, f1 = _dev(depth, cuda).requires_grad_(), _dev(feat, cuda).requires_grad_()
    want = bevpool.bev_pool_v2_collapsed(d1, f1, t(rd), t(rf), t(rb), shape, t(st), t(ln))
    want.backward(og)
    d2, f2 = _dev(depth, cuda).requires_grad_(), _dev(feat, cuda).requires_grad_()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        got = bevpool.bev_pool_v2_device_counts_autograd(d2, f2, caps[1], caps[2], caps[0], shape, caps[3], caps[4], counts)
        got.backward(og)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    torch.cuda.synchronize()
    assert float((got - want).detach().abs().max()) <= 1e-4 * float(want.detach().abs().max())
    assert torch.equal(d2.grad, d1.grad) and torch.equal(f2.grad, f1.grad)
    assert float(d1.grad.abs().max()) > 0 and float(f1.grad.abs().max()) > 0
