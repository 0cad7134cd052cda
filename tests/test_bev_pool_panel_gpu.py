"""GPU parity of the panel pooling as a latency kernel (csrc/bev_pool_panel.hip: cell-weight pre-pass + cells walked
out of LDS) against the C oracle of bev_pool_v2 (bev_pool_cuda.cu:21-48) within 1e-4, and against the MFMA form of the
same plan (same cell weights, same fmaf order over a panel's rows; a tile's panels are added in another order: 2e-5),
bitwise reproducible run to run."""
import numpy as np
import pytest
import torch

from ocrfdet_amd import bevpool, synthetic
from tests import helpers
from tests.test_bev_pool_mfma_gpu import _oracle, _t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['cfg0_1cam_128x352_bev64x64x4', 'ref_6cam_256x704_bev128x128x1',
                                  'cfg1_6cam_256x704_bev128x128x8', 'cfg2_6cam_2frame_bev200x200_render_hoa',
                                  'cfg4_6cam_8frame_512x1408_bev200x200'])
@pytest.mark.parametrize('branch', ['lss', 'ht'])
def test_config_ranks_match_the_oracle_and_the_mfma_form(cuda, oracle_lib, name, branch):
    """Every configuration of BASELINE.json the panel kernel is benched or shipped at — configs[1] (128 x 128 x 8, the
    0.05-ms line) and configs[4] (512 x 1408 input: 32 x 88 feature map) included —, both branches, both layouts."""
    cfg = synthetic.CONFIGS[name]
    if name.startswith(('cfg2', 'cfg4')):
        cfg = synthetic.PathConfig(**{**cfg.__dict__, 'n_frames': 1, 'render': False, 'hoa': False})
    rb, rd, rf, st, ln = (helpers.lss_ranks if branch == 'lss' else helpers.ht_ranks)(cfg)
    depth, feat = helpers.pool_inputs(cfg)
    X, Y, Z = cfg.bev_xyz
    shape = (cfg.batch, Z if branch == 'lss' else 1, Y, X, cfg.channels)
    for group in (8, 2):
        plan = bevpool.MfmaPoolPlan(_t(rd, cuda), _t(rf, cuda), _t(rb, cuda), shape, group=group)
        for layout in (0, 1):
            got = bevpool.bev_pool_v2_panel(_t(depth, cuda), _t(feat, cuda), plan, layout=layout)
            want = _oracle(oracle_lib, depth, feat, rd, rf, rb, shape, st, ln, layout)
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(got, bevpool.bev_pool_v2_mfma(_t(depth, cuda), _t(feat, cuda), plan, layout=layout), rtol=2e-5, atol=2e-5)
        again = bevpool.bev_pool_v2_panel(_t(depth, cuda), _t(feat, cuda), plan, layout=1)
        assert torch.equal(got, again)                                   # bitwise reproducible


@pytest.mark.parametrize('c', [64, 80, 96, 128])
@pytest.mark.parametrize('yx', [(8, 8), (13, 21), (40, 64)])
def test_random_problems_ragged_grids_and_channel_counts(cuda, oracle_lib, c, yx):
    """Grids that are not multiples of the 8 x 8 tile, several planes, skewed interval lengths (cells of many points,
    tiles of many panels and several units), empty tiles."""
    rng = np.random.default_rng(c + yx[0])
    Y, X = yx
    B, Z = 2, 3
    n_vox = B * Z * Y * X
    depth, feat, rd, rf, rb, st, ln = helpers.random_pool_problem(rng, 30000, n_vox, c, n_feat=900)
    shape = (B, Z, Y, X, c)
    plan = bevpool.MfmaPoolPlan(_t(rd, cuda), _t(rf, cuda), _t(rb, cuda), shape, group=2)
    for layout in (0, 1, 2) if c == 80 else (1,):
        got = bevpool.bev_pool_v2_panel(_t(depth, cuda), _t(feat, cuda), plan, layout=layout)
        if layout == 2:
            want = _oracle(oracle_lib, depth, feat, rd, rf, rb, shape, st, ln, 0)          # (B, C, Z, Y, X) -> voxel rows
            want = np.ascontiguousarray(want.transpose(0, 2, 3, 4, 1)).reshape(-1, c)
            got = got.reshape(-1, c)
        else:
            want = _oracle(oracle_lib, depth, feat, rd, rf, rb, shape, st, ln, layout)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=2e-4)
    a, b = (f(_t(depth, cuda), _t(feat, cuda), plan, layout=1) for f in (bevpool.bev_pool_v2_panel, bevpool.bev_pool_v2_mfma))
    torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6 * float(b.abs().max()) + 1e-6)
    assert torch.equal(a, bevpool.bev_pool_v2_panel(_t(depth, cuda), _t(feat, cuda), plan, layout=1))


def test_dense_panels_take_the_quarter_window_path(cuda, oracle_lib):
    """Every (voxel, row) cell of a tile present: 64 x 48 = 3 072 cells per panel, three times the LDS cell window."""
    rng = np.random.default_rng(5)
    B, Z, Y, X, c = 1, 1, 8, 16, 80
    n_rows = 100                                       # > 2 panels per tile
    v, r = np.meshgrid(np.arange(Y * X), np.arange(n_rows), indexing='ij')
    rb = np.repeat(v.ravel(), 2).astype(np.int32)      # two points per cell
    rf = np.repeat(r.ravel(), 2).astype(np.int32)
    order = np.argsort(rb, kind='stable')
    rb, rf = rb[order], rf[order]
    rd = rng.integers(0, 5000, rb.size).astype(np.int32)
    depth = rng.random(5000, dtype=np.float32)
    feat = rng.standard_normal((n_rows, c)).astype(np.float32)
    st = np.flatnonzero(np.r_[True, rb[1:] != rb[:-1]]).astype(np.int32)
    ln = np.diff(np.r_[st, rb.size]).astype(np.int32)
    shape = (B, Z, Y, X, c)
    plan = bevpool.MfmaPoolPlan(_t(rd, cuda), _t(rf, cuda), _t(rb, cuda), shape, group=8)
    assert int((plan.panel_cell_off[1:] - plan.panel_cell_off[:-1]).max()) == 64 * 48
    got = bevpool.bev_pool_v2_panel(_t(depth, cuda), _t(feat, cuda), plan, layout=1)
    want = _oracle(oracle_lib, depth, feat, rd, rf, rb, shape, st, ln, 1)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(got, bevpool.bev_pool_v2_mfma(_t(depth, cuda), _t(feat, cuda), plan, layout=1), rtol=2e-5, atol=1e-4)
    assert torch.equal(got, bevpool.bev_pool_v2_panel(_t(depth, cuda), _t(feat, cuda), plan, layout=1))


def test_empty_and_single_point_inputs(cuda):
    shape = (1, 1, 16, 16, 80)
    e = torch.zeros(0, dtype=torch.int32, device=cuda)
    depth = torch.rand(100, device=cuda)
    feat = torch.randn(10, 80, device=cuda)
    plan = bevpool.MfmaPoolPlan(e, e, e, shape)
    out = bevpool.bev_pool_v2_panel(depth, feat, plan, layout=1, out=torch.full((1, 80, 16, 16), 7.0, device=cuda))
    assert float(out.abs().max()) == 0.0                              # every tile is written, empty ones as zeros
    one = torch.tensor([37], dtype=torch.int32, device=cuda)
    plan = bevpool.MfmaPoolPlan(torch.tensor([5], dtype=torch.int32, device=cuda),
                                torch.tensor([3], dtype=torch.int32, device=cuda), one, shape)
    out = bevpool.bev_pool_v2_panel(depth, feat, plan, layout=1)
    want = torch.zeros(1, 80, 16, 16, device=cuda)
    want[0, :, 37 // 16, 37 % 16] = depth[5] * feat[3]
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-7)


def test_one_weight_launch_serves_two_plans(cuda):
    cfg = synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4']
    depth, feat = helpers.pool_inputs(cfg)
    X, Y, Z = cfg.bev_xyz
    plans = []
    for ranks, z in ((helpers.lss_ranks, Z), (helpers.ht_ranks, 1)):
        rb, rd, rf, st, ln = ranks(cfg)
        plans.append(bevpool.MfmaPoolPlan(_t(rd, cuda), _t(rf, cuda), _t(rb, cuda), (cfg.batch, z, Y, X, cfg.channels), group=8))
    d, f = _t(depth, cuda), _t(feat, cuda)
    want = [bevpool.bev_pool_v2_panel(d, f, p) for p in plans]
    for p in plans:
        p.cw.fill_(float('nan'))
    bevpool.bev_pool_cell_weights(d, plans[0], plans[1])
    got = [bevpool.bev_pool_v2_panel(d, f, p, weights_ready=True) for p in plans]
    assert all(torch.equal(a, b) for a, b in zip(got, want))
