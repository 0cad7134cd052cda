"""GPU parity of the MFMA panel pooling (csrc/bev_pool_mfma.hip, bevpool.MfmaPoolPlan) against the C oracle of
bev_pool_v2 (bev_pool_cuda.cu:21-48) and against the tile kernel: within 1e-4 (the sums run in another order: W cells
first, then an exact f32 MFMA chain over the tile's rows), bitwise reproducible run to run."""
import numpy as np
import pytest
import torch

from ocrfdet_amd import bevpool, synthetic
from tests import helpers

pytestmark = pytest.mark.gpu


def _t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _oracle(oracle_lib, depth, feat, rd, rf, rb, shape, st, ln, layout):
    want = oracle_lib.bev_pool_v2(depth, feat, rd, rf, rb, shape, st, ln)        # (B, C, Z, Y, X)
    if layout == 1:
        want = np.concatenate([want[:, :, z] for z in range(want.shape[2])], 1)
    return want


@pytest.mark.parametrize('name', ['cfg0_1cam_128x352_bev64x64x4', 'ref_6cam_256x704_bev128x128x1',
                                  'cfg2_6cam_2frame_bev200x200_render_hoa'])
@pytest.mark.parametrize('branch', ['lss', 'ht'])
def test_config_ranks_match_the_oracle(cuda, oracle_lib, name, branch):
    cfg = synthetic.CONFIGS[name]
    if name.startswith('cfg2'):
        cfg = synthetic.PathConfig(**{**cfg.__dict__, 'n_frames': 1, 'render': False, 'hoa': False})
    rb, rd, rf, st, ln = (helpers.lss_ranks if branch == 'lss' else helpers.ht_ranks)(cfg)
    depth, feat = helpers.pool_inputs(cfg)
    X, Y, Z = cfg.bev_xyz
    shape = (cfg.batch, Z if branch == 'lss' else 1, Y, X, cfg.channels)
    plan = bevpool.MfmaPoolPlan(_t(rd, cuda), _t(rf, cuda), _t(rb, cuda), shape, group=2 if branch == 'lss' else 8)
    for layout in (0, 1):
        got = bevpool.bev_pool_v2_mfma(_t(depth, cuda), _t(feat, cuda), plan, layout=layout)
        want = _oracle(oracle_lib, depth, feat, rd, rf, rb, shape, st, ln, layout)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    again = bevpool.bev_pool_v2_mfma(_t(depth, cuda), _t(feat, cuda), plan, layout=1)
    assert torch.equal(got, again)                                   # bitwise reproducible (slabs added in slice order)


@pytest.mark.parametrize('c', [64, 80, 128])
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
    assert plan.n_slab_slices > 0 or Y * X <= 64
    for layout in (0, 1):
        got = bevpool.bev_pool_v2_mfma(_t(depth, cuda), _t(feat, cuda), plan, layout=layout)
        want = _oracle(oracle_lib, depth, feat, rd, rf, rb, shape, st, ln, layout)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=2e-4)


def test_empty_and_single_point_inputs(cuda):
    shape = (1, 1, 16, 16, 80)
    e = torch.zeros(0, dtype=torch.int32, device=cuda)
    depth = torch.rand(100, device=cuda)
    feat = torch.randn(10, 80, device=cuda)
    plan = bevpool.MfmaPoolPlan(e, e, e, shape)
    out = bevpool.bev_pool_v2_mfma(depth, feat, plan, layout=1, out=torch.full((1, 80, 16, 16), 7.0, device=cuda))
    assert float(out.abs().max()) == 0.0                              # every tile is written, empty ones as zeros
    one = torch.tensor([37], dtype=torch.int32, device=cuda)
    plan = bevpool.MfmaPoolPlan(torch.tensor([5], dtype=torch.int32, device=cuda),
                                torch.tensor([3], dtype=torch.int32, device=cuda), one, shape)
    out = bevpool.bev_pool_v2_mfma(depth, feat, plan, layout=1)
    want = torch.zeros(1, 80, 16, 16, device=cuda)
    want[0, :, 37 // 16, 37 % 16] = depth[5] * feat[3]
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-7)


def test_hot_path_ht_pool_backends_agree(cuda):
    from ocrfdet_amd import hotpath
    cfg = synthetic.PathConfig(**{**synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa'].__dict__,
                                  'render': False, 'hoa': False})
    a = hotpath.HotPath(cfg, cuda, lss_pool_backend='tile', ht_pool_backend='mfma')
    b = hotpath.HotPath(cfg, cuda, lss_pool_backend='tile', ht_pool_backend='tile')
    depth, feat = a.make_inputs(3)
    la, ha = a.step(depth, feat)[:2]
    lb, hb = b.step(depth, feat)[:2]
    assert torch.equal(la, lb)
    torch.testing.assert_close(ha, hb, rtol=1e-4, atol=1e-5)
    assert a.ht.mfma_plan is not None and b.ht.mfma_plan is None
