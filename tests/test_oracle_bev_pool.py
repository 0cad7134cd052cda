"""CPU: pins the C oracle of bev_pool_v2 (oracle/bev_pool_ref.c) to the reference's one
known-answer test (mmdet3d/ops/bev_pool_v2/bev_pool.py:145-176) and to the index_add identity
that follows from bev_pool_cuda.cu:39-47."""
import numpy as np
import torch

from tests import helpers
from ocrfdet_amd import synthetic


def _kat():
    depth = np.array([0.3, 0.4, 0.2, 0.1, 0.7, 0.6, 0.8, 0.9], np.float32).reshape(1, 1, 2, 2, 2)
    feat = np.ones((1, 1, 2, 2, 2), np.float32)
    rd = np.array([0, 4, 1, 6], np.int32)
    rf = np.array([0, 0, 1, 2], np.int32)
    rb = np.array([0, 0, 1, 1], np.int32)
    return depth, feat, rd, rf, rb


def test_kat_forward(oracle_lib):
    depth, feat, rd, rf, rb = _kat()
    st, ln = oracle_lib.intervals_from_sorted(rb)
    out = oracle_lib.bev_pool_v2(depth, feat, rd, rf, rb, (1, 1, 2, 2, 2), st, ln)
    assert out.shape == (1, 2, 1, 2, 2)                       # (B,C,Z,Y,X)
    assert np.float32(out.sum()) == np.float32(4.4)           # bev_pool.py:168
    raw = oracle_lib.bev_pool_v2_raw(depth, feat, rd, rf, rb, (1, 1, 2, 2, 2), st, ln)
    np.testing.assert_array_equal(raw.reshape(-1, 2)[:2], np.float32([[1.0, 1.0], [1.2, 1.2]]))


def test_kat_backward(oracle_lib):
    depth, feat, rd, rf, rb = _kat()
    out_grad = np.ones((1, 1, 2, 2, 2), np.float32)          # d(sum)/d(out)
    gd, gf = oracle_lib.bev_pool_v2_backward(out_grad, depth, feat, rd, rf, rb)
    np.testing.assert_allclose(gd.reshape(-1), [2., 2., 0., 0., 2., 0., 2., 0.])          # :169-172
    np.testing.assert_allclose(gf.reshape(-1), [1., 1., .4, .4, .8, .8, 0., 0.], rtol=1e-6)  # :173-176


def _index_add(depth, feat, rd, rf, rb, shape):
    B, Z, Y, X, C = shape
    d = torch.from_numpy(depth).reshape(-1)[torch.from_numpy(rd).long()].double()
    f = torch.from_numpy(feat).reshape(-1, C)[torch.from_numpy(rf).long()].double()
    out = torch.zeros(B * Z * Y * X, C, dtype=torch.float64)
    out.index_add_(0, torch.from_numpy(rb).long(), d[:, None] * f)
    return out.view(B, Z, Y, X, C).permute(0, 4, 1, 2, 3).numpy()


def test_index_add_identity_lss_and_ht(oracle_lib):
    cfg = synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4']
    depth, feat = helpers.pool_inputs(cfg)
    X, Y, Z = cfg.bev_xyz
    for (rb, rd, rf, st, ln), shape in ((helpers.lss_ranks(cfg), (1, Z, Y, X, cfg.channels)),
                                        (helpers.ht_ranks(cfg), (1, 1, Y, X, cfg.channels))):
        out = oracle_lib.bev_pool_v2(depth, feat, rd, rf, rb, shape, st, ln)
        ref = _index_add(depth, feat, rd, rf, rb, shape)
        np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)
        assert np.count_nonzero(out.any(axis=1)) <= len(st)   # only interval voxels are written


def test_backward_matches_autograd_of_index_add(oracle_lib):
    rng = np.random.default_rng(3)
    c = 8
    depth, feat, rd, rf, rb, st, ln = helpers.random_pool_problem(rng, 4000, 300, c, n_depth=4000, n_feat=120)
    rd = rng.permutation(4000).astype(np.int32)               # unique depth cells: no store race
    out_grad = rng.standard_normal((1, 1, 1, 300, c)).astype(np.float32)
    gd, gf = oracle_lib.bev_pool_v2_backward(out_grad, depth.reshape(1, 1, 4000, 1, 1), feat.reshape(1, 1, 1, 120, c), rd, rf, rb)
    d = torch.from_numpy(depth).double().requires_grad_()
    f = torch.from_numpy(feat).double().requires_grad_()
    out = torch.zeros(300, c, dtype=torch.float64).index_add(
        0, torch.from_numpy(rb).long(), d[torch.from_numpy(rd).long()][:, None] * f[torch.from_numpy(rf).long()])
    (out * torch.from_numpy(out_grad).double().view(300, c)).sum().backward()
    np.testing.assert_allclose(gd.reshape(-1), d.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gf.reshape(-1, c), f.grad.numpy(), rtol=1e-4, atol=1e-5)
