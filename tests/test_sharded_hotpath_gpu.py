"""GPU: ``hotpath.ShardedHotPath`` — the camera-frame sharded step of bench.py's N > 1 default.  On one GPU the
collectives are no-ops (world 1) or run over gloo between two processes on the same device (plumbing); what is
checked here is that pooling subsets of cameras into plane blocks of the fused grid and summing them reproduces
the unsharded pools (1e-4: fp32 summation order differs), and that the world-1 path is bitwise the unsharded one."""
import numpy as np
import pytest
import torch

from ocrfdet_amd import hotpath, sharding, synthetic

pytestmark = pytest.mark.gpu


def _small_cfg(n_frames=2):
    base = synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4']
    return synthetic.PathConfig(**{**base.__dict__, 'name': 'small4cam', 'n_cams': 4, 'n_frames': n_frames})


def test_world1_equals_unsharded_step(cuda):
    cfg = _small_cfg()
    hp = hotpath.HotPath(cfg, cuda, overlap=False)
    depth, feat = hp.make_inputs(seed=3)
    lss, ht = hp.step(depth, feat)[:2]
    sp = hotpath.ShardedHotPath(cfg, cuda, 0, 1)
    full = sp.step(sp.make_inputs(seed=3))[0]
    X, Y, Z = cfg.bev_xyz
    assert full.shape == (2, (Z + 1) * cfg.channels, Y, X)
    assert torch.equal(full[:, :Z * cfg.channels], lss) and torch.equal(full[:, Z * cfg.channels:], ht)


@pytest.mark.parametrize('world', [2, 4, 6, 8])
def test_camera_split_partials_sum_to_the_whole(cuda, world):
    """Every rank's partial fused grid, summed the way the exchange sums them (emulated on one device), equals the
    unsharded pools."""
    cfg = _small_cfg()
    hp = hotpath.HotPath(cfg, cuda, overlap=False)
    depth, feat = hp.make_inputs(seed=5)
    lss, ht = hp.step(depth, feat)[:2]
    want = torch.cat((lss, ht), 1)
    X, Y, Z = cfg.bev_xyz
    P = (Z + 1) * cfg.channels
    total = torch.zeros_like(want)
    plan = sharding.CameraFramePlan(cfg.n_cams, 2, world, P)
    for rank in range(world):
        sp = hotpath.ShardedHotPath(cfg, cuda, rank, world)          # no process group: collectives inactive
        assert [sorted(s.cams) for s in sp.subs.values()] == [plan.cams_of(rank, f) for f in plan.frames_of(rank)]
        inputs = sp.make_inputs(seed=5)
        for f, sub in sp.subs.items():
            d, ft = inputs[f]
            part = torch.empty(P, Y, X, device=cuda)
            sub.pool(sub.lss, d, ft, out=part[:Z * cfg.channels])
            sub.pool(sub.ht, d, ft, out=part[Z * cfg.channels:])
            total[f] += part
    torch.testing.assert_close(total, want, rtol=1e-4, atol=1e-4)


def test_touched_strips_cover_every_nonzero_of_a_partial(cuda):
    """The static strip lists of the wedge-sparse exchange (ShardedHotPath -> BevExchange.set_touched): outside them a
    rank's partial grid is exactly zero, and a camera subset touches only a part of the BEV."""
    cfg = _small_cfg()
    X, Y, Z = cfg.bev_xyz
    P = (Z + 1) * cfg.channels
    for world, rank in ((4, 1), (6, 4), (8, 6)):
        sp = hotpath.ShardedHotPath(cfg, cuda, rank, world)          # no process group: set_touched keeps the own lists
        ex = sp.exchange
        inputs = sp.make_inputs(seed=7)
        for f, sub in sp.subs.items():
            assert f in ex.touched
            d, ft = inputs[f]
            tgt = ex.pool_target(f)
            sub.pool(sub.lss, d, ft, out=tgt[:Z * cfg.channels])
            sub.pool(sub.ht, d, ft, out=tgt[Z * cfg.channels:])
            inside = torch.zeros(Y * X, dtype=torch.bool, device=cuda)
            inside[ex._flat[f][0]] = True
            outside = tgt.view(P, Y * X)[:, ~inside]
            assert outside.numel() == 0 or float(outside.abs().max()) == 0.0
            assert 0 < int(ex.touched[f][0].numel()) < ex.n_strips
