"""Diagnostic: the panel pooling as a latency kernel (bev_pool_panel.hip) against the MFMA form and the tile kernel on
the cfg2 rank vectors (both frames, LSS and HT): agreement and device time of each, the weight pre-pass alone, and both
poolings of a step back to back.     python tools/time_pool_panel.py [--group 8] [--config NAME]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import bevpool, hotpath, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cfg2_6cam_2frame_bev200x200_render_hoa')
ap.add_argument('--group', type=int, default=8)
ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--unit-cost', type=float, default=None)
ap.add_argument('--only', default=None)
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = synthetic.PathConfig(**{**synthetic.CONFIGS[a.config].__dict__, 'render': False, 'hoa': False})
hp = hotpath.HotPath(cfg, dev, ht_pool_backend='tile')
depth, feat = hp.make_inputs(0)


def wall(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


plans = {}
for name, pl in (('lss', hp.lss), ('ht', hp.ht)):
    if a.only and name != a.only:
        continue
    ref = hp.pool(pl, depth, feat)
    mp = plans[name] = bevpool.MfmaPoolPlan(pl.ranks_depth, pl.ranks_feat, pl.ranks_bev, pl.bev_shape, group=a.group, unit_cost=a.unit_cost)
    got = bevpool.bev_pool_v2_panel(depth, feat, mp)
    mf = bevpool.bev_pool_v2_mfma(depth, feat, mp)
    torch.cuda.synchronize()
    ncell = (mp.panel_cell_off[1:] - mp.panel_cell_off[:-1]).float()
    print(f'{name}: units {mp.n_units} panels {mp.n_panels} cells {mp.n_cells} (per panel mean {ncell.mean().item():.0f} max {int(ncell.max())}) '
          f'slab slices {mp.n_slab_slices}  max|panel - tile| {(got - ref).abs().max().item():.3e}  max|panel - mfma| {(got - mf).abs().max().item():.3e}')
    print(f'   tile {wall(lambda: hp.pool(pl, depth, feat)):6.1f} us   mfma {wall(lambda: bevpool.bev_pool_v2_mfma(depth, feat, mp)):6.1f} us   '
          f'weights {wall(lambda: bevpool.bev_pool_cell_weights(depth, mp)):6.1f} us   '
          f'panel (weights ready) {wall(lambda: bevpool.bev_pool_v2_panel(depth, feat, mp, weights_ready=True)):6.1f} us   '
          f'weights + panel {wall(lambda: bevpool.bev_pool_v2_panel(depth, feat, mp)):6.1f} us', flush=True)
    import numpy as np
    from ocrfdet_amd import _lib
    buf = torch.zeros(mp.n_units * 8, dtype=torch.int64, device=dev)
    _lib.lib().ocrf_diag_pool_panel_stamps(_lib.ptr(buf))
    bevpool.bev_pool_v2_panel(depth, feat, mp, weights_ready=True)
    torch.cuda.synchronize()
    _lib.lib().ocrf_diag_pool_panel_stamps(None)
    st = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    tot = st[:, :5].sum(1)
    print('   stamps (cycles per unit): header %.0f wait %.0f sums %.0f tile->lds %.0f leave %.0f | unit total mean %.0f p50 %.0f p99 %.0f max %.0f'
          % (*st[:, :5].mean(0), tot.mean(), *np.percentile(tot, [50, 99]), tot.max()))
    npan = max(st[:, 5].sum(), 1)
    print('   per panel: wait %.0f sums %.0f; cells per unit mean %.0f max %.0f' % (st[:, 1].sum() / npan, st[:, 2].sum() / npan, st[:, 6].mean(), st[:, 6].max()))
    for u in np.argsort(-tot)[:4]:
        print('   unit %5d: total %.0f = header %.0f wait %.0f sums %.0f tile->lds %.0f leave %.0f | panels %d cells %d slices %d' % (u, tot[u], *st[u, :5], st[u, 5], st[u, 6], st[u, 7]))


if a.only:
    sys.exit(0)


def both():
    bevpool.bev_pool_cell_weights(depth, plans['lss'], plans['ht'])
    bevpool.bev_pool_v2_panel(depth, feat, plans['lss'], weights_ready=True)
    bevpool.bev_pool_v2_panel(depth, feat, plans['ht'], weights_ready=True)


def both_old():
    hp.pool(hp.lss, depth, feat)
    bevpool.bev_pool_v2_mfma(depth, feat, plans['ht'])


print(f'one weight launch + both panel poolings {wall(both):6.1f} us   (tile LSS + MFMA HT {wall(both_old):6.1f} us)')
