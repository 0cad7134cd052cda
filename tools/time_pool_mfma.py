"""Diagnostic: the MFMA panel pooling against the tile kernel on the cfg2 rank vectors (both frames, LSS and HT):
max |difference|, plan statistics and device time of each.     python tools/time_pool_mfma.py [--group 4] [--config NAME]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, bevpool, hotpath, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cfg2_6cam_2frame_bev200x200_render_hoa')
ap.add_argument('--group', type=int, default=4)
ap.add_argument('--iters', type=int, default=50)
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = synthetic.PathConfig(**{**synthetic.CONFIGS[a.config].__dict__, 'render': False, 'hoa': False})
hp = hotpath.HotPath(cfg, dev, ht_pool_backend='tile')      # hp.pool = the VALU tile kernel for both rank sets
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


for name, pl in (('lss', hp.lss), ('ht', hp.ht)):
    ref = hp.pool(pl, depth, feat)
    mp = bevpool.MfmaPoolPlan(pl.ranks_depth, pl.ranks_feat, pl.ranks_bev, pl.bev_shape, group=a.group)
    got = bevpool.bev_pool_v2_mfma(depth, feat, mp)
    torch.cuda.synchronize()
    err = (got - ref).abs().max().item()
    print(f'{name}: points {mp.n_points} tiles {mp.n_tiles} units {mp.n_units} panels {mp.n_panels} unique (tile,row) '
          f'{mp.unique_rows} cells {mp.n_cells} slab slices {mp.n_slab_slices}  max|mfma - tile| {err:.3e} '
          f'(max |ref| {ref.abs().max().item():.3f})')
    got2 = bevpool.bev_pool_v2_mfma(depth, feat, mp)
    print('   bitwise reproducible:', bool(torch.equal(got, got2)))
    print(f'   tile kernel {wall(lambda: hp.pool(pl, depth, feat)):7.1f} us   mfma {wall(lambda: bevpool.bev_pool_v2_mfma(depth, feat, mp)):7.1f} us')
    import numpy as np
    buf = torch.zeros(mp.n_units * 8, dtype=torch.int64, device=dev)
    _lib.lib().ocrf_diag_pool_mfma_stamps(_lib.ptr(buf))
    bevpool.bev_pool_v2_mfma(depth, feat, mp)
    torch.cuda.synchronize()
    _lib.lib().ocrf_diag_pool_mfma_stamps(None)
    st = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    tot = st[:, :6].sum(1)
    print('   stamps (cycles per unit): F+zero %.0f cells %.0f mfma %.0f tile->lds %.0f (-) %.0f slab + write %.0f | unit total mean %.0f p50 %.0f p99 %.0f max %.0f'
          % (*st[:, :6].mean(0), tot.mean(), *np.percentile(tot, [50, 99]), tot.max()))
    print('   panels per unit: mean %.2f max %d; per panel: F+zero %.0f cells %.0f mfma %.0f' % (
        st[:, 6].mean(), st[:, 6].max(), st[:, 0].sum() / st[:, 6].sum(), st[:, 1].sum() / st[:, 6].sum(), st[:, 2].sum() / st[:, 6].sum()))
    top = np.argsort(-tot)[:5]
    for u in top:
        print('   unit %5d: total %.0f = F+zero %.0f cells %.0f mfma %.0f tile->lds %.0f (-) %.0f slab + write %.0f | panels %d slices of its tile %d'
              % (u, tot[u], *st[u, :6], st[u, 6], st[u, 7]))
