"""Diagnostic: what the sorted blend's workgroups do at cfg2 (frame 0, 6 views): per-wave phase cycles and record
counts from the instrumented build (ocrf_diag_plan_stats).   python tools/diag_plan_blend.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--grid', type=int, default=0)
ap.add_argument('--fuse', type=int, default=0)
a = ap.parse_args()
dev = torch.device('cuda:0')
_lib.lib().ocrf_tune_set(11, a.grid)
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev, fuse_frames=bool(a.fuse))
H, W = cfg.input_size
gx, gy = (W + 15) // 16, (H + 15) // 16
n_wg = gx * ((gy + 1) // 2) * len(hp.cams) * (hp.batch if a.fuse else 1)
buf = torch.zeros(n_wg * 4 * 8, dtype=torch.int64, device=dev)
hp.render()
torch.cuda.synchronize()
_lib.lib().ocrf_diag_plan_stats(_lib.ptr(buf))
hp._render_planned(hp.render_plans[0])
torch.cuda.synchronize()
_lib.lib().ocrf_diag_plan_stats(None)
s = buf.cpu().numpy().reshape(n_wg, 4, 8).astype(np.float64)
cyc = s[:, :, :3]
tot = cyc.sum(2)
print('workgroups', n_wg, ' kept per view', hp.render_plans[0][0].kept)
print('time per wave in 10-ns ticks (s_memrealtime, 100 MHz): scan %.0f stage %.0f blend %.0f  -> shares %s' % (
    cyc[..., 0].mean(), cyc[..., 1].mean(), cyc[..., 2].mean(), np.round(cyc.mean((0, 1)) / cyc.mean((0, 1)).sum(), 3)))
wg_time = tot.max(1)
print('workgroup time: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f' % (
    wg_time.mean(), *np.percentile(wg_time, [50, 90, 99]), wg_time.max()))
scanned, staged, listed, evald = s[:, 0, 3], s[:, 0, 4], s[:, :, 5], s[:, :, 6]
print('per workgroup: scanned %.0f  staged %.0f  | per wave listed %.1f  evaluated %.1f' % (
    scanned.mean(), staged.mean(), listed.mean(), evald.mean()))
print('listed / staged %.3f   evaluated / listed %.3f   max-wave evaluated / mean-wave evaluated %.3f' % (
    listed.mean() / staged.mean(), evald.mean() / listed.mean(), evald.max(1).mean() / evald.mean()))
print('pixel-records evaluated per launch %.3e (128 px per wave-record)' % (evald.sum() * 128))
# blend-phase imbalance: time of the slowest wave vs the mean wave in a workgroup
b = cyc[..., 2]
print('blend cycles: slowest wave / mean wave per workgroup %.3f' % (b.max(1).mean() / b.mean()))
end = s[:, 0, 7]
start = end - tot[:, 0]
print('launch span by the tile pairs own clocks: %.1f us; tile pair mean %.2f us; sum of tile-pair time %.1f us = %.1f us per slot on %d slots' % (
    (end.max() - start.min()) / 100.0, tot[:, 0].mean() / 100.0, tot[:, 0].sum() / 100.0, tot[:, 0].sum() / 100.0 / (a.grid or 1280), a.grid or 1280))
gyp = (gy + 1) // 2
w_idx = np.arange(n_wg)
row = (w_idx % (gx * gyp)) // gx
view = w_idx // (gx * gyp)
print('mean tile-pair time (us) by tile-pair row, top to bottom:', [round(tot[row == r, 0].mean() / 100.0, 1) for r in range(gyp)])
print('mean tile-pair time (us) by view:', [round(tot[view == v, 0].mean() / 100.0, 1) for v in range(int(view.max()) + 1)])
print('start time (us) by view:', [round((start[view == v].mean() - start.min()) / 100.0, 1) for v in range(int(view.max()) + 1)])
order = np.argsort(start)
print('start of tile pair #0, #500, #1000, #1500, #2000, last (us after the first): ', [round((start[order[i]] - start.min()) / 100.0, 1) for i in (0, 500, 1000, 1500, 2000, len(order) - 1) if i < len(order)])
ts = np.linspace(start.min(), end.max(), 25)
print('tile pairs running at 24 instants across the span:', [int(((start <= t) & (end > t)).sum()) for t in ts[:-1]])
print('end of the same: ', [round((end[order[i]] - start.min()) / 100.0, 1) for i in (0, 500, 1000, 1500, 2000, len(order) - 1) if i < len(order)])
# wall time of the instrumented launch vs the cycles its workgroups report: effective clock x slot utilisation
_lib.lib().ocrf_diag_plan_stats(_lib.ptr(buf))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
hp._render_planned(hp.render_plans[0], phase='both')
e1.record()
torch.cuda.synchronize()
_lib.lib().ocrf_diag_plan_stats(None)
ms = e0.elapsed_time(e1)
slots = a.grid or int(_lib.lib().ocrf_diag_plan_resident())
print('instrumented update + blend: %.1f us; sum of workgroup cycles / %d slots = %.0f cycles -> %.2f GHz-equivalents if the slots were busy throughout'
      % (1e3 * ms, slots, tot.max(1).sum() / slots, tot.max(1).sum() / slots / (1e3 * ms) / 1e3))
