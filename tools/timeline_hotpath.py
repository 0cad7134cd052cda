"""Diagnostic: where the cfg2 step's time goes, on the device clock (ocrf_diag_stamp, 100 MHz): the step's two chains
re-issued here with a stamp after every stage — main stream: HOA-1/2 -> LSS pool -> HT pool -> HOA-3; side stream: per
frame update -> blend.     python tools/timeline_hotpath.py [--bw auto|N]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--bw', default='auto')
ap.add_argument('--ht', default='mfma')
ap.add_argument('--lss', default='tile')
ap.add_argument('--lss-group', type=int, default=2)
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev, ht_pool_backend=a.ht, blend_workgroups='auto' if a.bw == 'auto' else int(a.bw),
                     lss_pool_backend=a.lss, lss_mfma_group=a.lss_group)
depth, feat = hp.make_inputs(0)
for _ in range(10):
    hp.step(depth, feat)
torch.cuda.synchronize()
side = hotpath.shared_stream(dev, 'render')
cur = torch.cuda.current_stream(dev)
names = ['start', 'hoa12 (after the pools)', 'lss', 'ht', 'hoa3', 's_start', 'update (fused) | upd0+blend0', 'blend (fused) | upd1+blend1']
N = 60
stamps = torch.zeros(N, len(names), dtype=torch.int64, device=dev)
for it in range(N):
    row = stamps[it]
    _lib.diag_stamp(row, 0)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        _lib.diag_stamp(row, 5)
        plans = hp._plans()               # one fused plan (both frames) by default, or one per frame
        for k, entry in enumerate(plans):
            if len(plans) == 1:
                o = hp._render_planned(entry, phase='update')
                _lib.diag_stamp(row, 6)
                hp._render_planned(entry, phase='blend', out=o)
                _lib.diag_stamp(row, 7)
            else:
                hp._render_planned(entry)
                _lib.diag_stamp(row, 6 + k)
    lss = hp.pool(hp.lss, depth, feat)
    _lib.diag_stamp(row, 2)
    ht = hp.pool(hp.ht, depth, feat)
    _lib.diag_stamp(row, 3)
    ob = hp.hoa_opacity_bev()
    _lib.diag_stamp(row, 1)
    hp.hoa_step(ht, ob)
    _lib.diag_stamp(row, 4)
    cur.wait_stream(side)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
rel = (s - s[:, :1]) / 100.0        # us at 100 MHz
med = np.median(rel[20:], 0)
print('steady-state period %.1f us' % np.median(np.diff(s[20:, 0]) / 100.0))
for n, v in zip(names, med):
    print(f'{n:32s} {v:8.1f} us')
print('step end       %8.1f us' % max(med[4], med[7]))
