"""Diagnostic: the Gaussian heads kernel (neck_ops.gauss_heads) alone at the cfg2 neck shape, device time by HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, neck_ops, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
nk = hotpath.NeckPath(cfg, dev)
m = nk.module
X, Y, _ = cfg.bev_xyz
B, Zh, C = nk.batch, m.num_height, m.out_channels
ht = torch.randn(B, C, Y, X, device=dev)
rgb = torch.rand(B, Zh, Y * X, 3, device=dev) * 255
prm = m._head_params()
for _ in range(5):
    out = neck_ops.gauss_heads(ht, rgb, prm, Zh)
ts = []
for _ in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = neck_ops.gauss_heads(ht, rgb, prm, Zh)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print('gauss_heads alone: median %.1f us  min %.1f us; checksum %.6f' % (np.median(ts), np.min(ts), float(sum(o.double().sum() for o in out))))
