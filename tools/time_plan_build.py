"""Diagnostic: the render plan rebuild of the cfg2 step alone (12 views x 520 000 Gaussians).
    python tools/time_plan_build.py [n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev)
plan, f0, nf, g = hp._plans()[0]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(5):
    plan.rebuild(g['cams'])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    plan.rebuild(g['cams'])
torch.cuda.synchronize()
print('rebuild: %.1f us (kept %d, capacity %d)' % (1e6 * (time.perf_counter() - t0) / n, sum(plan.kept), plan.capacity))
