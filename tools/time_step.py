"""Diagnostic: the cfg2 step (HotPath defaults, one host call) and the planned blend's device time in it / alone.
    python tools/time_step.py [--gaussians init] [--blocks 5] [--steps 200]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--gaussians', default='init')
ap.add_argument('--blocks', type=int, default=5)
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--config', default='cfg2_6cam_2frame_bev200x200_render_hoa')
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS[a.config]
hp = hotpath.HotPath(cfg, dev, gaussians=a.gaussians)
depth, feat = hp.make_inputs(seed=0)
k = [0]


def step():
    if hp.alternate:
        hp.set_phase(k[0] & 1)
        k[0] += 1
    hp.step(depth, feat)


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(30):
    step()
ts = [timed(a.steps) for _ in range(a.blocks)]
t = _lib.KernelTimer(_lib.K_RASTER_BLEND_SORTED, 64)
t.arm()
for _ in range(30):
    step()
torch.cuda.synchronize()
t.disarm()
in_step = float(np.median(t.read_ms())) * 1e3
t.close()
hp.overlap = False
hp._compiled.clear(), hp._warm_keys.clear()
for _ in range(6):
    step()
t = _lib.KernelTimer(_lib.K_RASTER_BLEND_SORTED, 64)
t.arm()
for _ in range(30):
    step()
torch.cuda.synchronize()
t.disarm()
alone = float(np.median(t.read_ms())) * 1e3
t.close()
hp.check_render_plans()
print('step median %.4f min %.4f ms | blend in step %.1f us, alone %.1f us' % (float(np.median(ts)), min(ts), in_step, alone), flush=True)
