"""Diagnostic: the planned render of cfg2's twelve views alone on the device (head + sorted blend), device duration
of each kernel (ocrf_timer_*), under the library's diagnostic knobs.
    python tools/time_blend_r5.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev, overlap=False)
L = _lib.lib()
hp.render()
torch.cuda.synchronize()


def kernels(ids, n=30):
    out = {}
    for name, kid in ids:
        t = _lib.KernelTimer(kid, 64)
        t.arm()
        for _ in range(n):
            hp.render()
        torch.cuda.synchronize()
        t.disarm()
        ms = t.read_ms()
        out[name] = round(float(np.median(ms)) * 1e3, 1)
        t.close()
    return out


ids = [('head', _lib.K_RASTER_PLAN_UPDATE), ('blend', _lib.K_RASTER_BLEND_SORTED)]
for grid in (0, 896, 1024, 1152):
    for head in (0, -1, 16384):
        for variant in (0, 1, 2):
            if (head != 0 or variant != 0) and grid != 0:
                continue
            L.ocrf_tune_set(11, grid), L.ocrf_tune_set(13, head), L.ocrf_tune_set(14, variant)
            for _ in range(3):
                hp.render()
            print(f'grid {grid:5d} head {head:6d} variant {variant}: ', kernels(ids), flush=True)
L.ocrf_tune_set(11, 0), L.ocrf_tune_set(13, 0), L.ocrf_tune_set(14, 0)
hp.check_render_plans()
