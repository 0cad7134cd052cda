"""Diagnostic: the drop-in neck with PER-FORWARD geometry on the device inside the captured graph
(bench.py --scope neck --index-prep per_step --device-geometry) under the module's schedule options.
    python tools/ab_neck_devgeom.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return float(np.median(ts))


def one(**opts):
    nk = hotpath.NeckPath(cfg, dev, accelerate=False)
    nk.module.device_geometry = True
    for k, v in opts.items():
        setattr(nk.module, k, v)
    for _ in range(3):
        nk.step()
    nk.capture()
    cams = [0] * nk.batch
    return timed(lambda: nk.step_graphed(cams))


for rep in range(2):
    for fh in (True, False):
        for fc in ('pools', 'heads'):
            print('fork_ht_prep=%d fork_c_after=%-6s  %.4f ms' % (fh, fc, one(fork_ht_prep=fh, fork_c_after=fc)), flush=True)
