"""Diagnostic: the drop-in neck (cached geometry, one hipGraph replay per step) under the module's schedule options.
    python tools/ab_neck_r5.py [name=value ...]       e.g.  fork_c_after=heads
Every line: one freshly built and captured module, median of 5 blocks of 100 replays."""
import itertools
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
    nk = hotpath.NeckPath(cfg, dev)
    for k, v in opts.items():
        setattr(nk.module, k, v)
    for _ in range(3):
        nk.step()
    nk.capture()
    cams = [0] * nk.batch
    return timed(lambda: nk.step_graphed(cams))


def parse(v):
    return int(v) if v.lstrip('-').isdigit() else v


if len(sys.argv) > 1:
    opts = {a.split('=')[0]: parse(a.split('=')[1]) for a in sys.argv[1:]}
    print(opts, '%.4f ms' % one(**opts))
else:
    for rep in range(2):
        for fc in ('pools', 'heads', 'render'):
            print('fork_c_after=%-6s  %.4f ms' % (fc, one(fork_c_after=fc)), flush=True)
