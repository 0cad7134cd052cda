"""Diagnostic: the cfg2 step timed the way the driver does (blocks of 20 steps between synchronisations) and in blocks
of 200, issued by one host call per step (the default) and call by call (one_call=False).
    python tools/step_blocks_r5.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')


def blocks(fn, k, n=9):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / k)
    return float(np.median(out)), min(out), max(out)


for one_call in (True, False):
    hp = hotpath.HotPath(cfg, dev, one_call=one_call)
    depth, feat = hp.make_inputs()
    for _ in range(30):
        hp.step(depth, feat)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hp.step(depth, feat)
        ts.append(time.perf_counter() - t0)
    print('one_call=%s: host work of one step with the queue empty: median %.1f us' % (one_call, 1e6 * float(np.median(ts))))
    for k in (20, 200):
        print('  blocks of %3d steps: median %.4f  min %.4f  max %.4f ms/step' % ((k,) + blocks(lambda: hp.step(depth, feat), k)))
    hp.check_render_plans()
    del hp
