"""Diagnostic: what the two launches behind the first pass of the planned blend cost the cfg2 step (ocrf_tune_set(15, 1):
single pass — valid only while no tile pair runs out of prepared records, as on the init set).
    python tools/ab_single_pass.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs(seed=0)
L = _lib.lib()


def timed(n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        hp.step(depth, feat)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(30):
    hp.step(depth, feat)
for rep in range(3):
    for single in (0, 1):
        L.ocrf_tune_set(15, single)
        hp._compiled.clear(), hp._warm_keys.clear()
        for _ in range(10):
            hp.step(depth, feat)
        print('single pass' if single else 'two passes ', ' '.join('%.4f' % timed() for _ in range(5)), 'ms', flush=True)
L.ocrf_tune_set(15, 0)
hp.check_render_plans()
