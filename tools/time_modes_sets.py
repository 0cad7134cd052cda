"""Diagnostic: the cfg2 step per render mode on a Gaussian set (alternating parameters).
    python tools/time_modes_sets.py [objects|stress|init]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
kind = sys.argv[1] if len(sys.argv) > 1 else 'objects'
for name, kw in (('planned', {}), ('per_call', dict(render_mode='per_call')), ('guard_device', dict(render_guard='device')),
                 ('per_sample', dict(index_prep_mode='per_step', device_geometry=True, render_mode='per_call'))):
    hp = hotpath.HotPath(cfg, dev, gaussians=kind, alternate=True, **kw)
    depth, feat = hp.make_inputs()
    k = [0]

    def step():
        hp.set_phase(k[0] & 1)
        k[0] += 1
        hp.step(depth, feat)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    hp.check_render_plans()
    print('%-8s %-14s %.4f ms' % (kind, name, float(np.median(ts))), flush=True)
    del hp
