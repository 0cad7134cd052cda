"""Diagnostic: the cfg2 step in every mode bench.py reports, one process (median of 5 blocks of 100 steps), and the captured neck.
    python tools/modes_r5.py [neck]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']


def timed(fn, steps=100, blocks=5):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return float(np.median(out))


modes = dict(cached={}, per_call=dict(render_mode='per_call'), guard_device=dict(render_guard='device'),
             per_step_devgeom=dict(index_prep_mode='per_step', device_geometry=True),
             per_sample=dict(index_prep_mode='per_step', device_geometry=True, render_mode='per_call'),
             per_sample_plan=dict(index_prep_mode='per_step', device_geometry=True, plan_rebuild='per_step'))
if 'extra' in sys.argv[1:]:
    modes = dict(per_sample=modes['per_sample'],
                 per_sample_2_render_streams=dict(modes['per_sample'], render_streams=2),
                 per_call=modes['per_call'], per_call_2_render_streams=dict(modes['per_call'], render_streams=2))
for name, kw in modes.items():
    hp = hotpath.HotPath(cfg, dev, **kw)
    depth, feat = hp.make_inputs()
    print('%-18s %.4f ms' % (name, timed(lambda: hp.step(depth, feat))), flush=True)
    del hp
if 'neck' in sys.argv[1:]:
    nk = hotpath.NeckPath(cfg, dev)
    for _ in range(3):
        nk.step()
    nk.capture()
    cams = [0] * nk.batch
    print('%-18s %.4f ms' % ('neck (graph)', timed(lambda: nk.step_graphed(cams))), flush=True)
