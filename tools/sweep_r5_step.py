"""Diagnostic: the cfg2 step under HotPath options, one process (median of blocks of 100 steps each).
    python tools/sweep_r5_step.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']


def timed(hp, depth, feat, blocks=5, steps=100):
    for _ in range(20):
        hp.step(depth, feat)
    torch.cuda.synchronize()
    out = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            hp.step(depth, feat)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return float(np.median(out)), min(out)


variants = [dict(blend_workgroups=g) for g in (512, 576, 640, 704, 768, 1024)] if 'grid' in sys.argv[1:] else \
    [dict(blend_workgroups=g) for g in (608, 640, 672, 704, 736, 768)] * 2 if 'grid6' in sys.argv[1:] else \
    [dict(blend_workgroups=g, hoa_first=h) for h in (None, True) for g in (640, 672, 704)] * 2 if 'fine' in sys.argv[1:] else \
    [dict(render_mode='per_call'), dict(render_mode='per_call', lss_pool_backend='tile', ht_pool_backend='mfma'),
     dict(render_mode='per_call', one_call=False), dict(render_guard='device'), dict(render_guard='device', one_call=False),
     dict(render_guard='device', blend_workgroups=896)]
for kw in variants:
    hp = hotpath.HotPath(cfg, dev, **kw)
    depth, feat = hp.make_inputs()
    med, mn = timed(hp, depth, feat)
    hp.check_render_plans()
    print(kw, 'median %.4f ms  min %.4f ms' % (med, mn), 'one call' if hp._compiled else getattr(hp, 'one_call_refused', 'call by call'), flush=True)
    del hp
