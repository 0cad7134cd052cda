"""Diagnostic: configs[4] (48 views of 512 x 1408) under the blend's in-step grid.    python tools/sweep_cfg4_grid.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402
dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg4_6cam_8frame_512x1408_bev200x200']


def timed(fn, steps=30, blocks=5):
    for _ in range(5):
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


for bw in [int(a) if a.isdigit() else a for a in (sys.argv[1:] or ['auto', '576', '832', '1024'])]:
    hp = hotpath.HotPath(cfg, dev, blend_workgroups=bw)
    depth, feat = hp.make_inputs()
    print('blend_workgroups=%s  %.4f ms' % (bw, timed(lambda: hp.step(depth, feat))), flush=True)
    hp.check_render_plans()
    del hp
