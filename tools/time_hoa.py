"""Diagnostic: kernel-by-kernel and whole-call time of the HOA stages at the cfg2 shape (B = 2, 13 / 80 x 200 x 200).
    python tools/time_hoa.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, torch.device('cuda:0'))
depth, feat = hp.make_inputs(seed=0)
geom = hp.pool(hp.ht, hp.ht_weight if hasattr(hp, 'ht_weight') else depth, feat) if False else None
out = hp.step(depth, feat)
torch.cuda.synchronize()
ids = {k: getattr(_lib, k) for k in dir(_lib) if k.startswith('K_HOA')}
for name, kid in sorted(ids.items(), key=lambda kv: kv[1]):
    t = _lib.KernelTimer(kid, 256)
    torch.cuda.synchronize()
    t.arm()
    for _ in range(10):
        ob = hp.hoa_opacity_bev()
        hp.hoa_step(out[1], ob)
    torch.cuda.synchronize()
    t.disarm()
    ms = t.read_ms()
    t.close()
    if ms:
        print('%-22s launches/step %4.1f  mean %6.1f us  per step %6.1f us' % (name, len(ms) / 10, 1e3 * sum(ms) / len(ms), 1e2 * sum(ms)))
for label, fn in (('hoa_opacity_bev (HOA-1 + HOA-2)', hp.hoa_opacity_bev),
                  ('hoa_step (HOA-1/2/3)', lambda: hp.hoa_step(out[1]))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    print('%-34s %.1f us per call' % (label, 2e4 * (time.perf_counter() - t0)))
