"""Diagnostic: the two chains of the cfg2 step alone and together (wall clock over back-to-back repetitions).
    python tools/chains_r4.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs(0)
hp.step(depth, feat)
torch.cuda.synchronize()


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e6)
    ts.sort()
    return ts[2]


def main_chain():
    lss = hp.pool(hp.lss, depth, feat)
    ht = hp.pool(hp.ht, depth, feat)
    ob = hp.hoa_opacity_bev()
    hp.hoa_step(ht, ob)


def pools():
    hp.pool(hp.lss, depth, feat)
    hp.pool(hp.ht, depth, feat)


def hoa12():
    hp.hoa_opacity_bev()


ht = hp.pool(hp.ht, depth, feat)
ob = hp.hoa_opacity_bev()
print('step (overlapped)        %7.1f us' % timed(lambda: hp.step(depth, feat)))
print('main chain alone         %7.1f us' % timed(main_chain))
print('  pools                  %7.1f us' % timed(pools))
print('  HOA-1/2                %7.1f us' % timed(hoa12))
print('  HOA-3                  %7.1f us' % timed(lambda: hp.hoa_step(ht, ob)))
for bw in ('auto', 512, 768, 1024, 0):
    hp.blend_workgroups = bw
    hp.overlap = bw == 'auto'            # 'auto' picks two workgroups per CU only when overlapping
    print('render chain alone bw=%-5s %6.1f us' % (bw, timed(lambda: hp.render())))
