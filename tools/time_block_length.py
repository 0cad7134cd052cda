"""Diagnostic: the cfg2 step timed in blocks of K steps bracketed by synchronisations (what bench.py --steps K does):
block time = a + b K — b is the steady-state period, a what a block pays once (the cold first step, the last step's tail).

    python tools/time_block_length.py [--render-mode planned]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cfg2_6cam_2frame_bev200x200_render_hoa')
ap.add_argument('--render-mode', default='planned')
a = ap.parse_args()
dev = torch.device('cuda:0')
hp = hotpath.HotPath(synthetic.CONFIGS[a.config], dev, render_mode=a.render_mode)
depth, feat = hp.make_inputs(seed=0)
for _ in range(40):
    hp.step(depth, feat)
torch.cuda.synchronize()


def block(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        hp.step(depth, feat)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0)


Ks = (1, 2, 5, 10, 20, 50, 100, 200)
med = {}
for K in Ks:
    ts = sorted(block(K) for _ in range(21))
    med[K] = (ts[10], ts[0])
    print(f'K = {K:3d}: block {ts[10]:9.1f} us median ({ts[0]:9.1f} min)  -> {ts[10] / K:7.2f} us per step ({ts[0] / K:7.2f})')
x = np.array(Ks[2:], float)
for name, i in (('median', 0), ('min', 1)):
    y = np.array([med[K][i] for K in Ks[2:]])
    b, c = np.polyfit(x, y, 1)
    print(f'{name}: block = {c:.1f} us + {b:.2f} us x K')
