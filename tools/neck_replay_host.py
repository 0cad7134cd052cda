"""Diagnostic: how long the HOST is inside one replay of the captured neck graph (hipGraphLaunch returns when every
node is enqueued) beside the replay's device time.    python tools/neck_replay_host.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
nk = hotpath.NeckPath(cfg, dev)
for k, v in (a.split('=') for a in sys.argv[1:]):
    setattr(nk.module, k, int(v) if v.isdigit() else v)
for _ in range(3):
    nk.step()
nk.capture()
g = nk._graphed
print('graph census:', getattr(g, 'census', None))
host, total = [], []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g._graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e6)
    total.append((t2 - t0) * 1e6)
print('one replay from idle: host inside replay() %.1f us (min %.1f), until the device is done %.1f us' % (
    np.median(host), np.min(host), np.median(total)))
