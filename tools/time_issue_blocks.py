"""Diagnostic: host issue order of the step's two chains (hp.issue_order) timed the way the driver does: blocks of 20 steps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs()
for _ in range(20):
    hp.step(depth, feat)
torch.cuda.synchronize()


def blocks(k, n):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            hp.step(depth, feat)
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / k)
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


for rep in range(3):
    for o in ('render_first', 'lss_first', 'pools_first'):
        hp.issue_order = o
        for _ in range(5):
            hp.step(depth, feat)
        print('%-13s blocks of 20 (x15): median %.4f min %.4f max %.4f | blocks of 200 (x3): median %.4f' % ((o,) + blocks(20, 15) + blocks(200, 3)[:1]), flush=True)
