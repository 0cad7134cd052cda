"""Diagnostic: the per-sample step (ranks rebuilt on the device + per-call render, nothing cached), eager, for a kernel trace.
    rocprofv3 --kernel-trace --stats -- python3 tools/prof_per_sample.py [planned]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
planned = len(sys.argv) > 1 and sys.argv[1] == 'planned'
hp = hotpath.HotPath(cfg, dev, index_prep_mode='per_step', device_geometry=True,
                     render_mode='planned' if planned else 'per_call', plan_rebuild='per_step' if planned else 'never')
depth, feat = hp.make_inputs(0)
for _ in range(10):
    hp.step(depth, feat)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    hp.step(depth, feat)
torch.cuda.synchronize()
print('eager %.1f us per step' % (1e6 * (time.perf_counter() - t0) / 40))
