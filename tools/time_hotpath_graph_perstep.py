"""Diagnostic: the per-step hot path with device geometry (calibration algebra + index preparation + pools + renders
+ HOA) issued eagerly vs replayed as one hipGraph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev, index_prep_mode='per_step', device_geometry=True)
depth, feat = hp.make_inputs()
for _ in range(20):
    hp.step(depth, feat)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    hp.step(depth, feat)
torch.cuda.synchronize()
print('eager %.4f ms/step' % (1e3 * (time.perf_counter() - t0) / 300))
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        hp.step(depth, feat)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = hp.step(depth, feat)
torch.cuda.synchronize()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    g.replay()
torch.cuda.synchronize()
print('graph %.4f ms/step' % (1e3 * (time.perf_counter() - t0) / 300))
ref = hp.step(depth, feat)
torch.cuda.synchronize()
print('lss equal', torch.equal(ref[0], out[0]), 'ht equal', torch.equal(ref[1], out[1]))
