"""Diagnostic: is the cfg2 step bound by the host?  (a) issue time per step with the queue kept EMPTY (a device
synchronize before every step: nothing to wait for, so the call's duration is pure host work), (b) the usual back-to-back
loop: issue time and wall per step, (c) the same with the GPU work cut (no render / tiny blend) to see whether the wall
follows the device or the host."""
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
for _ in range(30):
    hp.step(depth, feat)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hp.step(depth, feat)
    ts.append(time.perf_counter() - t0)
ts.sort()
print('host work of one step, queue empty: median %.1f us  p10 %.1f  p90 %.1f' % (1e6 * ts[100], 1e6 * ts[20], 1e6 * ts[180]))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        hp.step(depth, feat)
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print('back to back: issue %.1f us/step, wall %.1f us/step' % (1e6 * ti / 300, 1e6 * ta / 300))
# per-call host durations inside the back-to-back loop (does the host ever wait?)
torch.cuda.synchronize()
d = []
t_prev = time.perf_counter()
for _ in range(300):
    hp.step(depth, feat)
    t = time.perf_counter()
    d.append(t - t_prev)
    t_prev = t
torch.cuda.synchronize()
d = sorted(d[20:])
print('per-call duration in the loop: median %.1f us  p10 %.1f  p90 %.1f  max %.1f' % (1e6 * d[len(d) // 2], 1e6 * d[len(d) // 10], 1e6 * d[9 * len(d) // 10], 1e6 * d[-1]))
