"""Diagnostic: host time of the cfg2 step by section (queue kept empty: a device synchronize before every step)."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, bevpool, hoa, hotpath, raster_plan, synthetic  # noqa: E402

cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs()
for _ in range(30):
    hp.step(depth, feat)
torch.cuda.synchronize()
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0
            cnt[label] += 1
    setattr(obj, name, w)


for n in ('_plans', '_set_busy', 'render', '_main_chain', 'pool_step', 'hoa_opacity_bev', 'hoa_step', '_render_planned'):
    wrap(hp, n)
wrap(torch.cuda.Stream, 'wait_stream', 'Stream.wait_stream')
wrap(raster_plan.RasterPlan, 'render', 'RasterPlan.render')
wrap(bevpool, 'bev_pool_v2_planned')
wrap(bevpool, 'bev_pool_v2_mfma')
L = _lib.lib()
N = 200
tot = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hp.step(depth, feat)
    tot += time.perf_counter() - t0
print('step %.1f us' % (1e6 * tot / N))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-24s %7.1f us/step  (%d calls/step)' % (k, 1e6 * v / N, cnt[k] // N))
# raw costs of the primitives
torch.cuda.synchronize()
s = torch.cuda.Stream(dev)
t0 = time.perf_counter()
for _ in range(1000):
    with torch.cuda.stream(s):
        pass
print('with torch.cuda.stream(s): %.1f us' % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter()
for _ in range(1000):
    s.wait_stream(torch.cuda.current_stream(dev))
print('wait_stream: %.1f us' % (1e3 * (time.perf_counter() - t0)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000):
    torch.empty((2, 80, 200, 200), device=dev)
print('torch.empty: %.1f us' % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter()
for _ in range(1000):
    _lib.stream_ptr(dev)
print('stream_ptr: %.1f us' % (1e3 * (time.perf_counter() - t0)))
x = torch.zeros(16, device=dev)
t0 = time.perf_counter()
for _ in range(1000):
    _lib.ptr(x)
print('ptr: %.2f us' % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter()
for _ in range(1000):
    L.ocrf_tune_set(99, 0)
print('ctypes call (2 ints): %.2f us' % (1e3 * (time.perf_counter() - t0)))
torch.cuda.synchronize()
