"""Diagnostic: the cfg2 step timed the way the driver does (blocks of 20 steps between synchronisations, median of 5; also 200)
— eager vs the main chain (pools + HOA) replayed as one hipGraph with the renders issued eagerly on the side stream."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
backend = sys.argv[1] if len(sys.argv) > 1 else None
hp = hotpath.HotPath(cfg, dev, **({'lss_pool_backend': backend, 'ht_pool_backend': backend} if backend else {}))
print('pooling back end:', backend or 'default (tile LSS, MFMA HT)')
depth, feat = hp.make_inputs()
for _ in range(20):
    hp.step(depth, feat)
torch.cuda.synchronize()


def blocks(fn, k, n=7):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / k)
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


def main_chain():
    return tuple(hp._main_chain(depth, feat))


s = torch.cuda.Stream(dev)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        main_chain()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = main_chain()
torch.cuda.synchronize()
side = hotpath.shared_stream(dev, 'render')


def step_mixed():
    cur = torch.cuda.current_stream(dev)
    hp._set_busy(1)
    side.wait_stream(cur)
    r = hp.render([side] * hp.batch)
    g.replay()
    hp._set_busy(0)
    cur.wait_stream(side)
    return r


for _ in range(10):
    step_mixed()
torch.cuda.synchronize()
for k in (20, 200):
    for rep in range(2):
        print('blocks of %3d: eager median %.4f (min %.4f max %.4f)   main chain as a graph median %.4f (min %.4f max %.4f)'
              % ((k,) + blocks(lambda: hp.step(depth, feat), k) + blocks(step_mixed, k)), flush=True)
ref = hp.step(depth, feat)
torch.cuda.synchronize()
print('lss equal', torch.equal(ref[0], out[0]), 'gated equal', torch.equal(ref[3], out[2]))
