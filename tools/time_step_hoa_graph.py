"""Diagnostic: the cfg2 step with ONLY HOA-1/2 (eight launches, ~90 us of host work) replayed as a hipGraph inside the
otherwise eager step — does the step follow the host's issue time?"""
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


def blocks(fn, k, n=7):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        ti = time.perf_counter() - t0
        torch.cuda.synchronize()
        out.append((1e3 * (time.perf_counter() - t0) / k, 1e3 * ti / k))
    out.sort()
    return out[len(out) // 2]


ref = hp.step(depth, feat)
torch.cuda.synchronize()
print('eager:            blocks of 200: wall %.4f issue %.4f | blocks of 20: wall %.4f issue %.4f'
      % (blocks(lambda: hp.step(depth, feat), 200) + blocks(lambda: hp.step(depth, feat), 20)), flush=True)
# capture HOA-1/2
orig = hp.hoa_opacity_bev
s = torch.cuda.Stream(dev)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        orig()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ob_static = orig()
torch.cuda.synchronize()


def graphed(defer=False):
    g.replay()
    return ob_static


hp.hoa_opacity_bev = graphed
for _ in range(10):
    out = hp.step(depth, feat)
torch.cuda.synchronize()
print('HOA-1/2 as graph: blocks of 200: wall %.4f issue %.4f | blocks of 20: wall %.4f issue %.4f'
      % (blocks(lambda: hp.step(depth, feat), 200) + blocks(lambda: hp.step(depth, feat), 20)), flush=True)
print('gated equal', torch.equal(ref[3], out[3]), 'opacity bev equal', torch.equal(ref[4], out[4]))
for o in ('pools_first',):
    hp.issue_order = o
    for _ in range(10):
        hp.step(depth, feat)
    print('  + issue %s: blocks of 200: wall %.4f issue %.4f | blocks of 20: wall %.4f issue %.4f'
          % ((o,) + blocks(lambda: hp.step(depth, feat), 200) + blocks(lambda: hp.step(depth, feat), 20)), flush=True)
