"""Diagnostic: throughput of the cfg2 step with TWO steps in flight on one GPU — two HotPath instances (own plans and
scratch buffers), each on its own caller stream (their render chains share the 'render' side stream or get one each),
steps alternating between them — against one instance."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
own_side = '--own-side' in sys.argv
hps = [hotpath.HotPath(cfg, dev) for _ in range(2)]
if own_side:
    hps[1]._side = [torch.cuda.Stream(dev)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
depth, feat = hps[0].make_inputs()
for i in range(20):
    hps[i % 2].step(depth, feat)
torch.cuda.synchronize()


def blocks(fn, k, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            fn(i)
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / k)
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


def one(i):
    hps[0].step(depth, feat)


def two(i):
    with torch.cuda.stream(streams[i % 2]):
        hps[i % 2].step(depth, feat)


for _ in range(10):
    two(_)
torch.cuda.synchronize()
for rep in range(2):
    for k in (200, 20):
        print('blocks of %3d: one in flight median %.4f (min %.4f max %.4f) | two in flight median %.4f (min %.4f max %.4f)'
              % ((k,) + blocks(one, k) + blocks(two, k)), flush=True)
ref = hps[0].step(depth, feat)
with torch.cuda.stream(streams[1]):
    out = hps[1].step(depth, feat)
torch.cuda.synchronize()
print('outputs equal:', all(torch.equal(a, b) for a, b in ((ref[0], out[0]), (ref[1], out[1]), (ref[3], out[3]), (ref[2][0]['color'], out[2][0]['color']))))
for hp in hps:
    hp.check_render_plans()

# the same with each instance's whole step as one hipGraph (host out of the way)
if '--graphs' in sys.argv:
    graphs = []
    for i in range(2):
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                hps[i].step(depth, feat)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[i]):
            o = hps[i].step(depth, feat)
        graphs.append((g, o))
    torch.cuda.synchronize()

    def one_g(i):
        graphs[0][0].replay()

    def two_g(i):
        with torch.cuda.stream(streams[i % 2]):
            graphs[i % 2][0].replay()

    for _ in range(10):
        two_g(_)
    torch.cuda.synchronize()
    for rep in range(2):
        for k in (200, 20):
            print('GRAPHS blocks of %3d: one in flight median %.4f (min %.4f max %.4f) | two in flight median %.4f (min %.4f max %.4f)'
                  % ((k,) + blocks(one_g, k) + blocks(two_g, k)), flush=True)
    print('graph outputs equal eager:', torch.equal(graphs[1][1][0], ref[0]), torch.equal(graphs[1][1][3], ref[3]))
