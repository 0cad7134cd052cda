"""Diagnostic: the hot-path step with only the MAIN-stream chain (pools + HOA: a linear chain of ~19 launches) replayed
as one hipGraph while the renders are issued eagerly on the side stream, vs everything eager."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs()
for _ in range(20): hp.step(depth, feat)
torch.cuda.synchronize()
def timeit(fn, n=300):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * ti / n, 1e3 * (time.perf_counter() - t0) / n
print('eager issue %.3f wall %.3f ms/step' % timeit(lambda: hp.step(depth, feat)))
def main_chain():
    lss, ht = hp.pool_step(depth, feat)
    return (lss, ht) + tuple(hp.hoa_step(ht))
s = torch.cuda.Stream(dev); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): main_chain()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = main_chain()
torch.cuda.synchronize()
side = torch.cuda.Stream(dev)
def step_mixed():
    cur = torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    r = hp.render([side] * hp.batch)
    g.replay()
    cur.wait_stream(side)
    return r
for _ in range(10): step_mixed()
torch.cuda.synchronize()
print('mixed (main chain graphed, renders eager) issue %.3f wall %.3f ms/step' % timeit(step_mixed))
ref = hp.step(depth, feat); torch.cuda.synchronize()
print('lss equal', torch.equal(ref[0], out[0]), 'gated equal', torch.equal(ref[3], out[2]))
print('main chain graph alone issue %.3f wall %.3f' % timeit(lambda: g.replay()))
print('main chain eager alone issue %.3f wall %.3f' % timeit(main_chain))
print('renders alone (eager, main stream) issue %.3f wall %.3f' % timeit(lambda: hp.render()))
