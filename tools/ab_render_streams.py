"""Diagnostic: the two frames' renders on ONE side stream (product) vs one side stream per frame."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs()
sides = [torch.cuda.Stream(dev) for _ in range(2)]
def step(mode):
    cur = torch.cuda.current_stream(dev)
    for s in sides: s.wait_stream(cur)
    if mode == 'one':
        r = hp.render([sides[0]] * hp.batch)
    else:
        r = hp.render(sides)
    ob = hp.hoa_opacity_bev()          # the product's order: HOA-1/2 first
    lss, ht = hp.pool_step(depth, feat)
    out = hp.hoa_step(ht, ob)
    for s in sides: cur.wait_stream(s)
    return r, lss, out
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for rnd in range(3):
    for mode in ('one', 'two'):
        print(mode, '%.3f ms/step' % timeit(lambda: step(mode)))
def renders_only(mode):
    cur = torch.cuda.current_stream(dev)
    for s in sides: s.wait_stream(cur)
    r = hp.render([sides[0]] * hp.batch) if mode == 'one' else hp.render(sides)
    for s in sides: cur.wait_stream(s)
for mode in ('one', 'two'):
    print('renders only', mode, '%.3f ms' % timeit(lambda: renders_only(mode)))
