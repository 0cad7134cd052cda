"""Diagnostic: the cfg2 step and its chains alone, each issued by ONE host call per repetition (ocrf_hotpath_step), wall
clock over back-to-back repetitions; the render chain alone by its kernels' device timers.
    python tools/chains_r5.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e6)
    return float(np.median(ts))


hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs(0)
print('step (two streams, one host call)   %7.1f us' % timed(lambda: hp.step(depth, feat)))


def recorded(fn):
    """fn's library calls as one replayable step on the current stream (fn ran once before)."""
    cur = torch.cuda.current_stream(dev)
    rec, pool = _lib.StepRecorder([cur]), torch.cuda.MemPool()
    with torch.cuda.use_mem_pool(pool, dev), rec:
        keep = fn()
    step = rec.build()
    return (lambda: step.run(cur.cuda_stream)), (keep, pool)


aux = hotpath.HotPath(cfg, dev, overlap=False, one_call=False)
aux._main_chain(depth, feat)
run_main, keep1 = recorded(lambda: aux._main_chain(depth, feat))
print('main chain alone (pools + HOA)      %7.1f us' % timed(run_main))
run_hoa, keep2 = recorded(lambda: aux.hoa_opacity_bev())
print('  HOA-1/2 alone                     %7.1f us' % timed(run_hoa))
pools = hotpath.HotPath(synthetic.PathConfig(**{**cfg.__dict__, 'render': False, 'hoa': False}), dev)
print('  poolings alone                    %7.1f us' % timed(lambda: pools.step(depth, feat)))
one = hotpath.HotPath(cfg, dev, overlap=False)
print('both chains in a row (one stream)   %7.1f us' % timed(lambda: one.step(depth, feat)))
for grid in (0, 704, 1024):
    _lib.lib().ocrf_tune_set(11, grid)
    out = {}
    for name, kid in (('head', _lib.K_RASTER_PLAN_UPDATE), ('blend', _lib.K_RASTER_BLEND_SORTED)):
        t = _lib.KernelTimer(kid, 64)
        t.arm()
        for _ in range(30):
            one.render()
        torch.cuda.synchronize()
        t.disarm()
        out[name] = round(float(np.median(t.read_ms())) * 1e3, 1)
        t.close()
    print('render chain alone, blend grid %5d: head %5.1f + blend %6.1f us (+ the extent check behind it)' % (
        grid, out['head'], out['blend']))
_lib.lib().ocrf_tune_set(11, 0)
hp.check_render_plans(), one.check_render_plans()
