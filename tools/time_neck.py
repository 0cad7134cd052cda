"""Diagnostic: the whole neck (``OcRFViewTransformerFull.view_transform_core``) at a bench
configuration with random-init weights of the reference architecture — wall time per forward and,
with --stages, the device time of each stage (torch.cuda.Event pairs around the stage calls).

    python tools/time_neck.py [config] [--accelerate] [--stages] [--iters N]
"""
import argparse
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config', nargs='?', default='cfg2_6cam_2frame_bev200x200_render_hoa')
    ap.add_argument('--accelerate', action='store_true')
    ap.add_argument('--stages', action='store_true')
    ap.add_argument('--iters', type=int, default=30)
    a = ap.parse_args()
    cfg = synthetic.CONFIGS[a.config]
    dev = torch.device('cuda:0')
    neck = hotpath.NeckPath(cfg, dev, accelerate=a.accelerate)
    B, step = neck.batch, neck.step
    random.seed(0)
    with torch.no_grad():
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.iters
    X, Y, Z = cfg.bev_xyz
    print(f'{cfg.name}: B={B} accelerate={a.accelerate}: {ms:.3f} ms per neck forward '
          f'({B * Z * Y * X / ms * 1e3:.3e} BEV voxels/s, {B / ms * 1e3:.1f} rendered views/s)')
    if a.accelerate:
        neck.capture()
        for _ in range(5):
            neck.step_graphed()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            neck.step_graphed()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.iters
        print(f'{cfg.name}: B={B} as ONE hipGraph: {ms:.3f} ms per neck forward '
              f'({B * Z * Y * X / ms * 1e3:.3e} BEV voxels/s, {B / ms * 1e3:.1f} rendered views/s)')
    if a.stages:
        import torch.autograd.profiler as prof
        with torch.no_grad(), prof.profile(use_device='cuda', record_shapes=True) as p:
            for _ in range(5):
                step()
            torch.cuda.synchronize()
        print(p.key_averages(group_by_input_shape=True).table(sort_by='device_time_total', row_limit=60, max_name_column_width=50,
                                                            max_shapes_column_width=90))


if __name__ == '__main__':
    main()
