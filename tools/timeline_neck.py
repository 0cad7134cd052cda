"""Diagnostic: where the strands of the neck's hipGraph replay actually are in time, WITHOUT a profiler
(rocprofv3's per-kernel signals perturb the overlap): one-thread stamp kernels (ocrf_diag_stamp,
100 MHz device clock) at the strand boundaries of ``_core_fused``, captured into the graph.

    python tools/timeline_neck.py [config] [--serial]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

NAMES = ['core start (cur)', 'B: colours sampled', 'B: end (NeRF branch, alpha volume, gt)', 'A: both pools done',
         'C: fusion done', 'C: ProbNet done', 'C: geometry gate done', 'A: heads done (before C is forked)', 'A: render done',
         'A: joined B, weighted images', 'A: HOA-1 done', 'A: HOA-2 done', 'end (joined C, HOA-3)']


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    cfg = synthetic.CONFIGS[args[0] if args else 'cfg2_6cam_2frame_bev200x200_render_hoa']
    dev = torch.device('cuda:0')
    neck = hotpath.NeckPath(cfg, dev, accelerate=True, parallel_branches='--serial' not in sys.argv)
    stamps = torch.zeros(16, dtype=torch.int64, device=dev)
    neck.module._transient['stamps'] = stamps
    neck.capture()
    for _ in range(10):
        neck.step_graphed()
    torch.cuda.synchronize()
    acc = torch.zeros(13, dtype=torch.float64)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        neck.step_graphed()
        torch.cuda.synchronize()
        s = stamps[:13].cpu().double()
        acc += (s - s[0]) / 100.0              # 100 MHz -> us
    wall = 1e3 * (time.perf_counter() - t0) / n
    for name, t in zip(NAMES, (acc / n).tolist()):
        print(f'{t:8.1f} us  {name}')
    print(f'({wall:.3f} ms per replay including the stamp read-back)')


if __name__ == '__main__':
    main()
