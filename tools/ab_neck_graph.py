"""A/B of neck graph variants in ONE process on one box (box-to-box spread is a few per cent):
alternating timed blocks of replays of each captured variant.

    python tools/ab_neck_graph.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402


def main():
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    dev = torch.device('cuda:0')
    necks = {}
    # edit the variants here: (label, function that configures the NeckPath before capture)
    variants = tuple((f'C forked after the {w}', (lambda w: lambda n: setattr(n.module, 'fork_c_after', w))(w))
                     for w in ('pools', 'heads', 'render'))
    for name, setup in variants:
        n = hotpath.NeckPath(cfg, dev, accelerate=True)
        setup(n)
        n.capture()
        necks[name] = n
    res = {k: [] for k in necks}
    for rep in range(5):
        for name, n in necks.items():
            for _ in range(10):
                n.step_graphed()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                n.step_graphed()
            torch.cuda.synchronize()
            res[name].append(1e3 * (time.perf_counter() - t0) / 100)
    for name, v in res.items():
        print(f'{name}: ' + ' '.join(f'{t:.3f}' for t in v) + f'  -> min {min(v):.3f} ms')


if __name__ == '__main__':
    main()
