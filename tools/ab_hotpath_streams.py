"""A/B in one process: the hot-path step with the render (side) stream at default vs high HIP stream priority,
and with the main stream at low priority.   python tools/ab_hotpath_streams.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402


def main():
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    dev = torch.device('cuda:0')
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
    print('priority range (least, greatest):', lo, hi)
    variants = {}
    for name, side_prio, main_prio in (('default priorities', 0, None), ('render stream high', -1, None),
                                       ('render stream high, main stream low', -1, 0)):
        hp = hotpath.HotPath(cfg, dev)
        hp._side.append(torch.cuda.Stream(dev, priority=side_prio))
        variants[name] = (hp, torch.cuda.Stream(dev, priority=main_prio) if main_prio is not None else None)
    depth, feat = variants['default priorities'][0].make_inputs()
    res = {k: [] for k in variants}
    for rep in range(5):
        for name, (hp, main) in variants.items():
            ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.current_stream())
            with ctx:
                for _ in range(20):
                    hp.step(depth, feat)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    hp.step(depth, feat)
                torch.cuda.synchronize()
            res[name].append(1e3 * (time.perf_counter() - t0) / 200)
    for name, v in res.items():
        print(f'{name}: ' + ' '.join(f'{t:.3f}' for t in v) + f'  -> min {min(v):.3f} ms')


if __name__ == '__main__':
    main()
