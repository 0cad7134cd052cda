"""Diagnostic: the kernels of ONE eager neck forward in issue order (name, duration, gap to the previous
kernel's end), from the kineto trace.

    python tools/seq_neck.py [config]
"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_6cam_2frame_bev200x200_render_hoa'
    cfg = synthetic.CONFIGS[name]
    neck = hotpath.NeckPath(cfg, torch.device('cuda:0'), accelerate=True)
    for _ in range(5):
        neck.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
        neck.step()
        torch.cuda.synchronize()
    ks = [e for e in p.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ks.sort(key=lambda e: e.time_range.start)
    t_prev, total = None, 0.0
    for e in ks:
        s, d = e.time_range.start, e.time_range.end - e.time_range.start
        gap = 0.0 if t_prev is None else s - t_prev
        t_prev = e.time_range.end
        total += d
        print(f'{d:8.1f} us  gap {gap:7.1f}  {e.name[:110]}')
    print(f'{len(ks)} kernels, {total:.1f} us of kernel time, span {ks[-1].time_range.end - ks[0].time_range.start:.1f} us')


if __name__ == '__main__':
    main()
