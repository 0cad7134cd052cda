"""Diagnostic: MIOpen fp32 3x3 convolutions of the neck (ProbNet at BEV 200x200, ResizeNetwork's stem) in NCHW
vs channels_last, default vs benchmark (find) mode.   python tools/time_conv_formats.py"""
import time

import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
shapes = [('ProbNet 80->40 @2x200x200', (2, 80, 200, 200), 40), ('ProbNet 40->40 @2x200x200', (2, 40, 200, 200), 40),
          ('stem 256->128 @12x16x44', (12, 256, 16, 44), 128), ('stem 64->32 @12x32x88', (12, 64, 32, 88), 32)]


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for name, shp, co in shapes:
        x = torch.randn(*shp, device=dev)
        w = torch.randn(co, shp[1], 3, 3, device=dev)
        xl, wl = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            a = t(lambda: F.conv2d(x, w, None, padding=1))
            b = t(lambda: F.conv2d(xl, wl, None, padding=1))
        print(f'benchmark={bench}  {name:28s} NCHW {a:7.1f} us   channels_last {b:7.1f} us')
