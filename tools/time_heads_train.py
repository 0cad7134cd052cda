"""Diagnostic: ocrf_gauss_heads / ocrf_gauss_heads_backward alone at a bench shape (HIP events, median of 30).

    python tools/time_heads_train.py [B] [Y] [X]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import neck_ops  # noqa: E402
from ocrfdet_amd import view_transformer_ocrf as vto  # noqa: E402


def med(fn, n=30):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    B, Y, X = (int(v) for v in (sys.argv[1:4] + ['2', '200', '200'][len(sys.argv) - 1:]))
    dev, zh, C = torch.device('cuda:0'), 13, 80
    vfe = vto.VoxelFeatureExtractor(1, zh).to(dev).eval()
    heads = [cls(C, 4, o).to(dev) for cls, o in ((vto.ScaleFactorMLP, 3), (vto.RotationFactorMLP, 4),
                                                 (vto.OpacityFactorMLP, 1), (vto.ColorFactorMLPGaussian, 3))]
    prm = neck_ops.pack_gauss_head_params(vfe, *heads)
    bev = torch.randn(B, C, Y, X, device=dev)
    rgb = torch.rand(B, zh, Y * X, 3, device=dev) * 255
    P = zh * Y * X
    g = [torch.randn(B, P, k, device=dev) for k in (1, 3, 4, 3)]
    for _ in range(3):
        neck_ops.gauss_heads(bev, rgb, prm, zh)
        neck_ops.gauss_heads_backward(bev, rgb, prm, zh, *g)
    f = med(lambda: neck_ops.gauss_heads(bev, rgb, prm, zh))
    b = med(lambda: neck_ops.gauss_heads_backward(bev, rgb, prm, zh, *g))
    rows = B * P
    print(f'B={B} {Y}x{X} Zh={zh} C={C}: {rows} voxel rows | forward {f[0]:.1f} us (min {f[1]:.1f}) | '
          f'backward (zero + kernel + 2 sums) {b[0]:.1f} us (min {b[1]:.1f})')


if __name__ == '__main__':
    main()
