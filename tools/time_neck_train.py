"""Diagnostic: the neck in TRAINING mode at a bench configuration — forward + backward of
``view_transform_core`` (per-forward index preparation, differentiable torch ops around the HIP
bev_pool_v2 / rasteriser forward + backward kernels), wall time per iteration and the top device ops.

    python tools/time_neck_train.py [config] [--iters N] [--profile]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config', nargs='?', default='cfg2_6cam_2frame_bev200x200_render_hoa')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--profile', action='store_true')
    ap.add_argument('--host', action='store_true', help='cProfile of the host side')
    ap.add_argument('--layers', action='store_true', help="the voxel heads as torch layers (fused_heads_training = False)")
    ap.add_argument('--cudnn-benchmark', action='store_true', help='torch.backends.cudnn.benchmark = True (MIOpen find mode)')
    ap.add_argument('--host-geometry', action='store_true', help='device_geometry = False: the calibration read to the host per forward')
    a = ap.parse_args()
    cfg = synthetic.CONFIGS[a.config]
    dev = torch.device('cuda:0')
    torch.backends.cudnn.benchmark = bool(a.cudnn_benchmark)
    neck = hotpath.NeckPath(cfg, dev, accelerate=False)
    m = neck.module.train()
    m.fused_heads_training = not a.layers
    if a.host_geometry:
        m.device_geometry = False
    pre = neck.depthnet_out
    depth0 = pre[:, :cfg.D].softmax(1)
    feat0 = pre[:, cfg.D + 2:cfg.D + 2 + cfg.channels].clone()

    def it():
        depth = depth0.clone().requires_grad_(True)
        feat = feat0.clone().requires_grad_(True)
        bev, _, logit, lst = m.view_transform_core(neck.inputs, depth, feat)
        loss = bev.square().mean() + logit.square().mean() + lst[0].mean() + lst[6].mean() + lst[4].square().mean()
        loss.backward()
        m.zero_grad(set_to_none=True)
        return loss

    for _ in range(3):
        it()
    # host-bound (~1 500 launches per iteration): the host's clock and placement move the figure by 20 % from block to
    # block on one box, so several blocks are timed and the best and the median are both printed
    blocks = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            it()
        torch.cuda.synchronize()
        blocks.append(1e3 * (time.perf_counter() - t0) / a.iters)
    blocks.sort()
    ms = blocks[0]
    print(f'{cfg.name}: B={neck.batch} training forward + backward: {ms:.2f} ms per iteration '
          f'(best of 7 blocks of {a.iters}; median {blocks[3]:.2f}, worst {blocks[-1]:.2f})')
    if a.host:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            it()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(60)
    if a.profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as p:
            for _ in range(3):
                it()
            torch.cuda.synchronize()
        ka = p.key_averages()
        print(f'device time {sum(e.self_device_time_total for e in ka) / 3e3:.2f} ms/iter, host (self CPU) {sum(e.self_cpu_time_total for e in ka) / 3e3:.2f} ms/iter')
        from torch.autograd import DeviceType
        kern = [e for e in ka if e.device_type == DeviceType.CUDA]
        print(f'kernels alone: {sum(e.self_device_time_total for e in kern) / 3e3:.2f} ms/iter in '
              f'{sum(e.count for e in kern) // 3} launches; host ops: {sum(e.count for e in ka if e.device_type == DeviceType.CPU) // 3}')
        for e in sorted(kern, key=lambda e: -e.self_device_time_total)[:40]:
            print(f'  k {e.self_device_time_total / 3e3:8.3f} ms/iter  n={e.count // 3:4d}  {e.key[:130]}')
        for e in sorted(ka, key=lambda e: -e.self_device_time_total)[:25]:
            print(f'  {e.self_device_time_total / 3e3:8.3f} ms/iter  n={e.count // 3:4d}  {e.key[:120]}')
        rows = [e for e in p.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith('aten::')]
        rows.sort(key=lambda e: -e.device_time_total)
        for e in rows[:40]:
            print(f'{e.device_time_total / 3e3:9.2f} ms/iter  n={e.count // 3:4d}  {e.key:38s} {str(e.input_shapes)[:150]}')


if __name__ == '__main__':
    main()
