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
from ocrfdet_amd import neck_ops, synthetic  # noqa: E402
from ocrfdet_amd import view_transformer_ocrf as vto  # noqa: E402


def build(cfg, dev, accelerate, seed=0):
    torch.manual_seed(seed)
    X, Y, _ = cfg.bev_xyz
    m = vto.OcRFViewTransformerFull(pc_range=list(cfg.pc_range), bev_h=Y, bev_w=X, num_height=cfg.num_height,
                                    grid_config=cfg.grid, input_size=cfg.input_size, downsample=cfg.downsample,
                                    in_channels=256, out_channels=cfg.channels, accelerate=accelerate)
    return m.to(dev).eval()


def inputs(cfg, dev, seed=0):
    B = cfg.batch * cfg.n_frames
    r = synthetic.rig(cfg.n_cams, cfg.input_size, B)
    Hf, Wf = cfg.feat_hw
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg.n_cams, 256, Hf, Wf, generator=g)
    raw = torch.randint(0, 256, (B, cfg.n_cams, 3, *cfg.input_size), generator=g).float()
    inp = [x] + [torch.from_numpy(r[k]) for k in ('rots', 'trans', 'intrins', 'post_rots', 'post_trans', 'bda')]
    inp += [torch.zeros(B, cfg.n_cams, 27), raw, raw, raw, torch.from_numpy(r['c2w'])]
    pre = torch.randn(B * cfg.n_cams, cfg.D + 2 + cfg.channels, Hf, Wf, generator=g)
    pre[:, :cfg.D] *= 3
    return [t.to(dev) for t in inp], pre.to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config', nargs='?', default='cfg2_6cam_2frame_bev200x200_render_hoa')
    ap.add_argument('--accelerate', action='store_true')
    ap.add_argument('--stages', action='store_true')
    ap.add_argument('--iters', type=int, default=30)
    a = ap.parse_args()
    cfg = synthetic.CONFIGS[a.config]
    dev = torch.device('cuda:0')
    m = build(cfg, dev, a.accelerate)
    inp, pre = inputs(cfg, dev)
    B = inp[0].shape[0]

    def step():
        depth, fdepth, sem, feat_cl = neck_ops.prefilter(pre, m.D, m.out_channels, m.depth_threshold, m.semantic_threshold)
        return m.view_transform(inp, fdepth, None, feat_cl)
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
    if a.stages:
        import torch.autograd.profiler as prof
        with torch.no_grad(), prof.profile(use_device='cuda') as p:
            for _ in range(5):
                step()
            torch.cuda.synchronize()
        print(p.key_averages().table(sort_by='device_time_total', row_limit=40, max_name_column_width=70))


if __name__ == '__main__':
    main()
