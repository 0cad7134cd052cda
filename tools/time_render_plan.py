"""Diagnostic: the render chain of ONE cfg2 frame (6 views, 520 k Gaussians) alone on the GPU — the per-call
pipeline (zero + preprocess + scan + scatter + blend) against the static render plan (update + sorted blend),
with the device duration of every kernel (ocrf_timer_*), and the SHA-1 of the outputs of both.
    python tools/time_render_plan.py [--wskip 0|1] [--guard host|device] [--config NAME]"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, raster_plan, synthetic  # noqa: E402
from ocrfdet_amd.diff_gaussian_rasterization import rasterize_views  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cfg2_6cam_2frame_bev200x200_render_hoa')
ap.add_argument('--wskip', type=int, default=0)
ap.add_argument('--guard', default='host')
ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--margin', type=float, default=1.25)
ap.add_argument('--grid', type=int, default=0)
a = ap.parse_args()

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS[a.config]
hp = hotpath.HotPath(cfg, dev)
g, rc = hp.gauss, hp.render_cams
H, W = cfg.input_size
xyz = hp.voxel_xyz[0].reshape(-1, 3)
# (the in-loop wave skip knob of round 3 is gone: the planned blend culls by wave at staging)
_lib.lib().ocrf_tune_set(11, a.grid)
print('resident workgroups by the occupancy API:', _lib.lib().ocrf_diag_plan_resident())


def dyn():
    return rasterize_views(xyz, g['rgb'], g['opacity'], g['scales'], g['rotations'], None, None, None, None, H, W, hp.bg,
                           packed_cameras=rc['packed'], want_n_contrib=False)


plan = raster_plan.RasterPlan(xyz, rc['packed'], H, W, scales=g['scales'], rotations=g['rotations'], margin=a.margin)
print('plan: kept per view', plan.kept, 'of', plan.P, 'total', sum(plan.kept), 'capacity', plan.capacity, 'bound', plan.extent_bound)


def planned():
    return plan.render(g['rgb'], g['opacity'], g['scales'], g['rotations'], hp.bg, guard=a.guard)


def sha(o):
    return {k: hashlib.sha1(o[k].cpu().numpy().tobytes()).hexdigest()[:12] for k in ('color', 'depth', 'final_T')}


print('per-call', sha(dyn()))
print('planned ', sha(planned()))
plan.check()


def wall(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


def kernels(fn, ids):
    out = {}
    for name, kid in ids:
        t = _lib.KernelTimer(kid, 64)
        t.arm()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t.disarm()
        ms = t.read_ms()
        out[name] = sum(ms) / max(len(ms), 1) * 1e3
        t.close()
    return out


print(f'per-call chain {wall(dyn):7.1f} us  ', kernels(dyn, [('preprocess', _lib.K_RASTER_PREPROCESS), ('scan', _lib.K_RASTER_SCAN),
                                                              ('scatter', _lib.K_RASTER_GATHER), ('blend', _lib.K_RASTER_BLEND)]))
print(f'planned chain  {wall(planned):7.1f} us  ', kernels(planned, [('update', _lib.K_RASTER_PLAN_UPDATE),
                                                                    ('blend_sorted', _lib.K_RASTER_BLEND_SORTED)]))
plan.check()
