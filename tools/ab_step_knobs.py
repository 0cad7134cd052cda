"""Diagnostic: the cfg2 step time under library knobs (ocrf_tune_set key=value pairs), one process per setting:
    python tools/ab_step_knobs.py 11=768 [10=1 ...] [--ht tile|mfma] [--steps 200]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('knobs', nargs='*')
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--ht', default='mfma')
ap.add_argument('--render-mode', default='planned')
ap.add_argument('--fuse', type=int, default=1)
ap.add_argument('--rstreams', type=int, default=1)
ap.add_argument('--hoa-first', type=int, default=None, help='default: what HotPath chooses')
ap.add_argument('--caller', type=int, default=None, help='render chain on the caller stream (1) or on the side stream (0; HotPath default)')
ap.add_argument('--hoa-stream', type=int, default=0)
ap.add_argument('--lss', default='tile')
ap.add_argument('--issue', default=None, help='host issue order: render_first (default) | lss_first | pools_first')
ap.add_argument('--stats-stream', default=None, help="HOA-3's channel statistics beside HOA-1/2: render | own (default: on the main chain)")
ap.add_argument('--lss-group', type=int, default=2)
ap.add_argument('--schedule', default=None, help='phased | overlap (default: what HotPath chooses)')
ap.add_argument('--fuse-out', type=int, default=None, help="HOA-2's output conv inside the HOA-3 gate (1, HotPath default) or as its own launch (0)")
ap.add_argument('--bw', default='auto', help="blend workgroups: auto | n | n0,n1 (per frame)")
a = ap.parse_args()
for kv in a.knobs:
    k, v = kv.split('=')
    _lib.check(_lib.lib().ocrf_tune_set(int(k), int(v)), 'tune')
dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev, ht_pool_backend=a.ht, render_mode=a.render_mode, fuse_frames=bool(a.fuse), render_streams=a.rstreams, lss_pool_backend=a.lss, lss_mfma_group=a.lss_group,
                     blend_workgroups=('auto' if a.bw == 'auto' else (int(a.bw) if ',' not in a.bw else [int(x) for x in a.bw.split(',')])))
if a.hoa_first is not None:
    hp.hoa_first = bool(a.hoa_first)
if a.schedule is not None:
    hp.schedule = a.schedule
if a.fuse_out is not None:
    hp.fuse_out_conv = bool(a.fuse_out)
if a.caller is not None:
    hp.render_on_caller_stream = bool(a.caller)
hp.hoa_stream = bool(a.hoa_stream)
if a.issue:
    hp.issue_order = a.issue
if a.stats_stream:
    hp.stats_stream = a.stats_stream
depth, feat = hp.make_inputs(0)
for _ in range(30):
    hp.step(depth, feat)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(a.steps):
        hp.step(depth, feat)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / a.steps * 1e3)
hp.check_render_plans()
ts.sort()
print([round(t,4) for t in ts], end=" "); print(f'{" ".join(a.knobs) or "defaults":20s} ht={a.ht} render={a.render_mode} fuse={a.fuse} rstreams={a.rstreams} bw={a.bw} fuse_out={a.fuse_out} schedule={a.schedule} caller={a.caller} hoa_first={a.hoa_first} hoa_stream={a.hoa_stream} lss={a.lss}/{a.lss_group} stats_stream={a.stats_stream} issue={a.issue}: median {ts[2]:.4f} min {ts[0]:.4f} max {ts[-1]:.4f} ms/step', flush=True)
