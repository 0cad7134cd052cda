"""Diagnostic: partition the chip between the hot path's two chains instead of hoping streams interleave.
The renders (VALU-bound blend) run on one HIP stream, pools + HOA (latency-bound) on another; each stream is created
with a CU mask (hipExtStreamCreateWithCUMask) or a priority.  ONE variant per process (the streams of a process share
a few hardware queues in creation order, so variants must not see each other's streams):
    python tools/ab_cu_mask.py --variant plain|prio_render|prio_main|split:R[:M]|where [--steps 200]
split:R     mask bits [0,R) for the render stream, [R,256) for the main stream (a bit is one CU; bit i lies on XCC i % 8,
            so a split is the same share of every XCD); split:R:M = main gets bits [256-M,256) (overlap allowed);
            R or M = 0: that stream unmasked."""
import argparse
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--config', default='cfg2_6cam_2frame_bev200x200_render_hoa')
ap.add_argument('--variant', default='plain')
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
L = _lib.lib()

if a.variant == 'where':
    def where(stream, n=4096):
        out = torch.zeros(n, dtype=torch.int32, device=dev)
        _lib.check(L.ocrf_diag_where(n, _lib.ptr(out), 2000, _lib.ctypes.c_void_p(stream.cuda_stream)), 'where')
        torch.cuda.synchronize()
        v = out.cpu().numpy().astype('uint32')
        xcc, cu, se, sh = v >> 16, (v >> 8) & 0xF, (v >> 13) & 0x7, (v >> 12) & 1
        per = collections.Counter(int(x) for x in xcc)
        return dict(sorted(per.items())), len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
    for name, bits in (('bits 0-7', range(8)), ('bits 0-31', range(32)), ('bits i%8==0', range(0, 256, 8)),
                       ('bits 0-191', range(192)), ('bits 192-255', range(192, 256))):
        print('mask', name, '-> workgroups per XCC, distinct CUs:', *where(_lib.masked_stream(dev, bits)), flush=True)
    sys.exit(0)

render_stream = main_stream = None
if a.variant == 'prio_render':
    render_stream = _lib.masked_stream(dev, None, -1)
elif a.variant == 'prio_main':
    render_stream = torch.cuda.Stream(dev)
    main_stream = _lib.masked_stream(dev, None, -1)
elif a.variant.startswith('split:'):
    parts = [int(x) for x in a.variant.split(':')[1:]]
    R = parts[0]
    M = parts[1] if len(parts) > 1 else 256 - R
    render_stream = _lib.masked_stream(dev, range(R)) if R else torch.cuda.Stream(dev)
    main_stream = _lib.masked_stream(dev, range(256 - M, 256)) if M else None
if render_stream is not None:
    hotpath._STREAMS[(0, 'render')] = render_stream

cfg = synthetic.CONFIGS[a.config]
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs(0)
with torch.cuda.stream(main_stream if main_stream is not None else torch.cuda.current_stream(dev)):
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
print(f'{a.variant:24s} median {ts[2]:.4f}  min {ts[0]:.4f}  max {ts[-1]:.4f} ms/step', flush=True)
