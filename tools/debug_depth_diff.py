"""Diagnostic: planned vs per-call median depth on the objects set (cfg2, one frame); the C oracle as the referee."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402
import oracle  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.PathConfig(**{**synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa'].__dict__, 'n_frames': 1, 'hoa': False})
hp = hotpath.HotPath(cfg, dev, gaussians='objects', alternate=True)
planned = hp.render()[0]
per_call = hotpath.HotPath.render(hp, want_n_contrib=True)[0]
torch.cuda.synchronize()
for k in ('color', 'depth', 'final_T'):
    d = (planned[k] != per_call[k])
    print(k, 'differing', int(d.sum()), 'of', d.numel())
from ocrfdet_amd import _lib
for label, head in (('second render (adapted head)', 0), ('third', 0), ('head forced full', 1 << 30), ('head none: all deferred', -1), ('head 256', 256)):
    _lib.lib().ocrf_tune_set(13, head)
    again = hp.render()[0]
    torch.cuda.synchronize()
    dd = (again['depth'] != per_call['depth'])
    print(label, 'depth differing', int(dd.sum()), [tuple(x) for x in dd.nonzero()[:4].tolist()])
_lib.lib().ocrf_tune_set(13, 0)
hp.plan_bins = None
hp.render_plans = None
nb = hp.render()[0]
torch.cuda.synchronize()
dd = (nb['depth'] != per_call['depth'])
print('no bins', 'depth differing', int(dd.sum()), [tuple(x) for x in dd.nonzero()[:4].tolist()])
d = (planned['depth'] != per_call['depth'])[:, 0]
idx = d.nonzero()
print('views with diffs', torch.unique(idx[:, 0]).tolist())
v = int(idx[0, 0])
oracle.build()
lib = oracle.lib() if hasattr(oracle, 'lib') else None
g, rc = hp.gauss, hp.render_cams
H, W = cfg.input_size
xyz = hp.voxel_xyz[0].reshape(-1, 3)
want = oracle.rasterize_forward(xyz.cpu().numpy(), g['rgb'].cpu().numpy(), g['opacity'].cpu().numpy(), g['scales'].cpu().numpy(),
                                g['rotations'].cpu().numpy(), rc['vm'][v].cpu().numpy(), rc['pm'][v].cpu().numpy(), rc['tfx'][v],
                                rc['tfy'][v], H, W, np.zeros(3, np.float32))
wd = torch.from_numpy(want['depth']).reshape(H, W)
for i in range(min(12, idx.shape[0])):
    _, y, x = idx[i].tolist()
    if idx[i, 0] != v:
        continue
    print('px', y, x, 'planned', float(planned['depth'][v, 0, y, x]), 'per_call', float(per_call['depth'][v, 0, y, x]),
          'oracle', float(wd[y, x]), 'final_T', float(planned['final_T'][v, y, x]), 'n_contrib', int(per_call['n_contrib'][v, y, x]))
pd, cd = planned['depth'][v, 0].cpu(), per_call['depth'][v, 0].cpu()
print('view', v, 'planned != oracle', int((pd != wd).sum()), ' per_call != oracle', int((cd != wd).sum()))
