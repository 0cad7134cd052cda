"""Diagnostic workload: cfg2 steps as bench.py issues them, nothing else.  _steps_only.py [n] [cached|per_sample|per_sample_plan]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
mode = sys.argv[2] if len(sys.argv) > 2 else 'cached'
kw = {'cached': {}, 'guard_device': dict(render_guard='device'), 'per_sample': dict(index_prep_mode='per_step', device_geometry=True, render_mode='per_call'),
      'per_sample_plan': dict(index_prep_mode='per_step', device_geometry=True, plan_rebuild='per_step')}[mode]
hp = hotpath.HotPath(cfg, dev, **kw)
depth, feat = hp.make_inputs()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    hp.step(depth, feat)
torch.cuda.synchronize()
