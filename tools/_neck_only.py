"""Diagnostic workload: the drop-in neck step of bench.py --scope neck (cached geometry, one hipGraph replay per step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
mode = sys.argv[1] if len(sys.argv) > 1 else 'graph'
nk = hotpath.NeckPath(cfg, dev)
for kv in os.environ.get('NECK_OPTS', '').split():          # module schedule options, e.g. NECK_OPTS="schedule=two"
    k, v = kv.split('=')
    setattr(nk.module, k, int(v) if v.isdigit() else v)
for _ in range(3):
    nk.step()
if mode == 'graph':
    nk.capture()
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
        nk.step_graphed()
else:
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
        nk.step()
torch.cuda.synchronize()
