"""Diagnostic workload: cfg2's twelve planned views, nothing else (for PMC passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev, overlap=False)
if len(sys.argv) > 1:
    _lib.lib().ocrf_tune_set(11, int(sys.argv[1]))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    hp.render()
torch.cuda.synchronize()
