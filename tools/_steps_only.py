"""Diagnostic workload: cfg2 steps as bench.py issues them (default HotPath), nothing else."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    hp.step(depth, feat)
torch.cuda.synchronize()
