"""Diagnostic driver: only the two planned poolings at cfg2, a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
if len(sys.argv) > 1:
    _lib.lib().ocrf_tune_set(0, int(sys.argv[1]))
hp = hotpath.HotPath(cfg, torch.device('cuda:0'))
depth, feat = hp.make_inputs()
for _ in range(6):
    hp.pool(hp.lss, depth, feat)
    hp.pool(hp.ht, depth, feat)
torch.cuda.synchronize()
