import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, torch.device('cuda:0'))
depth, feat = hp.make_inputs()
for _ in range(3): hp.step(depth, feat)
torch.cuda.synchronize()
