import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
r = synthetic.rig(cfg.n_cams, cfg.input_size, hp.batch)
hp._prepare_render(r, sys.argv[1] if len(sys.argv) > 1 else 'corrected')
for _ in range(5): hp.render()
torch.cuda.synchronize()
