import cProfile, pstats, sys, os, torch, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.time_neck import build, inputs
from ocrfdet_amd import synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
m = build(cfg, dev, False); inp, pre = inputs(cfg, dev)
with torch.no_grad():
    for _ in range(3): m._geometry(inp)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): m._geometry(inp)
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
