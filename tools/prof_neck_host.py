"""Diagnostic: cProfile of the host side of one neck step (launch-bound at small batch)."""
import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
neck = hotpath.NeckPath(cfg, torch.device('cuda:0'), accelerate='--per-step' not in sys.argv)
for _ in range(5): neck.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): neck.step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
