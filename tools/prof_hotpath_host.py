"""Diagnostic: host (Python + launch) time of one hot-path step vs its wall time."""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, torch.device('cuda:0'), overlap='--no-overlap' not in sys.argv)
depth, feat = hp.make_inputs()
for _ in range(20): hp.step(depth, feat)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): hp.step(depth, feat)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'issue {1e3 * t_issue / 200:.3f} ms/step, wall {1e3 * t_all / 200:.3f} ms/step')
pr = cProfile.Profile(); pr.enable()
for _ in range(50): hp.step(depth, feat)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
