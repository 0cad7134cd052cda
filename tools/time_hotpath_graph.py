"""Diagnostic: the hot-path step (pools + renders on the side stream + HOA) issued eagerly vs replayed as one
hipGraph.  The step is device-bound, not launch-bound: the replay is no faster (round 3: 0.326 vs 0.289 ms eager at cfg2, host
issue 0.23 ms), so bench.py keeps issuing it kernel by kernel (which also lets it time the dominant kernel in the region)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs()
for _ in range(20): hp.step(depth, feat)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): hp.step(depth, feat)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); print('eager %.4f ms/step (host issue %.4f ms/step)' % (1e3 * (time.perf_counter() - t0) / 300, 1e3 * t_issue / 300))
side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): hp.step(depth, feat)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = hp.step(depth, feat)
torch.cuda.synchronize()
for _ in range(10): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): g.replay()
torch.cuda.synchronize(); print('graph %.4f ms/step' % (1e3 * (time.perf_counter() - t0) / 300))
ref = hp.step(depth, feat); torch.cuda.synchronize()
print('lss equal', torch.equal(ref[0], out[0]), 'color equal', torch.equal(ref[2][0]['color'], out[2][0]['color']))
