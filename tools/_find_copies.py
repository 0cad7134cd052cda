import sys, torch
sys.path.insert(0, '/root/repo')
from ocrfdet_amd import hotpath, synthetic
from torch.profiler import profile, ProfilerActivity
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, torch.device('cuda:0'))
depth, feat = hp.make_inputs()
for _ in range(3): hp.step(depth, feat)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    hp.step(depth, feat)
    torch.cuda.synchronize()
for e in prof.events():
    n = e.name
    if any(s in n.lower() for s in ('memcpy', 'copy', 'memset', 'aten::to', 'aten::item', 'aten::fill', 'aten::zero', 'aten::mul', 'aten::add', 'aten::cat', 'aten::contiguous', 'aten::clone')):
        st = [s for s in (e.stack or []) if 'ocrfdet_amd' in s or 'hotpath' in s]
        print(n, e.device_type, round(e.cpu_time_total, 1), st[:3])
