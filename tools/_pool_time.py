import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic
for cname in ('cfg2_6cam_2frame_bev200x200_render_hoa', 'cfg1_6cam_256x704_bev128x128x8'):
    cfg = synthetic.CONFIGS[cname]
    cfg = synthetic.PathConfig(**{**cfg.__dict__, 'render': False, 'hoa': False})
    hp = hotpath.HotPath(cfg, torch.device('cuda:0'))
    depth, feat = hp.make_inputs(seed=0)
    for name, plan in (('lss', hp.lss), ('ht', hp.ht)):
        t = _lib.KernelTimer(_lib.K_BEV_POOL_FWD, 64); torch.cuda.synchronize(); t.arm()
        for _ in range(20):
            hp.pool(plan, depth, feat)
        torch.cuda.synchronize(); t.disarm(); ms = t.read_ms(); t.close()
        print('%-40s %-4s tile kernel median %.1f us' % (cname, name, 1e3 * statistics.median(ms)))
