"""Diagnostic A/B timing of bev_pool kernel variants in ONE process on ONE device
(cdna_hip_programming.md rule 24): interleaved rounds, median + min per variant."""
import ctypes, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic
names = sys.argv[1:] or ['cfg1_6cam_256x704_bev128x128x8', 'cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
L = _lib.lib()
for name in names:
    cfg = synthetic.CONFIGS[name]
    hp = hotpath.HotPath(cfg, dev)
    depth, feat = hp.make_inputs()
    res = {}
    for rnd in range(5):
        for sub in (32, 64):
            L.ocrf_tune_set(0, sub)
            for kid, kname in ((_lib.K_BEV_POOL_FWD, 'fwd'), (_lib.K_BEV_POOL_FIXUP, 'fix'), (_lib.K_BEV_POOL_NCHW, 'nchw')):
                for pname, plan in (('lss', hp.lss), ('ht', hp.ht)):
                    hp.pool(plan, depth, feat)
                    t = _lib.KernelTimer(kid, 20)
                    torch.cuda.synchronize()
                    t.arm()
                    for _ in range(20):
                        hp.pool(plan, depth, feat)
                    torch.cuda.synchronize()
                    t.disarm()
                    res.setdefault((sub, kname, pname), []).extend(t.read_ms())
                    t.close()
    print(name, 'Np lss/ht', hp.lss.n_points, hp.ht.n_points, 'Nv', hp.lss.n_intervals, hp.ht.n_intervals)
    for k in sorted(res):
        v = res[k]
        print('  sub=%d %-4s %-3s median %.1f us  min %.1f us' % (k[0], k[1], k[2], 1e3 * statistics.median(v), 1e3 * min(v)))
