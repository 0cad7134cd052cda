"""Diagnostic A/B timing of bev_pool kernel variants in ONE process on ONE device
(cdna_hip_programming.md rule 24): interleaved rounds, median + min per variant.
Variants: rounds per slice of a heavy tile (ocrf_tune_set key 0) x XCD-contiguous unit ranges (key 1).
    python tools/ab_bev_pool.py [config ...]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, bevpool, hotpath, synthetic
names = sys.argv[1:] or ['cfg1_6cam_256x704_bev128x128x8', 'cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
L = _lib.lib()
for name in names:
    cfg = synthetic.CONFIGS[name]
    hp = hotpath.HotPath(cfg, dev)
    depth, feat = hp.make_inputs()
    res = {}
    variants = [(r, 1, tv) for tv in (64, 32) for r in (2, 4, 8)]
    for rnd in range(4):
        for rounds, xcd, grid in variants:
            L.ocrf_tune_set(0, rounds), L.ocrf_tune_set(1, xcd), L.ocrf_tune_set(3, grid)
            for pname, plan in (('lss', hp.lss), ('ht', hp.ht)):
                plan.device_plan = None                       # plans are sized for the knob values: rebuild
                for mode in ('planned',):
                    run = (lambda: hp.pool(plan, depth, feat)) if mode == 'planned' else (
                        lambda: bevpool.bev_pool_v2_collapsed(depth, feat, plan.ranks_depth, plan.ranks_feat, plan.ranks_bev,
                                                              plan.bev_shape, plan.starts, plan.lengths))
                    run()
                    t = _lib.KernelTimer(_lib.K_BEV_POOL_FWD, 20)
                    torch.cuda.synchronize()
                    t.arm()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    t.disarm()
                    res.setdefault((rounds, xcd, grid, pname, mode, 'kernel'), []).extend(t.read_ms())
                    res.setdefault((rounds, xcd, grid, pname, mode, 'call'), []).append(e0.elapsed_time(e1) / 20)
                    t.close()
    L.ocrf_tune_set(0, 2), L.ocrf_tune_set(1, 1), L.ocrf_tune_set(3, 64)
    print(name, 'Np lss/ht', hp.lss.n_points, hp.ht.n_points, 'Nv', hp.lss.n_intervals, hp.ht.n_intervals)
    for k in sorted(res):
        v = res[k]
        if k[-1] == 'kernel':
            print('  rounds=%d xcd=%d tv=%-4d %-3s %-9s %-6s median %.1f us  min %.1f us' % (*k, 1e3 * statistics.median(v), 1e3 * min(v)))
