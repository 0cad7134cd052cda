"""Diagnostic (not part of the product path): per-phase cycle shares of the pooling kernel per work unit
(ocrf_diag_bev_pool_stamps): table + zero-fill, staging, gather, combine, write-out."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, bevpool, hotpath, synthetic
cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs()
L = _lib.lib()
L.ocrf_tune_set(0, int(sys.argv[2]) if len(sys.argv) > 2 else 4)
L.ocrf_tune_set(3, int(sys.argv[3]) if len(sys.argv) > 3 else 64)
for name, plan in (('lss', hp.lss), ('ht', hp.ht)):
    B, Z, Y, X, C = plan.bev_shape
    dp = bevpool.DevicePoolPlan(plan.ranks_depth, plan.ranks_feat, plan.ranks_bev, plan.bev_shape, plan.starts, plan.lengths)
    out = torch.empty(B, Z * C, Y, X, device=dev)
    ws = torch.empty(L.ocrf_bev_pool_planned_workspace_bytes(C, plan.n_points), dtype=torch.uint8, device=dev)
    nu = L.ocrf_bev_pool_max_units(C, plan.n_points, B, Z, Y, X)
    stamps = torch.zeros(nu * 8, dtype=torch.int64, device=dev)
    for it in range(3):
        err = L.ocrf_diag_bev_pool_stamps(C, plan.n_points, _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(dp.ranks_depth),
                                          _lib.ptr(dp.ranks_feat), _lib.ptr(dp.plan), _lib.ptr(out), B, Z, Y, X, 1, _lib.ptr(ws),
                                          _lib.ptr(stamps), _lib.stream_ptr(dev))
        assert err == 0
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(nu, 8)
    st = st[st[:, :5].sum(1) > 0]
    tot = st[:, :5].sum(1)
    print(name, 'units', len(st), 'points/unit median %d max %d' % (np.median(st[:, 5]), st[:, 5].max()))
    print('   cycles per unit: median %d  p90 %d  max %d' % (np.median(tot), np.percentile(tot, 90), tot.max()))
    print('   shares [table+zero, staging, gather, combine, write-out]:', np.round(st[:, :5].sum(0) / tot.sum(), 3))
    heavy = st[st[:, 5] >= np.percentile(st[:, 5], 90)]
    print('   heaviest 10 %% of the units:', np.round(heavy[:, :5].sum(0) / heavy[:, :5].sum(), 3), 'median cycles', int(np.median(heavy[:, :5].sum(1))))
