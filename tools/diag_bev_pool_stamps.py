"""Diagnostic (not part of the product path): per-phase cycle shares of bev_pool pass 1."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic
cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg1_6cam_256x704_bev128x128x8']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
depth, feat = hp.make_inputs()
L = _lib.lib()
L.ocrf_diag_bev_pool_v2_stamps.restype = ctypes.c_int
for name, plan in (('lss', hp.lss), ('ht', hp.ht)):
    out = torch.zeros(plan.bev_shape, device=dev)
    nbytes = L.ocrf_bev_pool_v2_workspace_bytes(cfg.channels, plan.n_points)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    nb = (plan.n_points + 767) // 768
    stamps = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
    for it in range(3):
        err = L.ocrf_diag_bev_pool_v2_stamps(
            ctypes.c_int(cfg.channels), ctypes.c_int(plan.n_intervals), ctypes.c_int(plan.n_points),
            _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(plan.ranks_depth), _lib.ptr(plan.ranks_feat),
            _lib.ptr(plan.ranks_bev), _lib.ptr(plan.starts), _lib.ptr(plan.lengths), _lib.ptr(out),
            _lib.ptr(ws), _lib.ptr(stamps), _lib.stream_ptr(dev))
        assert err == 0
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(nb, 8)[:, :6].astype(np.int64)
    d = np.diff(st, axis=1)
    print(name, 'blocks', nb, 'phase cycles median [stage+search, ivl stage, main loop, barrier, combine]:',
          np.median(d, axis=0), 'total median', np.median(st[:, 5] - st[:, 0]),
          'kernel span', st[:, 5].max() - st[:, 0].min())
