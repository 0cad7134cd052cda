import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev, render_mode='per_call')
r = synthetic.rig(cfg.n_cams, cfg.input_size, hp.batch)
L = _lib.lib()
ntile = 44 * 8 * 6      # one workgroup per vertical pair of 16x16 tiles
for conv in ('reference', 'corrected'):
    hp._prepare_render(r, conv)
    hp.render(); torch.cuda.synchronize()
    buf = torch.zeros(ntile * 8, dtype=torch.int64, device=dev)
    L.ocrf_diag_raster_stamps(ctypes.c_void_p(buf.data_ptr()))
    hp.batch_save = hp.batch; hp.batch = 1
    hp.render(); torch.cuda.synchronize()
    hp.batch = hp.batch_save
    L.ocrf_diag_raster_stamps(None)
    st = buf.cpu().numpy().reshape(ntile, 8)
    print(conv, 'per-tile cycles mean [scan, sort, ready, blend, carry]:', st[:, :5].mean(0).round(0), 'total mean', st[:, :5].sum(1).mean(),
          'max', st[:, :5].sum(1).max(), 'entries scanned mean/max', st[:, 5].mean(), st[:, 5].max(), 'records consumed mean/max', st[:, 6].mean(), st[:, 6].max())
    tot = st[:, :5].sum(1).reshape(6, 8, 44)
    print(conv, 'mean total cycles per tile-pair row (top -> bottom):', tot.mean((0, 2)).round(0))
    print(conv, 'mean total cycles per view:', tot.mean((1, 2)).round(0))
