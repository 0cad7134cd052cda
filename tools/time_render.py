"""Diagnostic: per-kernel time of the render stage for both camera conventions."""
import os, sys, statistics, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_6cam_2frame_bev200x200_render_hoa'
cfg = synthetic.CONFIGS[name]
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev, render_mode='per_call')      # (the per-call chain's kernels are what is timed)
r = synthetic.rig(cfg.n_cams, cfg.input_size, hp.batch)
for conv in ('reference', 'corrected'):
    hp._prepare_render(r, conv)
    outs = hp.render(want_n_contrib=True); torch.cuda.synchronize()
    o = outs[0]
    nv = [(o['radii'][v] > 0).sum().item() for v in range(o['radii'].shape[0])]
    print(conv, 'visible per view', nv, 'mean final_T', o['final_T'].mean().item(), 'max n_contrib', o['n_contrib'].max().item(),
          'mean n_contrib', o['n_contrib'].float().mean().item())
    for kid, kname in ((_lib.K_RASTER_PREPROCESS, 'preprocess'), (_lib.K_RASTER_SCAN, 'scan'), (_lib.K_RASTER_GATHER, 'gather'), (_lib.K_RASTER_BLEND, 'blend')):
        t = _lib.KernelTimer(kid, 40); torch.cuda.synchronize(); t.arm()
        for _ in range(10): hp.render()
        torch.cuda.synchronize(); t.disarm(); ms = t.read_ms(); t.close()
        print('  %-10s median %.1f us (n=%d)' % (kname, 1e3 * statistics.median(ms), len(ms)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hp.render()
    torch.cuda.synchronize(); print('  whole render() of %d frames x %d views: %.1f us' % (hp.batch, len(hp.cams), 1e5 * (time.perf_counter() - t0)))
