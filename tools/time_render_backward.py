"""Diagnostic: per-kernel time of the rasteriser backward at the bench workload's render shape, and its
agreement with the C oracle on one view."""
import os, sys, statistics, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic
from ocrfdet_amd import diff_gaussian_rasterization as dgr
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_6cam_2frame_bev200x200_render_hoa'
cfg = synthetic.CONFIGS[name]
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev)
rc, g = hp.render_cams, hp.gauss
H, W = cfg.input_size
xyz = hp.voxel_xyz[0].reshape(-1, 3)
fwd = dgr.rasterize_views(xyz, g['rgb'], g['opacity'], g['scales'], g['rotations'], rc['vm'], rc['pm'], rc['tfx'], rc['tfy'], H, W, hp.bg)
V = fwd['color'].shape[0]
gcol = torch.randn(V, 3, H, W, device=dev, generator=torch.Generator(dev).manual_seed(0))
def bwd():
    return dgr.rasterize_views_backward(gcol, fwd, xyz, g['rgb'], g['opacity'], g['scales'], g['rotations'], rc['vm'], rc['pm'],
                                        rc['tfx'], rc['tfy'], H, W, hp.bg)
out = bwd(); torch.cuda.synchronize()
print('P', xyz.shape[0], 'views', V, 'grad norms', {k: float(v.norm()) for k, v in out.items()})
for kid, kname in ((_lib.K_RASTER_BLEND_BWD, 'blend_bwd'), (_lib.K_RASTER_PRE_BWD, 'pre_bwd')):
    t = _lib.KernelTimer(kid, 40); torch.cuda.synchronize(); t.arm()
    for _ in range(10): bwd()
    torch.cuda.synchronize(); t.disarm(); ms = t.read_ms(); t.close()
    print('  %-10s median %.1f us (n=%d)' % (kname, 1e3 * statistics.median(ms), len(ms)))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): bwd()
torch.cuda.synchronize(); print('  whole backward of %d views: %.1f us' % (V, 1e5 * (time.perf_counter() - t0)))
if '--check' in sys.argv:
    import oracle
    n = lambda t: t.detach().cpu().numpy()
    tot = None
    t0 = time.perf_counter()
    for v in range(V):
        w = oracle.rasterize_backward(n(gcol[v]), n(xyz), n(g['rgb']), n(g['opacity']), n(g['scales']), n(g['rotations']),
                                      n(rc['vm'][v]), n(rc['pm'][v]), rc['tfx'][v], rc['tfy'][v], H, W, n(hp.bg))
        tot = w if tot is None else {k: tot[k] + w[k] for k in w}
    print('oracle backward of %d views: %.2f s' % (V, time.perf_counter() - t0))
    for k in ('means3D', 'colors', 'opacities', 'scales', 'rotations'):
        a, b = n(out[k]), tot[k]
        print('  %-10s max|err|/scale %.3e' % (k, np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)))
