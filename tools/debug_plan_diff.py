"""Diagnostic: planned vs per-call render of a tiny scene; prints where they differ."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from ocrfdet_amd import diff_gaussian_rasterization as dgr, raster_plan as rp
from tests import helpers

cuda = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(1)
W, H = 64, 32
xyz, rgb, opac, sc, rot = helpers.random_gaussians(rng, n, xy_extent=2.0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
xyz, rgb, opac, sc, rot = map(t, (xyz, rgb, opac, sc, rot))
view, full, tfx, tfy = helpers.simple_camera(W, H)
cams = dgr.pack_cameras(t(view).view(1, 4, 4), t(full).view(1, 4, 4), [tfx], [tfy], H, W, cuda)
bg = torch.zeros(3, device=cuda)
want = dgr.rasterize_views(xyz, rgb, opac, sc, rot, None, None, None, None, H, W, bg, packed_cameras=cams, want_n_contrib=True)
plan = rp.RasterPlan(xyz, cams, H, W, scales=sc, rotations=rot)
from ocrfdet_amd import _lib
for radii, variant in ((True, 0), (False, 0)):
    _lib.lib().ocrf_tune_set(14, variant)
    got = plan.render(rgb, opac, sc, rot, bg, want_radii=radii)
    torch.cuda.synchronize()
    d = (got['color'] - want['color']).abs().amax(1)[0].cpu().numpy()
    dt = (got['final_T'] - want['final_T']).abs()[0].cpu().numpy()
    print('radii', radii, 'variant', variant, 'max dcolor', d.max(), 'max dT', dt.max(), 'n diff', int((d > 0).sum()), 'of', d.size)
    print('rows', (d > 0).sum(1).tolist()); print('cols', (d > 0).sum(0).tolist())
    ys, xs = np.nonzero(d > 0)
    if len(ys):
        y, x = ys[0], xs[0]
        print('first', y, x, 'want', want['color'][0, :, y, x].tolist(), want['final_T'][0, y, x].item(), 'got',
              got['color'][0, :, y, x].tolist(), got['final_T'][0, y, x].item(), 'n_contrib', want['n_contrib'][0, y, x].item())
print('opac', opac.flatten().tolist())
