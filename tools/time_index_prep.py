"""Diagnostic: host time and device time of the per-step index preparation at the bench workload."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hotpath, synthetic, index_prep
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_6cam_2frame_bev200x200_render_hoa'
cfg = synthetic.CONFIGS[name]
cfg = synthetic.PathConfig(**{**cfg.__dict__, 'render': False, 'hoa': False})
dev = torch.device('cuda:0')
hp = hotpath.HotPath(cfg, dev, index_prep_mode='per_step')
for _ in range(5): hp.prepare_indices_hip(sync=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): hp.prepare_indices_hip(sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host issue time per call %.1f us; wall incl. device drain %.1f us' % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
args = hp._calib_host
t0 = time.perf_counter()
for _ in range(50):
    a = index_prep.lss_camera_block(*args)
    l, g, _, _ = index_prep.get_projection(*args)
    b = index_prep.ht_camera_block(l, g)
t1 = time.perf_counter()
print('host 3x3 algebra per call %.1f us' % ((t1 - t0) / 50 * 1e6))
t0 = time.perf_counter()
for _ in range(50):
    a.to(dev, non_blocking=True); b.to(dev, non_blocking=True)
t1 = time.perf_counter()
print('two uploads per call %.1f us' % ((t1 - t0) / 50 * 1e6))
(lv, lc), (hv, hc) = hp.prepare_indices_hip(sync=False)
torch.cuda.synchronize()
print('counts lss', lc.tolist(), 'capacity', lv[0].numel(), lv[3].numel(), '| ht', hc.tolist(), 'capacity', hv[0].numel(), hv[3].numel())
for kid, nm in ((_lib.K_LSS_KEYS, 'lss_keys'), (_lib.K_RADIX_HIST, 'radix_hist'), (_lib.K_RADIX_SCATTER, 'radix_scatter'), (_lib.K_SCAN, 'scan_apply'),
                (_lib.K_LSS_BOUNDS, 'lss_bounds'), (_lib.K_HT_COUNT, 'ht_count'), (_lib.K_HT_EMIT, 'ht_emit')):
    t = _lib.KernelTimer(kid, 100); torch.cuda.synchronize(); t.arm()
    for _ in range(10): hp.prepare_indices_hip(sync=False)
    torch.cuda.synchronize(); t.disarm(); ms = t.read_ms(); t.close()
    print('  %-14s median %.1f us x %d per call' % (nm, 1e3 * statistics.median(ms), len(ms) // 10))
depth, feat = hp.make_inputs()
for mode in ('per_step', 'cached'):
    hp.index_prep_mode = mode
    for _ in range(5): hp.step(depth, feat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): hp.step(depth, feat)
    torch.cuda.synchronize(); print('pools-only step, %s: %.1f us' % (mode, (time.perf_counter() - t0) / 50 * 1e6))
