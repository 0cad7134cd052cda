import sys, torch
sys.path.insert(0, '.')
from ocrfdet_amd import hotpath, synthetic
cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
dev = torch.device('cuda:0')
for kw in ({}, dict(lss_pool_backend='panel', ht_pool_backend='panel')):
    hp = hotpath.HotPath(cfg, dev, **kw)
    d, f = hp.make_inputs()
    hp.step(d, f); hp.step(d, f)
    print(kw, 'compiled', hp._compiled is not None, getattr(hp, 'one_call_refused', None), hp._compiled[1].n_calls if hp._compiled else None)
