"""Diagnostic: the cfg2 step on the three synthetic Gaussian sets (bench.gaussian_set_figures), one JSON object per set.
    python tools/gauss_sets.py [--sets init,stress,objects] [--steps 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--sets', default='init,stress,objects')
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--config', default=bench.DEFAULT_CONFIG)
ap.add_argument('--bins', default='4x2', help="candidate lists of the render plans: WxH tile pairs, or 'none'")
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS[a.config]
hp0 = hotpath.HotPath(synthetic.PathConfig(**{**cfg.__dict__, 'render': False, 'hoa': False}), dev)
depth, feat = hp0.make_inputs(seed=0)
del hp0
for kind in a.sets.split(','):
    bins = None if a.bins == 'none' else tuple(int(x) for x in a.bins.split('x'))
    r = bench.gaussian_set_figures(cfg, dev, kind, depth, feat, steps=a.steps, plan_bins=bins)
    print(kind, json.dumps(r), flush=True)
