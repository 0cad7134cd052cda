"""Diagnostic: SHA-1 of the rendered outputs of the bench scene (cfg2 frame 0, 6 views) — to check that a
kernel change that is meant to be bit-neutral is.   python tools/render_hash.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402

cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
hp = hotpath.HotPath(cfg, torch.device('cuda:0'))
for want in (True, False):              # the training-forward variant (tracks n_contrib) and the inference variant
    out = hp.render(want_n_contrib=want)[0]
    torch.cuda.synchronize()
    for k in ('color', 'depth', 'final_T', 'n_contrib', 'radii'):
        if k in out and out[k] is not None:
            print('n_contrib' if want else 'inference', k, hashlib.sha1(out[k].cpu().numpy().tobytes()).hexdigest())
