"""Diagnostic: one rank's ShardedHotPath step (world 1: no collective traffic) with its compute segments replayed by one host
call each against call by call — step time and the host time inside step().    python tools/time_sharded_host.py [config]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402
dev = torch.device('cuda:0')
cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg2_6cam_2frame_bev200x200_render_hoa']
for one in (True, False, True, False):
    sp = hotpath.ShardedHotPath(cfg, dev, 0, 1, one_call=one)
    ins = sp.make_inputs(seed=0)
    for _ in range(10):
        sp.step(ins)
    torch.cuda.synchronize()
    blocks, host = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            sp.step(ins)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / 100 * 1e3)
        host.append((t1 - t0) / 100 * 1e3)
    print('one_call=%-5s  step %.4f ms  host inside step() %.4f ms  segments %d' % (one, np.median(blocks), np.median(host), len(sp._segments)), flush=True)
    del sp
