"""Diagnostic: HOA-3 mask-gate launch shape (ocrf_tune_set 20 = channel groups, 21 = threads per workgroup) at cfg2.
    python tools/sweep_mask_gate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import _lib, hoa  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
m = hoa.ObatinOpacityMask().to(dev).eval()
x = torch.randn(2, 80, 200, 200, device=dev)
ob = torch.randn(2, 1, 200, 200, device=dev)
L = _lib.lib()
ref = None
for threads in (256, 128):
    for groups in (0, 2, 4, 5, 8, 10, 16):
        _lib.check(L.ocrf_tune_set(21, threads), 'tune')
        _lib.check(L.ocrf_tune_set(22, threads), 'tune')
        _lib.check(L.ocrf_tune_set(20, groups), 'tune')
        with torch.no_grad():
            for _ in range(5):
                mask, gated = m.gate(x, ob)
            torch.cuda.synchronize()
            if ref is None:
                ref = (mask.clone(), gated.clone())
            assert torch.equal(mask, ref[0]) and torch.equal(gated, ref[1])
            res = {}
            for kid in (_lib.K_HOA_MASK_GATE, _lib.K_HOA_STATS):
                t = _lib.KernelTimer(kid, 64)
                t.arm()
                for _ in range(30):
                    m.gate(x, ob)
                torch.cuda.synchronize()
                t.disarm()
                ms = sorted(t.read_ms())
                t.close()
                res[kid] = 1e3 * ms[len(ms) // 2]
        print('threads %3d groups %2d: mask_gate %5.1f us (stats %4.1f us)' % (threads, groups, res[_lib.K_HOA_MASK_GATE], res[_lib.K_HOA_STATS]), flush=True)
