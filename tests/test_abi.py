"""CPU: the C-ABI library builds/loads without a GPU and exports every symbol that
include/ocrf_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

from ocrfdet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'ocrf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b((?:ocrf_|bev_pool_v2)\w*)\s*\(', text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert 'bev_pool_v2' in syms and 'bev_pool_v2_grad' in syms and 'ocrf_bev_pool_v2' in syms


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f'libocrf_hip.so lacks {missing}'
    assert b'gfx950' in lib.ocrf_version()


def test_code_object_is_gfx950_only():
    so = os.path.join(ROOT, 'ocrfdet_amd', 'csrc', 'libocrf_hip.so')
    blob = open(so, 'rb').read()
    import re
    targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-f]+)', blob))   # offload bundle entry ids
    assert targets == {b'gfx950'}, targets


def test_workspace_query_is_host_only():
    lib = _lib.lib()
    n_vox = 128 * 128
    n = lib.ocrf_bev_pool_v2_workspace_bytes(80, 447232, n_vox)
    tiles = n_vox // 64                             # 64 voxels per tile; 12 lane groups x 32 points per round at C=80
    # dense voxel table + per-tile counters / info + a unit per tile and per slice of a heavy tile
    assert n >= n_vox * 8 + tiles * 16 + (tiles + 447232 // 1536) * 16
    assert lib.ocrf_bev_pool_v2_workspace_bytes(3, 1000, 100) == 0     # scalar path needs none
    assert lib.ocrf_bev_pool_v2_nchw_workspace_bytes(80, 8774, 447232, 1, 1, 128, 128) >= n - 4096
    assert lib.ocrf_bev_pool_plan_bytes(80, 447232, 1, 1, 128, 128) > 0
    assert lib.ocrf_bev_pool_planned_workspace_bytes(80, 447232) > 0


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from ocrfdet_amd import bevpool
    d = torch.zeros(1, 1, 2, 2, 2)
    f = torch.zeros(1, 1, 2, 2, 4)
    r = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(_lib.OcrfHipError):
        bevpool.bev_pool_v2(d, f, r, r, r, (1, 1, 2, 2, 4), r[:1], r[:1])
