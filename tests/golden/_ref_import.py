"""Imports the Python side of the reference hot path (read-only tree at /root/reference) in a
container that has none of its heavy dependencies (mmcv, mmdet, open3d, CUDA extensions ...).

Used ONLY by ``tests/golden/make_golden.py`` to generate the committed golden vectors; nothing in
``tests/`` proper, ``bench.py`` or ``ocrfdet_amd/`` imports this (the reference tree does not exist
on the GPU box).  No reference source is copied: the modules are imported from where they lie.

What is stubbed (module shells only — none of the arithmetic under test):
  mmcv.cnn.build_conv_layer, mmcv.runner.{BaseModule, force_fp32}, mmdet LearnedPositionalEncoding /
  BasicBlock, the NECKS registry, pyquaternion, nuscenes, open3d, cv2, imgaug, plyfile,
  torchvision, matplotlib, diff_gaussian_rasterization, and the CUDA extension behind bev_pool_v2
  (replaced by the index_add identity that follows from bev_pool_cuda.cu:39-47).
"""
import importlib
import sys
import types

import torch
import torch.nn as nn

REF = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


class LearnedPositionalEncoding(nn.Module):
    """mmdet 2.x behaviour (mmdet is not part of the reference tree)."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50):
        super().__init__()
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)

    def forward(self, mask):
        h, w = mask.shape[-2:]
        x = torch.arange(w, device=mask.device)
        y = torch.arange(h, device=mask.device)
        x_embed = self.col_embed(x)
        y_embed = self.row_embed(y)
        pos = torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1), y_embed.unsqueeze(1).repeat(1, w, 1)),
                        dim=-1).permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)
        return pos


def bev_pool_v2_index_add(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                          interval_starts, interval_lengths):
    """CPU stand-in for the CUDA extension: out[rb] += depth[rd] * feat[rf] then the wrapper's
    permute (bev_pool.py:86-92)."""
    B, Z, Y, X, C = [int(v) for v in bev_feat_shape]
    d = depth.contiguous().float().reshape(-1)[ranks_depth.long()]
    f = feat.contiguous().float().reshape(-1, C)[ranks_feat.long()]
    out = torch.zeros(B * Z * Y * X, C, dtype=torch.float32)
    out.index_add_(0, ranks_bev.long(), d[:, None] * f)
    return out.view(B, Z, Y, X, C).permute(0, 4, 1, 2, 3).contiguous()


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    identity = lambda *a, **k: (lambda f: f)   # noqa: E731
    _mod('mmcv')
    _mod('mmcv.cnn', build_conv_layer=lambda cfg, *a, **k: nn.Conv2d(*a, **k))
    _mod('mmcv.runner', BaseModule=nn.Module, force_fp32=identity)
    _mod('mmdet')
    _mod('mmdet.models')
    _mod('mmdet.models.utils', LearnedPositionalEncoding=LearnedPositionalEncoding)
    _mod('mmdet.models.backbones')
    _mod('mmdet.models.backbones.resnet', BasicBlock=nn.Identity)
    _mod('pyquaternion', Quaternion=object)
    _mod('nuscenes')
    _mod('nuscenes.utils')
    _mod('nuscenes.utils.geometry_utils', transform_matrix=None)
    _mod('open3d')
    _mod('cv2')
    _mod('imgaug', augmenters=types.ModuleType('augmenters'))
    sys.modules['imgaug.augmenters'] = sys.modules['imgaug'].augmenters
    _mod('plyfile', PlyData=object)
    _mod('torchvision')
    if 'matplotlib' not in sys.modules:
        try:
            import matplotlib  # noqa: F401
            import matplotlib.pyplot  # noqa: F401
        except Exception:
            _mod('matplotlib')
            _mod('matplotlib.pyplot')
    _mod('diff_gaussian_rasterization', GaussianRasterizationSettings=None, GaussianRasterizer=None)

    _pkg('mmdet3d', REF + '/mmdet3d')
    _pkg('mmdet3d.models', REF + '/mmdet3d/models')
    _pkg('mmdet3d.models.necks', REF + '/mmdet3d/models/necks')
    _pkg('mmdet3d.ops', REF + '/mmdet3d/ops')
    _pkg('mmdet3d.ops.bev_pool_v2', REF + '/mmdet3d/ops/bev_pool_v2')
    _mod('mmdet3d.models.builder', NECKS=_Registry())
    _mod('mmdet3d.ops.bev_pool_v2.bev_pool', bev_pool_v2=bev_pool_v2_index_add)

    # run anything that says .cuda() on the CPU
    torch.Tensor.cuda = lambda self, *a, **k: self

    vt = importlib.import_module('mmdet3d.models.necks.view_transformer')
    vto = importlib.import_module('mmdet3d.models.necks.view_transformer_ocrf')
    ca = importlib.import_module('mmdet3d.ops.cross_attention_2d')
    du = importlib.import_module('mmdet3d.models.necks.MVSGaussian.lib.utils.data_utils')
    return vt, vto, ca, du
