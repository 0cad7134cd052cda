"""ocrfdet_amd — MI355X (gfx950) implementation of OcRFDet's OcRF-render + BEV-pool + HOA hot path.

Drop-in mirrors of the reference's Python surface for that path (and nothing else):

  ==============================================  =================================================
  reference import                                here
  ==============================================  =================================================
  mmdet3d.ops.bev_pool_v2.bev_pool                ocrfdet_amd.bevpool
  diff_gaussian_rasterization (w-depth fork)      ocrfdet_amd.diff_gaussian_rasterization
  ...MVSGaussian.lib.gaussian_renderer.render     ocrfdet_amd.gaussian_renderer.render
  HOA blocks of view_transformer_ocrf.py          ocrfdet_amd.hoa
  index preparation of view_transformer(_ocrf)    ocrfdet_amd.index_prep
  ...necks.view_transformer_ocrf                  ocrfdet_amd.view_transformer_ocrf (OcRFViewTransformerFull
                                                  and its sub-modules; kernels: ocrfdet_amd.neck_ops)
  ==============================================  =================================================

Every op calls the C ABI of ``csrc/libocrf_hip.so`` (``include/ocrf_hip.h``) through ctypes with
raw device pointers; PyTorch is only the allocator / stream / ``torch.distributed`` provider.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
