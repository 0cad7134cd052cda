"""BEVPoolv2 voxel pooling — the MI355X op behind the reference's ``bev_pool_v2``.

Public surface (identical names / argument order / tensor conventions to
``mmdet3d/ops/bev_pool_v2/bev_pool.py`` so the reference's import line is the only edit):

``bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
interval_lengths)``  -> ``(B, C, Z, Y, X)`` contiguous fp32                     [bev_pool.py:86-92]
    depth ``(B,N,D,H,W)`` any float dtype, feat ``(B,N,H,W,C)`` channels-last, 1-D integer ranks,
    ``bev_feat_shape = (B,Z,Y,X,C)``.  Differentiable w.r.t. depth and feat.  Empty voxels are 0.
``QuickCumsumCuda``   the autograd Function (returns the pre-permute ``(B,Z,Y,X,C)`` tensor)
                                                                                [bev_pool.py:11-83]
``TRTBEVPoolv2``      export shim with the ``mmdeploy::bev_pool_v2`` symbolic   [bev_pool.py:95-142]
``bev_pool_v2_ext``   object with ``bev_pool_v2_forward`` / ``bev_pool_v2_backward`` taking the
                      pybind module's argument order — ``interval_lengths`` BEFORE
                      ``interval_starts``                               [src/bev_pool.cpp:30-39,74-85]

All compute is in ``csrc/bev_pool.hip`` behind the C ABI of ``include/ocrf_hip.h``.  There is no
CPU path: CPU tensors raise ``OcrfHipError``.
"""
import ctypes
import os

import torch

from . import _lib

__all__ = ['bev_pool_v2_device_counts', 'bev_pool_v2', 'bev_pool_v2_collapsed', 'TRTBEVPoolv2', 'QuickCumsumCuda',
           'bev_pool_v2_ext', 'runs_of', 'dense_runs']


def _want(t, dtype, name):
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.OcrfHipError(f'{name}: expected contiguous {dtype}, got {t.dtype} '
                                f'contiguous={t.is_contiguous()}')


def runs_of(sorted_ranks):
    """(starts, lengths) int32 of the runs of equal values in a sorted 1-D tensor — what
    view_transformer.py:245-252 and bev_pool.py:50-57 build with a boolean mask."""
    _, counts = torch.unique_consecutive(sorted_ranks, return_counts=True)
    ends = torch.cumsum(counts, 0)
    return (ends - counts).int(), counts.int()


def dense_runs(sorted_ranks, n_values):
    """(starts, lengths) int32 of the run of EVERY value 0..n_values-1 in a sorted 1-D tensor (length 0 where
    a value is absent).  Same runs as ``runs_of`` plus empty ones, but of a size the host knows: no
    device -> host read (``unique_consecutive`` has to return its count), so the backward of ``bev_pool_v2``
    does not stall the autograd thread.  The gradient kernels skip empty runs."""
    edges = torch.searchsorted(sorted_ranks, torch.arange(n_values + 1, device=sorted_ranks.device,
                                                          dtype=sorted_ranks.dtype))
    return edges[:-1].int(), (edges[1:] - edges[:-1]).int()


class _ExtensionAPI:
    """Call-compatible stand-in for the reference's pybind module (src/bev_pool.cpp:106-111)."""

    @staticmethod
    def bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev,
                            interval_lengths, interval_starts):
        _lib.require_cuda(depth, feat, out, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                          interval_starts)
        for name, t in (('depth', depth), ('feat', feat), ('out', out)):
            _want(t, torch.float32, name)
        for name, t in (('ranks_depth', ranks_depth), ('ranks_feat', ranks_feat),
                        ('ranks_bev', ranks_bev), ('interval_lengths', interval_lengths),
                        ('interval_starts', interval_starts)):
            _want(t, torch.int32, name)
        channels = feat.size(4)                       # `c = _feat.size(4)`, bev_pool.cpp:40
        n_iv = interval_lengths.size(0)
        n_pts = ranks_depth.numel()
        if ranks_feat.numel() != n_pts or ranks_bev.numel() != n_pts or interval_starts.numel() != n_iv:
            raise _lib.OcrfHipError('rank / interval vectors disagree in length')
        L = _lib.lib()
        dev = depth.device
        with _lib.on_device(dev):                  # OptionalCUDAGuard, bev_pool.cpp:42
            stream = _lib.stream_ptr(dev)
            if os.environ.get('OCRF_CHECK_INTERVALS', '0') == '1':
                flag = torch.zeros(1, dtype=torch.int32, device=dev)
                _lib.check(L.ocrf_bev_pool_v2_check_intervals(
                    n_iv, n_pts, _lib.ptr(interval_starts), _lib.ptr(interval_lengths),
                    _lib.ptr(flag), stream), 'ocrf_bev_pool_v2_check_intervals')
                bad = int(flag.item())
                if bad:
                    raise _lib.OcrfHipError(
                        f'intervals are not an ascending non-overlapping cover (flag={bad}); the '
                        'C entry point bev_pool_v2() accepts arbitrary layouts')
            n_vox = out.numel() // max(channels, 1)
            need = L.ocrf_bev_pool_v2_workspace_bytes(channels, n_pts, n_vox)
            scratch = _lib.workspace.get(dev, need, 'bev_pool')
            _lib.check(L.ocrf_bev_pool_v2(
                channels, n_iv, n_pts, n_vox, _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(ranks_depth),
                _lib.ptr(ranks_feat), _lib.ptr(ranks_bev), _lib.ptr(interval_starts),
                _lib.ptr(interval_lengths), _lib.ptr(out), _lib.ptr(scratch),
                ctypes.c_size_t(scratch.numel()), stream), 'ocrf_bev_pool_v2')

    @staticmethod
    def bev_pool_v2_backward(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth,
                             ranks_feat, ranks_bev, interval_lengths, interval_starts):
        _lib.require_cuda(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth, ranks_feat,
                          ranks_bev, interval_lengths, interval_starts)
        for name, t in (('out_grad', out_grad), ('depth_grad', depth_grad),
                        ('feat_grad', feat_grad), ('depth', depth), ('feat', feat)):
            _want(t, torch.float32, name)
        for name, t in (('ranks_depth', ranks_depth), ('ranks_feat', ranks_feat),
                        ('ranks_bev', ranks_bev), ('interval_lengths', interval_lengths),
                        ('interval_starts', interval_starts)):
            _want(t, torch.int32, name)
        channels = out_grad.size(4)                   # `c = _out_grad.size(4)`, bev_pool.cpp:86
        L = _lib.lib()
        dev = out_grad.device
        with _lib.on_device(dev):
            _lib.check(L.ocrf_bev_pool_v2_grad(
                channels, interval_lengths.size(0), _lib.ptr(out_grad), _lib.ptr(depth),
                _lib.ptr(feat), _lib.ptr(ranks_depth), _lib.ptr(ranks_feat), _lib.ptr(ranks_bev),
                _lib.ptr(interval_starts), _lib.ptr(interval_lengths), _lib.ptr(depth_grad),
                _lib.ptr(feat_grad), _lib.stream_ptr(dev)), 'ocrf_bev_pool_v2_grad')


bev_pool_v2_ext = _ExtensionAPI()


class QuickCumsumCuda(torch.autograd.Function):
    """Autograd wrapper with the reference's name and call signature (bev_pool.py:11-83)."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                bev_feat_shape, interval_starts, interval_lengths):
        # dtype/contiguity normalisation as bev_pool.py:19-25
        rb = ranks_bev.int().contiguous()
        rd = ranks_depth.int().contiguous()
        rf = ranks_feat.int().contiguous()
        d32 = depth.float().contiguous()
        f32 = feat.float().contiguous()
        starts = interval_starts.int().contiguous()
        lengths = interval_lengths.int().contiguous()
        pooled = f32.new_zeros(tuple(int(s) for s in bev_feat_shape))    # bev_pool.py:27
        bev_pool_v2_ext.bev_pool_v2_forward(d32, f32, pooled, rd, rf, rb, lengths, starts)
        ctx.save_for_backward(rb, d32, f32, rf, rd)
        return pooled

    @staticmethod
    def backward(ctx, grad_pooled):
        rb, d32, f32, rf, rd = ctx.saved_tensors
        # regroup the point list by feature pixel (bev_pool.py:47-57); stable so that runs of
        # equal ranks_feat keep their forward order
        rf_sorted, perm = torch.sort(rf, stable=True)
        starts_bp, lengths_bp = dense_runs(rf_sorted, f32.numel() // f32.size(-1))
        g_depth = torch.zeros_like(d32)
        g_feat = torch.zeros_like(f32)
        bev_pool_v2_ext.bev_pool_v2_backward(
            grad_pooled.contiguous(), g_depth, g_feat, d32, f32, rd[perm].contiguous(),
            rf_sorted.contiguous(), rb[perm].contiguous(), lengths_bp, starts_bp)
        return g_depth, g_feat, None, None, None, None, None, None


def _fusable(channels):
    return channels % 4 == 0 and 32 <= channels <= 256


class _FusedPool(torch.autograd.Function):
    """bev_pool_v2 + the reference's layout passes in one go (C ABI ``ocrf_bev_pool_v2_nchw``):
    ``layout`` 0 -> (B,C,Z,Y,X) (bev_pool.py:91), 1 -> (B,Z*C,Y,X) (view_transformer.py:194).
    The output is written exactly once (no zero-fill, no permute, no cat)."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                interval_starts, interval_lengths, layout):
        rb = ranks_bev.int().contiguous()
        rd = ranks_depth.int().contiguous()
        rf = ranks_feat.int().contiguous()
        d32 = depth.float().contiguous()
        f32 = feat.float().contiguous()
        starts = interval_starts.int().contiguous()
        lengths = interval_lengths.int().contiguous()
        _lib.require_cuda(d32, f32, rb, rd, rf, starts, lengths)
        B, Z, Y, X, C = (int(v) for v in bev_feat_shape)
        if f32.size(-1) != C:
            raise _lib.OcrfHipError(f'feat has {f32.size(-1)} channels, bev_feat_shape says {C}')
        n_iv, n_pts = starts.numel(), rd.numel()
        if rf.numel() != n_pts or rb.numel() != n_pts or lengths.numel() != n_iv:
            raise _lib.OcrfHipError('rank / interval vectors disagree in length')
        dev = d32.device
        out = torch.empty((B, C, Z, Y, X) if layout == 0 else (B, Z * C, Y, X),
                          dtype=torch.float32, device=dev)
        L = _lib.lib()
        with _lib.on_device(dev):
            need = L.ocrf_bev_pool_v2_nchw_workspace_bytes(C, n_iv, n_pts, B, Z, Y, X)
            scratch = _lib.workspace.get(dev, need, 'bev_pool_nchw')
            _lib.check(L.ocrf_bev_pool_v2_nchw(
                C, n_iv, n_pts, _lib.ptr(d32), _lib.ptr(f32), _lib.ptr(rd), _lib.ptr(rf),
                _lib.ptr(rb), _lib.ptr(starts), _lib.ptr(lengths), _lib.ptr(out), B, Z, Y, X,
                int(layout), _lib.ptr(scratch), ctypes.c_size_t(scratch.numel()),
                _lib.stream_ptr(dev)), 'ocrf_bev_pool_v2_nchw')
        ctx.save_for_backward(rb, d32, f32, rf, rd)
        ctx.geom = (B, Z, Y, X, C, int(layout))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rb, d32, f32, rf, rd = ctx.saved_tensors
        B, Z, Y, X, C, layout = ctx.geom
        if layout == 0:
            g = grad_out.permute(0, 2, 3, 4, 1)                       # (B,Z,Y,X,C)
        else:
            g = grad_out.view(B, Z, C, Y, X).permute(0, 1, 3, 4, 2)
        rf_sorted, perm = torch.sort(rf, stable=True)
        starts_bp, lengths_bp = dense_runs(rf_sorted, f32.numel() // f32.size(-1))
        g_depth = torch.zeros_like(d32)
        g_feat = torch.zeros_like(f32)
        bev_pool_v2_ext.bev_pool_v2_backward(
            g.contiguous().float(), g_depth, g_feat, d32, f32, rd[perm].contiguous(),
            rf_sorted.contiguous(), rb[perm].contiguous(), lengths_bp, starts_bp)
        return g_depth, g_feat, None, None, None, None, None, None, None


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev,
                bev_feat_shape, interval_starts, interval_lengths):
    """The reference's op: -> (B, C, Z, Y, X) contiguous fp32 (bev_pool.py:86-92)."""
    if _fusable(int(bev_feat_shape[-1])):
        return _FusedPool.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                                interval_starts, interval_lengths, 0)
    pooled = QuickCumsumCuda.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev,
                                   bev_feat_shape, interval_starts, interval_lengths)
    return pooled.permute(0, 4, 1, 2, 3).contiguous()        # (B,Z,Y,X,C) -> (B,C,Z,Y,X)


@torch.no_grad()
def bev_pool_v2_device_counts(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                              interval_starts, interval_lengths, counts, layout=1, scratch_tag='bev_pool_nchw'):
    """Forward-only pooling on rank vectors whose lengths live on the device: the five vectors are
    passed at their CAPACITY (as ``index_prep.*_hip(sync=False)`` returns them) together with
    ``counts`` = int32 device tensor [n_points, n_intervals], so that nothing between the index
    preparation and the pooling reads the device.  ``layout`` as in ``_FusedPool``.  ``scratch_tag``: the call's scratch
    (voxel table, unit list, slabs) is one buffer per tag — two poolings that run at the same time on two streams need
    two tags."""
    B, Z, Y, X, C = (int(v) for v in bev_feat_shape)
    if not _fusable(C):
        raise _lib.OcrfHipError(f'bev_pool_v2_device_counts needs C % 4 == 0 and C <= 256, got {C}')
    d32, f32 = depth.float().contiguous(), feat.float().contiguous()
    _lib.require_cuda(d32, f32, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, counts)
    for t in (ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, counts):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise _lib.OcrfHipError('rank / interval / count vectors must be contiguous int32')
    cap_pts, cap_iv = ranks_depth.numel(), interval_starts.numel()
    dev = d32.device
    out = torch.empty((B, C, Z, Y, X) if layout == 0 else (B, Z * C, Y, X), dtype=torch.float32, device=dev)
    L = _lib.lib()
    with _lib.on_device(dev):
        need = L.ocrf_bev_pool_v2_nchw_workspace_bytes(C, cap_iv, cap_pts, B, Z, Y, X)
        scratch = _lib.workspace.get(dev, need, scratch_tag)
        _lib.check(L.ocrf_bev_pool_v2_nchw_dyn(
            C, cap_iv, cap_pts, _lib.ptr(counts), _lib.ptr(d32), _lib.ptr(f32), _lib.ptr(ranks_depth),
            _lib.ptr(ranks_feat), _lib.ptr(ranks_bev), _lib.ptr(interval_starts), _lib.ptr(interval_lengths),
            _lib.ptr(out), B, Z, Y, X, int(layout), _lib.ptr(scratch), ctypes.c_size_t(scratch.numel()),
            _lib.stream_ptr(dev)), 'ocrf_bev_pool_v2_nchw_dyn')
    return out


class _FusedPoolCounts(torch.autograd.Function):
    """``_FusedPool`` on rank vectors whose LENGTHS LIVE ON THE DEVICE (``index_prep.*_hip(sync=False)``: five vectors at
    their capacity + ``counts`` = int32 [n_points, n_intervals]): forward by ``ocrf_bev_pool_v2_nchw_dyn``, backward by the
    same kernel as ``_FusedPool`` — so a training forward reads nothing back from the device.  The backward regroups the
    points by feature pixel like the reference (bev_pool.py:47-57); the entries past ``n_points`` (uninitialised) are given
    a feature index past the last pixel first, so the stable sort puts them behind every run and no run covers them."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts, interval_lengths,
                counts, layout, scratch_tag):
        d32, f32 = depth.float().contiguous(), feat.float().contiguous()
        with torch.no_grad():
            out = bev_pool_v2_device_counts(d32, f32, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                                            interval_lengths, counts, layout=layout, scratch_tag=scratch_tag)
        ctx.save_for_backward(ranks_bev, d32, f32, ranks_feat, ranks_depth, counts)
        ctx.geom = tuple(int(v) for v in bev_feat_shape) + (int(layout),)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rb, d32, f32, rf, rd, counts = ctx.saved_tensors
        B, Z, Y, X, C, layout = ctx.geom
        if layout == 0:
            g = grad_out.permute(0, 2, 3, 4, 1)                       # (B,Z,Y,X,C)
        else:
            g = grad_out.view(B, Z, C, Y, X).permute(0, 1, 3, 4, 2)
        n_pixels = f32.numel() // f32.size(-1)
        live = torch.arange(rf.numel(), device=rf.device, dtype=torch.int32) < counts[0]
        rf_sorted, perm = torch.sort(torch.where(live, rf, torch.full_like(rf, n_pixels)), stable=True)
        starts_bp, lengths_bp = dense_runs(rf_sorted, n_pixels)
        g_depth = torch.zeros_like(d32)
        g_feat = torch.zeros_like(f32)
        bev_pool_v2_ext.bev_pool_v2_backward(
            g.contiguous().float(), g_depth, g_feat, d32, f32, rd[perm].contiguous(),
            rf_sorted.contiguous(), rb[perm].contiguous(), lengths_bp, starts_bp)
        return (g_depth, g_feat) + (None,) * 9


def bev_pool_v2_device_counts_autograd(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                                       interval_lengths, counts, layout=1, scratch_tag='bev_pool_nchw'):
    """``bev_pool_v2_device_counts`` with autograd for ``depth`` and ``feat`` (``_FusedPoolCounts``)."""
    return _FusedPoolCounts.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                                  interval_lengths, counts, int(layout), scratch_tag)


class DevicePoolPlan:
    """The rank-only part of one pooling, built once for rank vectors that are cached across calls
    (C ABI ``ocrf_bev_pool_plan_build``): the dense voxel table, the list of work units (tiles, and slices
    of heavy tiles) and their split over the XCDs.  ``bev_pool_v2_planned`` is then ONE launch that needs
    only depth, feat, ranks_depth and ranks_feat.  Forward only; a plan serves one stream at a time (it
    holds the arrival counters of the cut tiles)."""

    def __init__(self, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts, interval_lengths):
        _lib.require_cuda(ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths)
        B, Z, Y, X, C = (int(v) for v in bev_feat_shape)
        if not _fusable(C):
            raise _lib.OcrfHipError(f'pool plans need C % 4 == 0 and C <= 256, got {C}')
        self.shape = (B, Z, Y, X, C)
        self.ranks_depth = ranks_depth.int().contiguous()
        self.ranks_feat = ranks_feat.int().contiguous()
        rb, st, ln = ranks_bev.int().contiguous(), interval_starts.int().contiguous(), interval_lengths.int().contiguous()
        self.n_points, self.n_intervals = self.ranks_depth.numel(), st.numel()
        dev = rb.device
        L = _lib.lib()
        self.plan = torch.empty(L.ocrf_bev_pool_plan_bytes(C, self.n_points, B, Z, Y, X), dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            _lib.check(L.ocrf_bev_pool_plan_build(C, self.n_intervals, self.n_points, _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln),
                                                  B, Z, Y, X, _lib.ptr(self.plan),
                                                  ctypes.c_size_t(self.plan.numel()), _lib.stream_ptr(dev)),
                       'ocrf_bev_pool_plan_build')
            # the slabs of cut tiles: owned by the plan (its lifetime, not the process-wide grow-only workspace), so a
            # plan can run beside another plan's launch and a dropped plan returns its memory
            self.scratch = torch.empty(max(int(L.ocrf_bev_pool_planned_workspace_bytes(C, self.n_points)), 256),
                                       dtype=torch.uint8, device=dev)


@torch.no_grad()
def bev_pool_v2_planned(depth, feat, plan, layout=1, out=None):
    """``bev_pool_v2_collapsed`` (layout 1: (B, Z*C, Y, X)) or ``bev_pool_v2`` (layout 0: (B,C,Z,Y,X)) for the
    rank vectors a ``DevicePoolPlan`` was built from; bit-identical results.  ``out``: optional contiguous fp32
    tensor of that many elements to write into (e.g. a slice of a fused multi-frame buffer)."""
    B, Z, Y, X, C = plan.shape
    d32, f32 = depth.float().contiguous(), feat.float().contiguous()
    _check_32bit_offsets(d32, f32)
    _lib.require_cuda(d32, f32)
    if f32.size(-1) != C:
        raise _lib.OcrfHipError(f'feat has {f32.size(-1)} channels, the plan was built for {C}')
    dev = d32.device
    shape = (B, C, Z, Y, X) if layout == 0 else (B, Z * C, Y, X)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=dev)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != B * C * Z * Y * X or out.device != dev:
        raise _lib.OcrfHipError('out must be a contiguous fp32 tensor of B*C*Z*Y*X elements on the inputs\' device')
    L = _lib.lib()
    with _lib.on_device(dev):
        scratch = plan.scratch
        _lib.check(L.ocrf_bev_pool_v2_nchw_planned(
            C, plan.n_points, _lib.ptr(d32), _lib.ptr(f32), _lib.ptr(plan.ranks_depth),
            _lib.ptr(plan.ranks_feat), _lib.ptr(plan.plan), _lib.ptr(out), B, Z, Y, X, int(layout), _lib.ptr(scratch),
            ctypes.c_size_t(scratch.numel()), ctypes.c_size_t(d32.numel() * 4), ctypes.c_size_t(f32.numel() * 4),
            _lib.stream_ptr(dev)), 'ocrf_bev_pool_v2_nchw_planned')
    return out


def _check_32bit_offsets(*tensors):
    """The tile / MFMA / panel pooling kernels address their operands with 32-bit byte offsets (raw buffer loads whose
    resource covers 2^31 - 1 bytes: a load beyond returns 0 silently).  A tensor of 2 GiB or more is refused here."""
    for t in tensors:
        if t is not None and t.numel() * t.element_size() >= (1 << 31):
            raise _lib.OcrfHipError(f'a pooling operand of {t.numel() * t.element_size()} bytes exceeds the 2 GiB the '
                                    'kernels\' 32-bit offsets cover: split the batch')


class MfmaPoolPlan:
    """Rank-only part of the MFMA panel pooling (csrc/bev_pool_mfma.hip): per 64-voxel tile its UNIQUE feature rows in
    panels of 64, per panel its non-zero (voxel slot, row slot) cells with their points in summation order, and the
    unit list (a tile's panels in groups of at most ``group``; heaviest units first; tiles of several units reduce
    through slabs).  Built once per calibration from the rank vectors with device-side torch index algebra (sort /
    unique / searchsorted — plan time, not step time); every point of the rank vectors is pooled into
    ``ranks_bev[p]`` (what the reference's intervals amount to when they cover the point list: bev_pool.py:40-57)."""

    def __init__(self, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, group=4, unit_cost=None):
        _lib.require_cuda(ranks_depth, ranks_feat, ranks_bev)
        B, Z, Y, X, C = (int(v) for v in bev_feat_shape)
        if C not in (64, 80, 96, 128):
            raise _lib.OcrfHipError(f'the MFMA pooling takes C in (64, 80, 96, 128), got {C}')
        self.shape = (B, Z, Y, X, C)
        dev = ranks_bev.device
        L = _lib.lib()
        KP = int(L.ocrf_bev_pool_mfma_panel_rows())
        TS = int(L.ocrf_bev_pool_mfma_tile_side())
        YX, TV = Y * X, TS * TS
        tcx, tcy = (X + TS - 1) // TS, (Y + TS - 1) // TS
        tpp = tcx * tcy
        n_tiles = B * Z * tpp
        rb, rf, rd = ranks_bev.long(), ranks_feat.long(), ranks_depth.long()
        keep = (rb >= 0) & (rb < B * Z * YX)
        rb, rf, rd = rb[keep], rf[keep], rd[keep]
        self.n_points = int(rb.numel())
        i32 = lambda t: t.to(torch.int32).contiguous()  # noqa: E731
        if self.n_points:
            plane, inpl = rb // YX, rb % YX
            y, x = inpl // X, inpl % X
            tile, v = plane * tpp + (y // TS) * tcx + x // TS, (y % TS) * TS + x % TS
            n_rows = int(rf.max()) + 1
            key = (tile * n_rows + rf) * TV + v
            order = torch.argsort(key, stable=True)                 # points of a cell stay in list order
            key_s = key[order]
            first = torch.ones_like(key_s, dtype=torch.bool)
            first[1:] = key_s[1:] != key_s[:-1]
            cell_start = torch.nonzero(first).flatten()
            cell_key = key_s[cell_start]
            cell_pair, cell_v = cell_key // TV, cell_key % TV
            pfirst = torch.ones_like(cell_pair, dtype=torch.bool)
            pfirst[1:] = cell_pair[1:] != cell_pair[:-1]
            pair_of_cell = torch.cumsum(pfirst.long(), 0) - 1
            pair_key = cell_pair[pfirst]
            pair_tile, pair_rf = pair_key // n_rows, pair_key % n_rows
            tiles = torch.arange(n_tiles + 1, device=dev)
            tile_pair_start = torch.searchsorted(pair_tile, tiles)                # first pair of every tile
            rslot = torch.arange(pair_key.numel(), device=dev) - tile_pair_start[pair_tile]
            n_pan_tile = (tile_pair_start[1:] - tile_pair_start[:-1] + KP - 1) // KP
        else:
            n_pan_tile = torch.zeros(n_tiles, dtype=torch.long, device=dev)
        tile_panel_off = torch.cumsum(n_pan_tile, 0) - n_pan_tile
        n_panels = int(n_pan_tile.sum())
        if self.n_points:
            pair_panel = tile_panel_off[pair_tile] + rslot // KP
            pair_r = rslot % KP
            panel_rows = torch.zeros(max(n_panels, 1) * KP, dtype=torch.long, device=dev)
            panel_rows[pair_panel * KP + pair_r] = pair_rf
            panel_nrows = torch.bincount(pair_panel, minlength=max(n_panels, 1))
            cell_panel = pair_panel[pair_of_cell]
            panel_cell_off = torch.searchsorted(cell_panel, torch.arange(n_panels + 1, device=dev))
            rd_sorted = rd[order]
            cell_end = torch.cat((cell_start[1:], torch.tensor([self.n_points], device=dev)))
            npts = cell_end - cell_start
            code = cell_v | (pair_r[pair_of_cell] << 8)
            # cells of a panel voxel-major (v, then row slot): bev_pool_panel.hip walks a voxel's cells in row order —
            # the k order of the MFMA form, which itself does not depend on the order inside a panel
            vm = torch.argsort((cell_panel * TV + cell_v) * KP + pair_r[pair_of_cell])
            cell_start, cell_end, npts, code, cell_v = cell_start[vm], cell_end[vm], npts[vm], code[vm], cell_v[vm]
            # voxel slot v of panel p owns the cells [voff[p][v], voff[p][v + 1]) (relative to the panel's first cell)
            panel_voff = torch.searchsorted(cell_panel * TV + cell_v, torch.arange(n_panels * TV, device=dev)) - \
                torch.repeat_interleave(panel_cell_off[:-1], TV)
            cell_code = ((code >> 8) | ((code & 0xFF) << 8)).to(torch.int16).contiguous()      # row slot | voxel slot << 8 (< 2^14)
            last = self.n_points - 1
            inline = npts <= 3
            rd0 = rd_sorted[cell_start]
            rd1 = rd_sorted[torch.clamp(cell_start + 1, max=last)]
            rd2 = rd_sorted[torch.clamp(cell_start + 2, max=last)]
            cells = torch.stack((code | (torch.where(inline, npts, torch.full_like(npts, 0xFFFF)) << 16),
                                 torch.where(inline, rd0, cell_start), torch.where(inline, rd1, npts),
                                 torch.where(inline, rd2, torch.zeros_like(rd2))), 1)
        else:
            panel_rows = torch.zeros(KP, dtype=torch.long, device=dev)
            panel_nrows = torch.zeros(1, dtype=torch.long, device=dev)
            panel_cell_off = torch.zeros(2, dtype=torch.long, device=dev)
            panel_voff = torch.zeros(TV, dtype=torch.long, device=dev)
            cell_code = torch.zeros(1, dtype=torch.int16, device=dev)
            cells = torch.zeros(1, 4, dtype=torch.long, device=dev)
            rd_sorted = torch.zeros(1, dtype=torch.long, device=dev)
        # units: consecutive panels of a tile, at most `group` of them and — with `unit_cost` — about that much estimated
        # time (cost of a panel = 1 + its longest voxel run / 8: what a panel of bev_pool_panel.hip takes, its row fetch
        # plus one trip per cell of the longest run); a tile without points is one unit without panels.  The few
        # tiles beside the rig (hundreds of rows, runs of 30-48 cells) are cut finely, the ordinary ones not at all.
        G = max(1, min(int(group), int(L.ocrf_bev_pool_mfma_max_unit_panels())))
        tiles_i = torch.arange(n_tiles, device=dev)
        if self.n_points and n_panels:
            pan_tile = torch.repeat_interleave(tiles_i, n_pan_tile)
            pan_idx = torch.arange(n_panels, device=dev) - tile_panel_off[pan_tile]          # index of the panel in its tile
            if unit_cost is not None:
                ends = torch.cat((panel_voff.reshape(n_panels, TV), (panel_cell_off[1:] - panel_cell_off[:-1])[:, None]), 1)
                longest = (ends[:, 1:] - ends[:, :-1]).max(1).values
                cost = 1.0 + longest.double() / 8.0
                cum = torch.cumsum(cost, 0)
                excl = cum - cost
                excl = excl - excl[tile_panel_off[pan_tile]]                                   # exclusive prefix inside the tile
                bucket = torch.floor(excl / float(unit_cost)).long()
            else:
                bucket = torch.zeros(n_panels, dtype=torch.long, device=dev)
            # a new unit starts where the bucket changes, and every G panels inside a bucket
            bfirst = torch.ones(n_panels, dtype=torch.bool, device=dev)
            bfirst[1:] = (bucket[1:] != bucket[:-1]) | (pan_tile[1:] != pan_tile[:-1])
            bstart = torch.cummax(torch.where(bfirst, torch.arange(n_panels, device=dev), torch.zeros_like(pan_idx)), 0).values
            ufirst = bfirst | ((torch.arange(n_panels, device=dev) - bstart) % G == 0)
            unit_p0 = torch.nonzero(ufirst).flatten()
            unit_p1 = torch.cat((unit_p0[1:], torch.tensor([n_panels], device=dev)))
            unit_tile_np = pan_tile[unit_p0]
            n_unit_tile = torch.bincount(unit_tile_np, minlength=n_tiles)
            empty = n_unit_tile == 0
            # empty tiles: one unit without panels each
            e_tiles = tiles_i[empty]
            unit_tile = torch.cat((unit_tile_np, e_tiles))
            p0 = torch.cat((unit_p0, torch.zeros_like(e_tiles)))
            p1 = torch.cat((unit_p1, torch.zeros_like(e_tiles)))
            first_unit_of_tile = torch.searchsorted(unit_tile_np, tiles_i)
            slice_ = torch.cat((torch.arange(unit_p0.numel(), device=dev) - first_unit_of_tile[unit_tile_np], torch.zeros_like(e_tiles)))
            n_unit_tile = torch.clamp(n_unit_tile, min=1)
        else:
            n_unit_tile = torch.ones(n_tiles, dtype=torch.long, device=dev)
            unit_tile, p0, p1, slice_ = tiles_i, torch.zeros_like(tiles_i), torch.zeros_like(tiles_i), torch.zeros_like(tiles_i)
        multi = n_unit_tile > 1
        slab_first = torch.cumsum(torch.where(multi, n_unit_tile, torch.zeros_like(n_unit_tile)), 0) - \
            torch.where(multi, n_unit_tile, torch.zeros_like(n_unit_tile))
        self.n_slab_slices = int(torch.where(multi, n_unit_tile, torch.zeros_like(n_unit_tile)).sum())
        units = torch.stack((unit_tile, p0, p1, slice_ | (n_unit_tile[unit_tile] << 16)), 1)
        heavy_first = torch.argsort(p1 - p0, descending=True, stable=True)
        self.units = i32(units[heavy_first])
        self.unit_slab = i32(slab_first[unit_tile][heavy_first])
        self.n_units = int(self.units.size(0))
        self.panel_rows, self.panel_nrows, self.panel_cell_off = i32(panel_rows), i32(panel_nrows), i32(panel_cell_off)
        self.cells, self.rd_sorted = i32(cells), i32(rd_sorted)
        self.panel_voff, self.cell_code = i32(panel_voff), cell_code
        self.cw = torch.zeros(int(cells.size(0)), dtype=torch.float32, device=dev)      # bev_pool_cell_weights' output
        self.n_panels, self.n_cells, self.n_tiles = n_panels, int(cells.size(0)), n_tiles
        self.unique_rows = int(panel_nrows.sum()) if self.n_points else 0
        self.arrive = torch.zeros(n_tiles, dtype=torch.int32, device=dev)
        self.slabs = torch.empty(int(L.ocrf_bev_pool_mfma_slab_bytes(C, self.n_slab_slices)), dtype=torch.uint8, device=dev)


@torch.no_grad()
def bev_pool_v2_mfma(depth, feat, plan, layout=1, out=None):
    """The pooling of ``plan``'s rank vectors on the matrix cores: (B, Z*C, Y, X) (layout 1) or (B,C,Z,Y,X)
    (layout 0); same result as ``bev_pool_v2_planned`` up to the summation order."""
    B, Z, Y, X, C = plan.shape
    d32, f32 = depth.float().contiguous(), feat.float().contiguous()
    _check_32bit_offsets(d32, f32)
    _lib.require_cuda(d32, f32)
    if f32.size(-1) != C:
        raise _lib.OcrfHipError(f'feat has {f32.size(-1)} channels, the plan was built for {C}')
    dev = d32.device
    shape = (B, C, Z, Y, X) if layout == 0 else (B, Z * C, Y, X)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=dev)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != B * C * Z * Y * X or out.device != dev:
        raise _lib.OcrfHipError('out must be a contiguous fp32 tensor of B*C*Z*Y*X elements on the inputs\' device')
    L = _lib.lib()
    with _lib.on_device(dev):
        _lib.check(L.ocrf_bev_pool_v2_nchw_mfma(
            C, plan.n_units, _lib.ptr(plan.units), _lib.ptr(plan.unit_slab), _lib.ptr(plan.panel_rows),
            _lib.ptr(plan.panel_nrows), _lib.ptr(plan.panel_cell_off), _lib.ptr(plan.cells), _lib.ptr(plan.rd_sorted), _lib.ptr(d32), _lib.ptr(f32), _lib.ptr(out), B, Z, Y, X, int(layout),
            _lib.ptr(plan.arrive), _lib.ptr(plan.slabs), _lib.stream_ptr(dev)), 'ocrf_bev_pool_v2_nchw_mfma')
    return out


@torch.no_grad()
def bev_pool_cell_weights(depth, plan, plan2=None):
    """Pre-pass of ``bev_pool_v2_panel``: the summed depth weight of every cell of ``plan`` (and of ``plan2`` — the LSS
    and the height-sampling plan of a step read the same depth tensor) in ONE launch -> ``plan.cw`` (``plan2.cw``)."""
    d32 = depth.float().contiguous()
    _check_32bit_offsets(d32)
    _lib.require_cuda(d32)
    L = _lib.lib()
    n2 = plan2.n_cells if (plan2 is not None and plan2.n_points) else 0
    n1 = plan.n_cells if plan.n_points else 0
    null = ctypes.c_void_p(0)
    with _lib.on_device(d32.device):
        _lib.check(L.ocrf_bev_pool_cell_weights(
            n1, _lib.ptr(plan.cells), _lib.ptr(plan.rd_sorted), _lib.ptr(plan.cw),
            n2, _lib.ptr(plan2.cells) if n2 else null, _lib.ptr(plan2.rd_sorted) if n2 else null,
            _lib.ptr(plan2.cw) if n2 else null, _lib.ptr(d32), _lib.stream_ptr(d32.device)), 'ocrf_bev_pool_cell_weights')
    return d32


@torch.no_grad()
def bev_pool_v2_panel(depth, feat, plan, layout=1, out=None, weights_ready=False):
    """The pooling of ``plan``'s rank vectors cell by cell out of LDS (csrc/bev_pool_panel.hip): same plan and same
    result as ``bev_pool_v2_mfma`` (bit for bit on finite inputs).  ``weights_ready``: ``bev_pool_cell_weights`` has
    already run for this ``depth`` on this stream (one launch for both poolings of a step)."""
    B, Z, Y, X, C = plan.shape
    f32 = feat.float().contiguous()
    _check_32bit_offsets(f32)
    _lib.require_cuda(f32)
    if f32.size(-1) != C:
        raise _lib.OcrfHipError(f'feat has {f32.size(-1)} channels, the plan was built for {C}')
    if not weights_ready:
        bev_pool_cell_weights(depth, plan)
    dev = f32.device
    shape = (B, C, Z, Y, X) if layout == 0 else (B, Z * C, Y, X)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=dev)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != B * C * Z * Y * X or out.device != dev:
        raise _lib.OcrfHipError('out must be a contiguous fp32 tensor of B*C*Z*Y*X elements on the inputs\' device')
    L = _lib.lib()
    with _lib.on_device(dev):
        _lib.check(L.ocrf_bev_pool_v2_nchw_panel(
            C, plan.n_units, _lib.ptr(plan.units), _lib.ptr(plan.unit_slab), _lib.ptr(plan.panel_rows),
            _lib.ptr(plan.panel_nrows), _lib.ptr(plan.panel_cell_off), _lib.ptr(plan.panel_voff), _lib.ptr(plan.cell_code),
            _lib.ptr(plan.cw), _lib.ptr(f32), _lib.ptr(out), B, Z, Y, X, int(layout),
            _lib.ptr(plan.arrive), _lib.ptr(plan.slabs), ctypes.c_size_t(f32.numel() * 4), _lib.stream_ptr(dev)),
            'ocrf_bev_pool_v2_nchw_panel')
    return out


def bev_pool_v2_collapsed(depth, feat, ranks_depth, ranks_feat, ranks_bev,
                          bev_feat_shape, interval_starts, interval_lengths):
    """``torch.cat(bev_pool_v2(...).unbind(dim=2), 1)`` -> (B, Z*C, Y, X), the tensor
    ``voxel_pooling_v2`` / ``fast_sampling`` hand on (view_transformer.py:190-195,
    view_transformer_ocrf.py:776-782), produced without the intermediate copies."""
    if _fusable(int(bev_feat_shape[-1])):
        return _FusedPool.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                                interval_starts, interval_lengths, 1)
    x = bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                    interval_starts, interval_lengths)
    return torch.cat(x.unbind(dim=2), 1)


class TRTBEVPoolv2(torch.autograd.Function):
    """Export shim (bev_pool.py:95-142): depth (N,D,H,W), feat (N,H,W,C) -> (1,Y,X,C)."""

    @staticmethod
    def symbolic(g, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                 interval_lengths, out_height=128, out_width=128):
        return g.op('mmdeploy::bev_pool_v2', depth, feat, ranks_depth, ranks_feat, ranks_bev,
                    interval_starts, interval_lengths,
                    out_height_i=out_height, out_width_i=out_width)

    @staticmethod
    def forward(g, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                interval_lengths, out_height=128, out_width=128):
        shape = (1, 1, out_height, out_width, feat.shape[-1])                  # (B,Z,Y,X,C)
        bev = bev_pool_v2(depth[None], feat[None], ranks_depth, ranks_feat, ranks_bev, shape,
                          interval_starts, interval_lengths)
        return bev.squeeze(2).permute(0, 2, 3, 1)
