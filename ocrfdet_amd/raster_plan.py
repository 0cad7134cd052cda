"""Static render plans: the rasteriser's front end computed once per (Gaussian means, cameras).

In OcRFDet the Gaussian means are the fixed voxel grid (``view_transformer_ocrf.py:651-673,690-692``) and the
cameras are fixed per calibration; only the S/R/A/C heads' outputs change per step (``:1130-1133``).  A
``RasterPlan`` keeps, per camera, the Gaussians that can ever be visible, already in the reference's blend order
(depth bits, then id — ``rasterizer_impl.cu:226-267``), so a render is two launches
(``csrc/raster_plan.hip``; C ABI ``ocrf_raster_plan_*`` / ``ocrf_rasterize_planned``).  Colour, depth and
``final_T`` are bit-identical to ``rasterize_views`` (``tests/test_raster_plan_gpu.py``).

The static cull is valid for Gaussians whose world-space extent ``scale_modifier * max|s| * |R(q)|_2`` stays within
``extent_bound``.  ``guard='device'`` arms the per-call pipeline behind the planned one on the GPU (exact whatever
the parameters do, graph-capturable); ``guard='host'`` only raises status bit 4, which ``check()`` turns into an
exception — the caller decides when to pay that synchronisation.  A plan is also keyed by its cameras on the device:
a call that passes the cameras it means (``render(cameras=...)``) is checked against the plan's there.
"""
import ctypes

import torch

from . import _lib
from .diff_gaussian_rasterization import _f32c

__all__ = ['RasterPlan', 'extent_of']


def extent_of(scales, rotations, scale_modifier=1.0):
    """max over the Gaussians of ``scale_modifier * max_k |s_k| * (|1 - |q|^2| + |q|^2)`` — what a plan's
    ``extent_bound`` must dominate (device scalar tensor)."""
    qq = (rotations.float() ** 2).sum(-1)
    return (float(scale_modifier) * scales.float().abs().amax(-1) * ((1.0 - qq).abs() + qq)).max()


class RasterPlan:
    """Plan for ``means3D`` (P,3) seen by the ``V`` cameras of ``packed_cameras`` (V,36; ``pack_cameras``).

    ``extent_bound``: a float, or None to take ``margin`` x the extent of the example ``scales`` / ``rotations``.
    ``capacity``: records (kept (Gaussian, view) pairs) the plan's buffers hold; None = one synchronising count of
    what these cameras keep, times ``headroom`` (poses of later ``rebuild`` calls keep a few per cent more or fewer).
    The build itself (``rebuild``) never reads anything back: ~ 20 launches, hipGraph-capturable — the reference
    recomputes the render cameras from the dataloader's ``c2w`` for EVERY sample
    (``view_transformer_ocrf.py:1140-1152``), so a plan per sample is a supported mode, not only a plan per
    calibration.
    ``bins``: ``(bin_w, bin_h)`` — candidate lists per bin of ``bin_w`` x ``bin_h`` tile PAIRS (default 4 x 2 = 64 x 64 px):
    the static part of the reference's per-tile lists (``rasterizer_impl.cu:70-138``), so that a tile pair tests the rects
    of its own candidates instead of the view's whole list; ``None``: no lists (every ``rebuild`` is then shorter — what a
    plan rebuilt per SAMPLE wants).  ``cand_capacity``: candidates the lists hold; None = one more synchronising count,
    times ``headroom``.  Lists that do not fit after a ``rebuild`` are ignored by the renders (slower, never wrong)."""

    def __init__(self, means3D, packed_cameras, image_height, image_width, extent_bound=None, scales=None,
                 rotations=None, scale_modifier=1.0, margin=2.0, capacity=None, headroom=1.25, bins=(4, 2),
                 cand_capacity=None):
        _lib.require_cuda(means3D, packed_cameras)
        if means3D.dim() != 2 or means3D.size(1) != 3:
            raise RuntimeError('means3D must have dimensions (num_points, 3)')
        self.device = dev = means3D.device
        self.means3D = _f32c(means3D)
        self.cameras = _f32c(packed_cameras).reshape(-1, 36).clone()      # owned: ``rebuild`` overwrites it in place
        self.P, self.V = int(self.means3D.size(0)), int(self.cameras.size(0))
        self.H, self.W = int(image_height), int(image_width)
        if self.P == 0 or self.V == 0 or self.V > 32:
            raise _lib.OcrfHipError('a render plan needs at least one Gaussian and 1..32 cameras')
        if extent_bound is None:
            if scales is None or rotations is None:
                raise _lib.OcrfHipError('RasterPlan needs extent_bound or example scales / rotations')
            extent_bound = float(margin) * float(extent_of(scales, rotations, scale_modifier))
        self.extent_bound = float(extent_bound)
        L = _lib.lib()
        self.kept = None                       # per-view counts of the sizing pass (None when a capacity was given)
        if capacity is None:
            self.kept = self.count()                               # the one synchronisation
            capacity = int(sum(self.kept) * float(headroom)) + 1024
        self.capacity = max(1, min(int(capacity), self.P * self.V))
        self.total_kept = self.capacity        # (name kept for the tools: strides and scratch sizes follow the capacity)
        with _lib.on_device(dev):
            self._build_ws = torch.empty(int(L.ocrf_raster_plan_build_workspace_bytes(self.P, self.V, self.capacity)),
                                         dtype=torch.uint8, device=dev)
            self.plan = torch.empty(max(int(L.ocrf_raster_plan_bytes(self.P, self.V, self.capacity)), 256),
                                    dtype=torch.uint8, device=dev)
        # sticky status word (bit 4: extent bound exceeded in some call, bit 8: bad view index / unusable plan,
        # bit 16: a call's cameras were not the plan's)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dyn = None                       # per-call scratch, owned by the plan (one stream at a time)
        self._chain_ws = None
        self._host_guarded = False             # a render with guard='host' happened since the last check()
        self.bins = None if bins is None else (int(bins[0]), int(bins[1]))
        # one pinned host int the device writes ("some view is deep") and ocrf_rasterize_planned reads on the host — no copy,
        # no wait — to choose the build of the blend (include/ocrf_hip.h)
        self._hint = torch.zeros(1, dtype=torch.int32).pin_memory() if self.bins is not None else None
        self._bins_buf = self._bins_ws = None
        self.cand_capacity = 0
        self.rebuild()
        if self.bins is not None:
            bw, bh = self.bins
            with _lib.on_device(dev):
                need = int(L.ocrf_raster_plan_bins_workspace_bytes(self.P, self.V, self.H, self.W, bw, bh,
                                                                   ctypes.c_long(self.capacity)))
                if need == 0:
                    raise _lib.OcrfHipError(f'RasterPlan: bins {self.bins} do not fit this image (at most 255 bins per axis)')
                self._bins_ws = torch.empty(need, dtype=torch.uint8, device=dev)
                if cand_capacity is None:
                    total = torch.zeros(1, dtype=torch.int32, device=dev)
                    self._build_bins(sizing=total)
                    self.candidates = int(total.item())           # the second synchronisation
                    cand_capacity = int(self.candidates * float(headroom)) + 4096
                self.cand_capacity = max(1, int(cand_capacity))
                self._bins_buf = torch.zeros(int(L.ocrf_raster_plan_bins_bytes(self.V, self.H, self.W, bw, bh,
                                                                               ctypes.c_long(self.cand_capacity))),
                                             dtype=torch.uint8, device=dev)
            self._build_bins()
            self._record_built()

    def count(self, packed_cameras=None):
        """Synchronising: kept (Gaussian, view) records per view for ``packed_cameras`` (default: the plan's)."""
        L = _lib.lib()
        cams = self.cameras if packed_cameras is None else _f32c(packed_cameras).reshape(-1, 36)
        with _lib.on_device(self.device):
            ws = torch.empty(int(L.ocrf_raster_plan_count_workspace_bytes(self.P)), dtype=torch.uint8, device=self.device)
            mask = torch.empty(self.P, dtype=torch.int32, device=self.device)
            total = torch.empty(1, dtype=torch.int32, device=self.device)
            _lib.check(L.ocrf_raster_plan_count(self.P, self.V, self.H, self.W, _lib.ptr(self.means3D), _lib.ptr(cams),
                                                ctypes.c_float(self.extent_bound), _lib.ptr(mask), _lib.ptr(total),
                                                _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_ptr(self.device)),
                       'ocrf_raster_plan_count')
        per_view = torch.stack([((mask >> v) & 1).sum() for v in range(self.V)]).cpu()
        kept = [int(v) for v in per_view]
        assert sum(kept) == int(total.item())
        return kept

    def _build_bins(self, sizing=None):
        L = _lib.lib()
        bw, bh = self.bins
        with _lib.on_device(self.device):
            _lib.check(L.ocrf_raster_plan_bins_build(
                _lib.ptr(self.plan), ctypes.c_size_t(self.plan.numel()), self.P, self.V, ctypes.c_long(self.capacity),
                self.H, self.W, ctypes.c_float(self.extent_bound), bw, bh,
                ctypes.c_long(0 if sizing is not None else self.cand_capacity),
                _lib.ptr(None if sizing is not None else self._bins_buf),
                ctypes.c_size_t(0 if sizing is not None else self._bins_buf.numel()), _lib.ptr(sizing),
                _lib.ptr(self._bins_ws), ctypes.c_size_t(self._bins_ws.numel()), _lib.stream_ptr(self.device)),
                'ocrf_raster_plan_bins_build')

    def _record_built(self):
        # renders issued on ANOTHER stream must not overtake the build (a half-built plan holds wild record indices)
        self._built = (torch.cuda.Event(), torch.cuda.current_stream(self.device))
        self._built[0].record(self._built[1])
        self._ordered_after_build = set()        # streams that already waited for this build (one wait per stream is enough)

    @torch.no_grad()
    def rebuild(self, packed_cameras=None, means3D=None):
        """(Re)build the plan on the current stream for new cameras (same count) and / or means — no host read, no
        allocation, kernels only.  A plan that does not fit the capacity is marked unusable on the device: renders of
        it raise status bit 8 (``check()``), with ``guard='device'`` they are taken over by the per-call pipeline."""
        if packed_cameras is not None:
            cams = _f32c(packed_cameras).reshape(-1, 36)
            if cams.shape != self.cameras.shape:
                raise _lib.OcrfHipError('rebuild: the camera count of a plan is fixed')
            if cams.data_ptr() != self.cameras.data_ptr():       # (a caller may write new poses into plan.cameras itself)
                self.cameras.copy_(cams, non_blocking=True)
        if means3D is not None:
            if means3D.shape != self.means3D.shape:
                raise _lib.OcrfHipError('rebuild: the Gaussian count of a plan is fixed')
            self.means3D.copy_(_f32c(means3D), non_blocking=True)
        L = _lib.lib()
        with _lib.on_device(self.device):
            _lib.check(L.ocrf_raster_plan_build(
                self.P, self.V, self.H, self.W, _lib.ptr(self.means3D), _lib.ptr(self.cameras),
                ctypes.c_float(self.extent_bound), ctypes.c_long(self.capacity), _lib.ptr(self._build_ws),
                ctypes.c_size_t(self._build_ws.numel()), _lib.ptr(self.plan), ctypes.c_size_t(self.plan.numel()),
                _lib.stream_ptr(self.device)), 'ocrf_raster_plan_build')
        if self._bins_buf is not None:
            self._build_bins()                   # the candidate lists belong to THESE lists (stale ones would be wrong)
        self._record_built()
        return self

    def _scratch(self, n_sets):
        L = _lib.lib()
        if self._bins_buf is not None:
            need = L.ocrf_rasterize_planned_bins_workspace_bytes(ctypes.c_long(self.capacity), n_sets, self.V, self.H, self.W,
                                                                 self.bins[0], self.bins[1], ctypes.c_long(self.cand_capacity))
        else:
            need = L.ocrf_rasterize_planned_workspace_bytes(ctypes.c_long(self.capacity), n_sets)
        if self._dyn is None or self._dyn.numel() < need:
            # zero-filled: the guard flag lives in it between calls (include/ocrf_hip.h, ocrf_rasterize_planned)
            self._dyn = torch.zeros(max(int(need), 256), dtype=torch.uint8, device=self.device)
        return self._dyn

    @torch.no_grad()
    def render(self, colors, opacities, scales, rotations, bg, scale_modifier=1.0, depth_mode='median',
               item_view=None, want_radii=None, guard='host', out=None, blend_workgroups=0, phase='both',
               yield_if=None, cameras=None, views_disjoint=False):
        """Render ``n_items = len(item_view)`` views: item z = plan view ``item_view[z]`` (int32 device tensor) with
        Gaussian set ``z // (n_items // S)`` of the ``(S, P, .)`` (or ``(P, .)``) parameter tensors; without
        ``item_view`` every set renders all ``V`` plan views in order.
        -> dict ``color`` (n_items,3,H,W), ``depth`` (n_items,1,H,W), ``final_T`` (n_items,H,W) [, ``radii``
        (n_items,P)], ``status`` (the plan's sticky device word).  ``blend_workgroups``: size of the blend's persistent
        grid; 0 = what the device holds at once (fastest alone), ~2 per CU when other streams should run beside it.
        ``yield_if``: int32 device word — with it the blend takes every slot of the device and the workgroups beyond
        ``blend_workgroups`` leave at once while the word is non-zero (a scheduling hint: same image either way).
        ``want_radii``: the per-(item, Gaussian) radii as an output ``radii`` (n_items,P) — the call then prepares EVERY
        record in front of the blend (the head kernel over the whole lists) instead of the head of each view's list.  None: yes with
        ``guard='device'`` (what a fired guard's per-call chain writes anyway), no with the host guard; a device-guarded
        caller that does not read them passes False and keeps the short front end.
        ``phase``: 'both', or 'update' then (same arguments, same ``out``) 'blend' — possibly on another stream, ordered by
        the caller's events (``guard='host'`` only).
        ``views_disjoint``: the caller states that every plan view is rendered by at most ONE set of this call (frames
        or samples that share a plan, each with its own views): the sets then share one copy of the per-call record
        arrays, every line of which is written once.  Checked on the device (status bit 8 if two sets name one view).
        ``cameras``: the (V,36) packed cameras this call means to render with (what the reference's ``render`` gets per
        call).  Compared with the plan's on the device: a difference raises status bit 16 and, with
        ``guard='device'``, the call is rendered by the per-call pipeline with these cameras instead."""
        _lib.require_cuda(colors, opacities, scales, rotations, bg)
        dev, P = self.device, self.P
        colors, sc, rot = _f32c(colors), _f32c(scales), _f32c(rotations)
        S = 1 if colors.dim() == 2 else int(colors.size(0))
        opac = _f32c(opacities).reshape(S, P)
        if colors.numel() != S * P * 3 or sc.numel() != S * P * 3 or rot.numel() != S * P * 4:
            raise _lib.OcrfHipError(f'parameter tensors do not match the plan ({S} sets of {P} Gaussians)')
        if item_view is not None:
            _lib.require_cuda(item_view)
            if item_view.dtype != torch.int32 or not item_view.is_contiguous():
                raise _lib.OcrfHipError('item_view must be a contiguous int32 device tensor')
            n_items = int(item_view.numel())
            if n_items == 0 or n_items % S:
                raise _lib.OcrfHipError('the number of items must be a positive multiple of the number of Gaussian sets')
        else:
            n_items = S * self.V                                  # every set renders the plan's views in order
            if S > 1:
                if getattr(self, '_tiled_views', None) is None or self._tiled_views.numel() != n_items:
                    self._tiled_views = (torch.arange(n_items, device=dev, dtype=torch.int32) % self.V).contiguous()
                item_view = self._tiled_views
        H, W = self.H, self.W
        bg = _f32c(bg).reshape(3)
        if out is None:
            out = dict(color=torch.empty(n_items, 3, H, W, device=dev), depth=torch.empty(n_items, 1, H, W, device=dev),
                       final_T=torch.empty(n_items, H, W, device=dev))
        if want_radii is None:
            want_radii = guard == 'device'
        # 1: device guard; + 2: the radii are an OUTPUT of the call (every record is then prepared by the head
        # kernel); a device-guarded call without want_radii hands the armed chain a radii buffer of its own only
        use_guard = {'host': 0, 'device': 1}[guard] | (2 if (want_radii and guard == 'device') else 0)
        self._host_guarded = self._host_guarded or not use_guard
        if cameras is not None:
            _lib.require_cuda(cameras)
            cameras = _f32c(cameras).reshape(-1, 36)
            if cameras.shape != self.cameras.shape:
                raise _lib.OcrfHipError('cameras must be the plan\'s (V, 36) block')
        radii = None
        if want_radii:
            radii = out.get('radii')
            if radii is None:
                radii = out['radii'] = torch.empty(n_items, P, dtype=torch.int32, device=dev)
        elif use_guard:
            # scratch of the armed per-call chain only (owned by the plan, not an output)
            if getattr(self, '_chain_radii', None) is None or self._chain_radii.numel() < n_items * P:
                self._chain_radii = torch.empty(n_items * P, dtype=torch.int32, device=dev)
            radii = self._chain_radii
        L = _lib.lib()
        cur = torch.cuda.current_stream(dev)
        if self._built[1] != cur and cur.cuda_stream not in self._ordered_after_build \
                and not torch.cuda.is_current_stream_capturing():
            # (inside a hipGraph capture the build is either part of the capture, on this branch, or long finished: a wait
            # on an event of a stream outside the capture would drag that stream into it.)  Once per stream and build: a
            # stream's later work is ordered after its own earlier wait, and a wait per render is a barrier packet at the
            # head of the render chain of every step.
            cur.wait_event(self._built[0])
            self._ordered_after_build.add(cur.cuda_stream)
        disjoint = S == 1 or (bool(views_disjoint) and item_view is not None)
        with _lib.on_device(dev):
            dyn = self._scratch(1 if disjoint else S)
            chain = None
            if use_guard:
                need = L.ocrf_rasterize_workspace_bytes(P, n_items)
                if self._chain_ws is None or self._chain_ws.numel() < need:
                    self._chain_ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
                chain = self._chain_ws
            _lib.check(L.ocrf_rasterize_planned(
                _lib.ptr(self.plan), ctypes.c_size_t(self.plan.numel()), P, self.V, ctypes.c_long(self.capacity),
                H, W, S, n_items, _lib.ptr(item_view), _lib.ptr(colors), _lib.ptr(opac), _lib.ptr(sc),
                ctypes.c_float(scale_modifier), _lib.ptr(rot), _lib.ptr(bg), {'median': 0, 'mean': 1}[depth_mode],
                _lib.ptr(out['color']), _lib.ptr(out['depth']), _lib.ptr(out['final_T']), _lib.ptr(radii),
                _lib.ptr(self.status), _lib.ptr(dyn), ctypes.c_size_t(dyn.numel()), use_guard,
                _lib.ptr(self.means3D), _lib.ptr(chain), ctypes.c_size_t(chain.numel() if chain is not None else 0),
                int(blend_workgroups), _lib.ptr(yield_if), {'both': 0, 'update': 1, 'blend': 2}[phase],
                _lib.ptr(cameras), int(disjoint), _lib.ptr(self._bins_buf),
                ctypes.c_size_t(self._bins_buf.numel() if self._bins_buf is not None else 0),
                self.bins[0] if self.bins else 0, self.bins[1] if self.bins else 0, ctypes.c_long(self.cand_capacity),
                ctypes.c_void_p(self._hint.data_ptr() if self._hint is not None else 0), _lib.stream_ptr(dev)),
                'ocrf_rasterize_planned')
        out['status'] = self.status
        return out

    def _read_status(self):
        st = int(self.status.item())
        self.status.zero_()
        host_guarded, self._host_guarded = self._host_guarded, False
        if st & 8:
            raise _lib.OcrfHipError('RasterPlan: a render used a view index outside the plan, or the plan is unusable '
                                    '(its records exceeded the capacity, a view-space depth lay outside [0.125, 8191) m, '
                                    'or a scan of the build gave up): rebuild with a larger capacity')
        return st, host_guarded

    def check(self):
        """Synchronising read of the sticky status word: raises if a render since the last ``check`` was NOT valid — a
        Gaussian beyond ``extent_bound`` or cameras other than the plan's in a call with ``guard='host'``, a bad view
        index, an unusable plan.  The same events in calls with ``guard='device'`` were rendered exactly by the per-call
        pipeline: informational (``exceeded()``).  Clears the word."""
        st, host_guarded = self._read_status()
        if (st & 4) and host_guarded:
            raise _lib.OcrfHipError(
                f'RasterPlan: a Gaussian exceeded the plan\'s extent bound {self.extent_bound:g} — renders with '
                "guard='host' since the last check are invalid; rebuild the plan with a larger bound or render with "
                "guard='device'")
        if (st & 16) and host_guarded:
            raise _lib.OcrfHipError("RasterPlan: a call's cameras were not the plan's — renders with guard='host' since the "
                                    "last check show the PLAN's pose; rebuild(cameras) first or render with guard='device'")
        return True

    def exceeded(self):
        """Like ``check`` but returns whether the extent bound was exceeded (or a call's cameras differed) instead of
        raising for it."""
        st, _ = self._read_status()
        return bool(st & (4 | 16))
