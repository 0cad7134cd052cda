"""ORACLE — test infrastructure only.

CPU restatement of the reference's hot path, used as the *checker* by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.  Nothing under
``ocrfdet_amd/`` imports this package; the product path fails loudly without its HIP library.

Parts:
  * ``liboracle.so`` (plain C, ``oracle/bev_pool_ref.c``, ``oracle/rasterize_ref.c``) — the two
    native kernels of the reference restated loop-for-thread.
  * ``oracle.index_prep`` (numpy) — LSS / HT rank preparation.
  * ``oracle.hoa`` (numpy) — Height-aware Opacity-based Attention blocks.
  * ``oracle.sh`` (numpy) — the rasteriser's spherical-harmonics colours and their backward.

Each function cites the reference file:line it follows.  See DESIGN.md "Oracle" for what pins it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with gcc (a few seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("bev_pool_ref.c", "rasterize_ref.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_rasterize_forward.restype = ctypes.c_long
        _LIB.oracle_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    """OpenMP team size of the oracle's entry points from now on (1: the scalar port)."""
    lib().oracle_set_num_threads(int(n))


# ----------------------------------------------------------------------------------------------
# bev_pool_v2
# ----------------------------------------------------------------------------------------------
def bev_pool_v2_raw(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                    interval_starts, interval_lengths):
    """``QuickCumsumCuda.forward`` (mmdet3d/ops/bev_pool_v2/bev_pool.py:16-41): returns the
    (B,Z,Y,X,C) tensor the extension writes, before the wrapper's permute."""
    depth, feat = _f32(depth), _f32(feat)
    rd, rf, rb = _i32(ranks_depth), _i32(ranks_feat), _i32(ranks_bev)
    st, ln = _i32(interval_starts), _i32(interval_lengths)
    out = np.zeros(tuple(int(s) for s in bev_feat_shape), dtype=np.float32)   # bev_pool.py:27
    c = feat.shape[-1]                                                        # bev_pool.cpp:40
    lib().oracle_bev_pool_v2(ctypes.c_int(c), ctypes.c_int(st.shape[0]), _p(depth), _p(feat),
                             _p(rd), _p(rf), _p(rb), _p(st), _p(ln), _p(out))
    return out


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                interval_starts, interval_lengths):
    """``bev_pool_v2`` (bev_pool.py:86-92): (B,C,Z,Y,X) contiguous."""
    out = bev_pool_v2_raw(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                          interval_starts, interval_lengths)
    return np.ascontiguousarray(out.transpose(0, 4, 1, 2, 3))


def intervals_from_sorted(ranks):
    """Run-length intervals of a sorted rank vector (view_transformer.py:245-252,
    bev_pool.py:50-57)."""
    ranks = np.asarray(ranks)
    kept = np.ones(ranks.shape[0], dtype=bool)
    kept[1:] = ranks[1:] != ranks[:-1]
    starts = np.nonzero(kept)[0].astype(np.int32)
    lengths = np.zeros_like(starts)
    if starts.size:
        lengths[:-1] = starts[1:] - starts[:-1]
        lengths[-1] = ranks.shape[0] - starts[-1]
    return starts, lengths


def bev_pool_v2_backward(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev):
    """``QuickCumsumCuda.backward`` (bev_pool.py:43-83).  ``out_grad`` is (B,Z,Y,X,C).

    The reference's ``argsort`` is unstable; a stable sort is used here.  Where a depth cell
    repeats inside the list (HT path) the reference's plain store is last-writer-wins in
    unspecified order (bev_pool_cuda.cu:103-104) — compare such cells with care.
    """
    out_grad, depth, feat = _f32(out_grad), _f32(depth), _f32(feat)
    rd, rf, rb = _i32(ranks_depth), _i32(ranks_feat), _i32(ranks_bev)
    order = np.argsort(rf, kind="stable")
    rf, rd, rb = _i32(rf[order]), _i32(rd[order]), _i32(rb[order])
    st, ln = intervals_from_sorted(rf)
    depth_grad = np.zeros_like(depth)
    feat_grad = np.zeros_like(feat)
    c = out_grad.shape[-1]                                                    # bev_pool.cpp:86
    lib().oracle_bev_pool_v2_grad(ctypes.c_int(c), ctypes.c_int(st.shape[0]), _p(out_grad),
                                  _p(depth), _p(feat), _p(rd), _p(rf), _p(rb), _p(st), _p(ln),
                                  _p(depth_grad), _p(feat_grad))
    return depth_grad, feat_grad


# ----------------------------------------------------------------------------------------------
# rasteriser forward
# ----------------------------------------------------------------------------------------------
def rasterize_forward(means3D, colors_precomp, opacities, scales, rotations, viewmatrix,
                      projmatrix, tanfovx, tanfovy, image_height, image_width, bg,
                      scale_modifier=1.0, depth_mode="median", ambiguity=None):
    """Forward of ``GaussianRasterizer`` with ``colors_precomp`` and (scales, rotations)
    (rasterize_points.cu:35-115 -> rasterizer_impl.cu:198-336).  Returns a dict with
    ``color`` (3,H,W), ``depth`` (1,H,W), ``final_T`` (H,W), ``n_contrib`` (H,W), ``radii`` (P),
    ``num_rendered`` and the per-Gaussian state (``means2D``, ``depths``, ``conic_opacity``,
    ``tiles_touched``)."""
    means3D, colors, opac = _f32(means3D), _f32(colors_precomp), _f32(opacities).reshape(-1)
    scales, rots = _f32(scales), _f32(rotations)
    vm, pm, bg = _f32(viewmatrix).reshape(16), _f32(projmatrix).reshape(16), _f32(bg).reshape(3)
    P = means3D.shape[0]
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:57-59
    H, W = int(image_height), int(image_width)
    out = dict(
        color=np.empty((3, H, W), np.float32), depth=np.empty((1, H, W), np.float32),
        final_T=np.empty((H, W), np.float32), n_contrib=np.empty((H, W), np.uint32),
        radii=np.zeros((P,), np.int32), means2D=np.zeros((P, 2), np.float32),
        depths=np.zeros((P,), np.float32), conic_opacity=np.zeros((P, 4), np.float32),
        tiles_touched=np.zeros((P,), np.uint32))
    mode = {"median": 0, "mean": 1}[depth_mode]
    amb = None
    if ambiguity is not None:
        # (tol_alpha, tol_T): also return ``ambiguous`` (H,W) uint8 — pixels one of whose decisions lies within
        # these relative tolerances of its threshold (an exp of another rounding may decide them the other way)
        amb = np.zeros((H, W), np.uint8)
        lib().oracle_set_ambiguity_map(_p(amb), ctypes.c_float(ambiguity[0]), ctypes.c_float(ambiguity[1]))
    R = lib().oracle_rasterize_forward(
        ctypes.c_int(P), _p(bg), ctypes.c_int(W), ctypes.c_int(H), _p(means3D), _p(colors),
        _p(opac), _p(scales), ctypes.c_float(scale_modifier), _p(rots), _p(vm), _p(pm),
        ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), ctypes.c_int(mode), _p(out["color"]),
        _p(out["depth"]), _p(out["final_T"]), _p(out["n_contrib"]), _p(out["radii"]),
        _p(out["means2D"]), _p(out["depths"]), _p(out["conic_opacity"]), _p(out["tiles_touched"]))
    if amb is not None:
        lib().oracle_set_ambiguity_map(None, ctypes.c_float(0.0), ctypes.c_float(0.0))
        out["ambiguous"] = amb
    if R < 0:
        raise MemoryError("oracle_rasterize_forward: allocation failed")
    out["num_rendered"] = int(R)
    return out


def rasterize_backward(grad_color, means3D, colors_precomp, opacities, scales, rotations, viewmatrix,
                       projmatrix, tanfovx, tanfovy, image_height, image_width, bg, scale_modifier=1.0):
    """Backward of the colour output of ``rasterize_forward`` w.r.t. means3D, colours, opacities,
    scales, rotations (rasterize_points.cu:117-196 -> rasterizer_impl.cu:338-434 ->
    cuda_rasterizer/backward.cu).  ``grad_color`` is (3,H,W).  Returns a dict with
    ``means3D`` (P,3), ``means2D`` (P,3), ``colors`` (P,3), ``opacities`` (P,1), ``scales`` (P,3),
    ``rotations`` (P,4), ``cov3D`` (P,6)."""
    means3D, colors, opac = _f32(means3D), _f32(colors_precomp), _f32(opacities).reshape(-1)
    scales, rots = _f32(scales), _f32(rotations)
    vm, pm, bg = _f32(viewmatrix).reshape(16), _f32(projmatrix).reshape(16), _f32(bg).reshape(3)
    g = _f32(grad_color)
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    assert g.shape == (3, H, W)
    out = dict(means2D=np.zeros((P, 3), np.float32), colors=np.zeros((P, 3), np.float32),
               opacities=np.zeros((P, 1), np.float32), means3D=np.zeros((P, 3), np.float32),
               cov3D=np.zeros((P, 6), np.float32), scales=np.zeros((P, 3), np.float32),
               rotations=np.zeros((P, 4), np.float32))
    L = lib()
    L.oracle_rasterize_backward.restype = ctypes.c_long
    R = L.oracle_rasterize_backward(
        ctypes.c_int(P), _p(bg), ctypes.c_int(W), ctypes.c_int(H), _p(means3D), _p(colors), _p(opac), _p(scales),
        ctypes.c_float(scale_modifier), _p(rots), _p(vm), _p(pm), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy),
        _p(g), _p(out['means2D']), _p(out['colors']), _p(out['opacities']), _p(out['means3D']), _p(out['cov3D']),
        _p(out['scales']), _p(out['rotations']))
    if R < 0:
        raise MemoryError('oracle_rasterize_backward: allocation failed')
    return out
