"""Index preparation for the two ``bev_pool_v2`` calls of ``OcRFViewTransformerFull``.

LSS branch (frustum -> ego frame -> voxel ranks), reference
``mmdet3d/models/necks/view_transformer.py``: ``create_frustum`` :77-106, ``get_lidar_coor`` :108-147,
``voxel_pooling_prepare_v2`` :197-255.
HT branch (pillar sample points -> image cells), reference
``mmdet3d/models/necks/view_transformer_ocrf.py``: ``get_reference_points_3d`` :651-673,
``get_projection`` :675-685, ``get_sampling_point`` :687-740, ``fast_sample_prepare`` :785-852.

Split of work (DESIGN.md "index preparation"):
  * per-camera 3x3 / 3x4 algebra and the init-time templates (frustum, pillar grid) are host-side
    torch code, the same calls the reference makes, so they are identical by construction;
  * the per-point arithmetic is written as explicit float32 multiply / add steps (k = 0,1,2 left to
    right, no fused multiply-add), the order torch's CPU batched matmul evaluates them in — that is
    what makes the voxel indices bit-exact against vectors dumped from the reference
    (tests/test_index_prep.py), on the CPU and on the GPU alike.

The functions above the "HIP" section are device-agnostic torch code (plumbing); the sort is
``torch.sort(stable=True)`` where the reference's ``argsort`` is unstable, so the order inside an
interval is deterministic here and unspecified there.

HIP section (``voxel_pooling_prepare_v2_hip``, ``fast_sample_prepare_hip``): the same two
preparations as fused device kernels (csrc/index_prep.hip, C ABI ``ocrf_lss_prepare`` /
``ocrf_ht_prepare``) — per-point arithmetic, an LSD radix sort shaped for voxel-id keys (LSS) or no
sort at all (HT), intervals — a dozen launches instead of ~40 torch passes and two argsorts, with
identical outputs (tests/test_index_prep_gpu.py).
"""
import ctypes

import torch

from . import _lib

__all__ = ['create_frustum', 'get_lidar_coor', 'voxel_pooling_prepare_v2', 'get_reference_points_3d',
           'get_projection', 'get_sampling_point', 'fast_sample_prepare', 'grid_infos',
           'lss_camera_block', 'ht_camera_block', 'voxel_pooling_prepare_v2_hip', 'fast_sample_prepare_hip']


def grid_infos(grid_config):
    """``create_grid_infos`` (view_transformer.py:59-75): lower bound, interval, size (float32)."""
    axes = [grid_config[a] for a in 'xyz']
    lower = torch.Tensor([a[0] for a in axes])
    interval = torch.Tensor([a[2] for a in axes])
    size = torch.Tensor([(a[1] - a[0]) / a[2] for a in axes])
    return lower, interval, size


def create_frustum(depth_cfg, input_size, downsample):
    """(D, H, W, 3) float32 template of (u, v, d) with SID depth bins (view_transformer.py:77-106).
    Host init code; same torch calls as the reference."""
    h_in, w_in = input_size
    h_f, w_f = h_in // downsample, w_in // downsample
    n_bins = torch.arange(*depth_cfg, dtype=torch.float).shape[0]
    cfg = torch.tensor(depth_cfg).float()
    k = torch.arange(n_bins).float()
    bins = torch.exp(torch.log(cfg[0]) + k / (n_bins - 1) * torch.log((cfg[1] - 1) / cfg[0]))
    d = bins.view(-1, 1, 1).expand(-1, h_f, w_f)
    u = torch.linspace(0, w_in - 1, w_f, dtype=torch.float).view(1, 1, w_f).expand(n_bins, h_f, w_f)
    v = torch.linspace(0, h_in - 1, h_f, dtype=torch.float).view(1, h_f, 1).expand(n_bins, h_f, w_f)
    return torch.stack((u, v, d), -1)


def _apply3(m, x, y, z):
    """Rows of ``m`` (..., 3, 3) applied to (x, y, z): separate multiplies and adds, k ascending."""
    def row(i):
        return m[..., i, 0] * x + m[..., i, 1] * y + m[..., i, 2] * z
    return row(0), row(1), row(2)


def get_lidar_coor(frustum, rots, trans, cam2imgs, post_rots, post_trans, bda):
    """Frustum points in the (augmented) ego frame, (B, N, D, H, W, 3) (view_transformer.py:108-147)."""
    B, N, _ = trans.shape
    dev = rots.device
    # tiny host-style algebra, identical calls to the reference's
    inv_post = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3)
    combine = rots.matmul(torch.inverse(cam2imgs)).view(B, N, 1, 1, 1, 3, 3)
    bda_m = bda.view(B, 1, 1, 1, 1, 3, 3)
    pts = frustum.to(dev).to(rots.dtype)[None, None] - post_trans.view(B, N, 1, 1, 1, 3)
    x, y, z = _apply3(inv_post, pts[..., 0], pts[..., 1], pts[..., 2])
    x, y = x * z, y * z                                     # un-project by depth (:138-139)
    x, y, z = _apply3(combine, x, y, z)
    t = trans.view(B, N, 1, 1, 1, 3)
    x, y, z = x + t[..., 0], y + t[..., 1], z + t[..., 2]
    x, y, z = _apply3(bda_m, x, y, z)
    return torch.stack((x, y, z), -1)


def _runs(sorted_ranks):
    _, counts = torch.unique_consecutive(sorted_ranks, return_counts=True)
    ends = torch.cumsum(counts, 0)
    return (ends - counts).int().contiguous(), counts.int().contiguous()


def voxel_pooling_prepare_v2(coor, grid_lower_bound, grid_interval, grid_size):
    """coor (B,N,D,H,W,3) -> ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths
    (int32), or five ``None`` when no point falls in the grid (view_transformer.py:197-255)."""
    B, N, D, H, W, _ = coor.shape
    n_pts = B * N * D * H * W
    dev = coor.device
    cell = ((coor - grid_lower_bound.to(coor)) / grid_interval.to(coor)).long().view(n_pts, 3)  # trunc toward 0
    gx, gy, gz = (int(v) for v in grid_size.tolist())
    inside = (cell[:, 0] >= 0) & (cell[:, 0] < gx) & (cell[:, 1] >= 0) & (cell[:, 1] < gy) & \
             (cell[:, 2] >= 0) & (cell[:, 2] < gz)
    idx = torch.nonzero(inside).squeeze(1)                  # == ranks_depth of the kept points
    if idx.numel() == 0:
        return None, None, None, None, None
    cell = cell[idx]
    per_b = n_pts // B
    batch = torch.div(idx, per_b, rounding_mode='floor')
    hw = H * W
    ranks_feat = torch.div(idx, D * hw, rounding_mode='floor') * hw + idx % hw   # (b*N+n)*HW + h*W + w
    ranks_bev = batch * (gz * gy * gx) + cell[:, 2] * (gy * gx) + cell[:, 1] * gx + cell[:, 0]
    ranks_bev, order = torch.sort(ranks_bev, stable=True)
    starts, lengths = _runs(ranks_bev)
    return (ranks_bev.int().contiguous(), idx[order].int().contiguous(),
            ranks_feat[order].int().contiguous(), starts, lengths)


def get_reference_points_3d(H, W, Z=8, num_points_in_pillar=13, bs=1, device='cuda', dtype=torch.float):
    """Normalised pillar sample grid (bs, P, H*W, 3): 5 'local' + (P-5) 'global' heights
    (view_transformer_ocrf.py:651-673).  Host-style init code."""
    P = num_points_in_pillar
    z_levels = torch.cat((torch.linspace(3, Z - 1, 5, dtype=dtype, device=device),
                          torch.linspace(0.5, Z - 0.5, P - 5, dtype=dtype, device=device)))
    zs = z_levels.view(-1, 1, 1).expand(P, H, W) / Z
    xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device).view(1, 1, W).expand(P, H, W) / W
    ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device).view(1, H, 1).expand(P, H, W) / H
    grid = torch.stack((xs, ys, zs), -1).reshape(P, H * W, 3)
    return grid[None].repeat(bs, 1, 1, 1)


def get_projection(rots, trans, intrins, post_rots, post_trans, bda):
    """lidar2img (B,N,3,4), img_aug (B,N,3,4), and the R / t parts (view_transformer_ocrf.py:675-685)."""
    B, N = rots.shape[:2]
    bda_n = bda.view(B, 1, 3, 3).repeat(1, N, 1, 1)
    k_r = intrins.matmul(torch.inverse(rots))
    l2i_r = k_r.matmul(torch.inverse(bda_n))
    l2i_t = -k_r.matmul(trans.unsqueeze(-1))
    return (torch.cat((l2i_r, l2i_t), -1), torch.cat((post_rots, post_trans.unsqueeze(-1)), -1),
            l2i_r, l2i_t)


def _apply34(m, x, y, z, w):
    def row(i):
        return m[..., i, 0] * x + m[..., i, 1] * y + m[..., i, 2] * z + m[..., i, 3] * w
    return row(0), row(1), row(2)


def get_sampling_point(reference_points, pc_range, depth_range, lidar2img, img_aug, image_shapes):
    """Project pillar samples into every camera (view_transformer_ocrf.py:687-740).

    ``reference_points`` (B,Z,Nq,3) normalised is scaled to metres IN PLACE, like the reference
    (:690-692) — callers read the voxel centres back from it.
    -> coor (B,N,Z,Nq,3) = (u/W_in, v/H_in, (d-d0)/(d1-d0)), mask (B,N,Z,Nq,1) bool,
       [points_lidar (B,Z*Nq,4), uv (B,N,Z,Nq,2), lidar2img (B,N,1,3,4)]."""
    rp = reference_points
    rp[..., 0:1] = rp[..., 0:1] * (pc_range[3] - pc_range[0]) + pc_range[0]
    rp[..., 1:2] = rp[..., 1:2] * (pc_range[4] - pc_range[1]) + pc_range[1]
    rp[..., 2:3] = rp[..., 2:3] * (pc_range[5] - pc_range[2]) + pc_range[2]
    B, Z, Nq = rp.shape[:3]
    N = lidar2img.size(1)
    flat = rp.view(B, -1, 3)
    ones = torch.ones_like(flat[..., :1])
    points_lidar = torch.cat((flat, ones), -1)
    x, y, z, w = (points_lidar[..., i].view(B, 1, Z * Nq) for i in range(4))
    l2i = lidar2img.view(B, N, 1, 3, 4)
    aug = img_aug.view(B, N, 1, 3, 4)
    cx, cy, cz = _apply34(l2i, x, y, z, w)
    eps = 1e-5
    depth = cz.clone()
    mask = cz > eps
    den = torch.maximum(cz, torch.ones_like(cz) * eps)
    u, v, _ = _apply34(aug, cx / den, cy / den, cz, torch.ones_like(cz))
    u = u / image_shapes[1]
    v = v / image_shapes[0]
    uv = torch.stack((u, v), -1).view(B, N, Z, Nq, 2)
    depth = depth.view(B, N, Z, Nq, 1)
    mask = mask.view(B, N, Z, Nq, 1)
    mask = mask & (uv[..., 0:1] > 0.0) & (uv[..., 0:1] < 1.0) & (uv[..., 1:2] > 0.0) & (uv[..., 1:2] < 1.0)
    if depth_range is not None:
        depth = (depth - depth_range[0]) / (depth_range[1] - depth_range[0])
        mask = mask & (depth > 0.0) & (depth < 1.0)
    mask = torch.nan_to_num(mask)
    return torch.cat((uv, depth), -1), mask, [points_lidar, uv, l2i]


def fast_sample_prepare(coor, mask, W, H, D):
    """coor (B,N,Z,Nq,3) normalised + mask -> the five int32 rank vectors of the HT pooling
    (view_transformer_ocrf.py:785-852).  ``W``/``H``: feature-map size, ``D``: depth bins."""
    B, N, Z, Nq, _ = coor.shape
    n_pts = B * N * Z * Nq
    scale = torch.tensor([W, H, D], dtype=coor.dtype, device=coor.device)
    cell = (coor * scale).round().long().view(n_pts, 3)     # half-to-even
    hi = torch.tensor([W - 1, H - 1, D - 1], device=coor.device)
    cell = torch.minimum(cell.clamp_(min=0), hi)
    idx = torch.nonzero(mask.reshape(-1)).squeeze(1)
    if idx.numel() == 0:
        return None, None, None, None, None
    cell = cell[idx]
    cam = torch.div(idx, Z * Nq, rounding_mode='floor')      # b*N + n
    pillar = torch.div(idx, N * Z * Nq, rounding_mode='floor') * Nq + idx % Nq   # b*Nq + q
    ranks_depth = cam * (D * W * H) + cell[:, 2] * (W * H) + cell[:, 1] * W + cell[:, 0]
    ranks_depth.clamp_(min=0, max=B * N * D * W * H - 1)
    ranks_feat = cam * (W * H) + cell[:, 1] * W + cell[:, 0]
    ranks_feat.clamp_(min=0, max=B * N * W * H - 1)
    ranks_bev, order = torch.sort(pillar, stable=True)
    starts, lengths = _runs(ranks_bev)
    return (ranks_bev.int().contiguous(), ranks_depth[order].int().contiguous(),
            ranks_feat[order].int().contiguous(), starts, lengths)


# ----------------------------------------------------------------------------------------------
# HIP
# ----------------------------------------------------------------------------------------------
def lss_camera_block(rots, trans, cam2imgs, post_rots, post_trans, bda):
    """(B*N, 33) float32: inv(post_rots) | rots.inv(cam2imgs) | post_trans | trans | bda per
    camera-frame — the tiny algebra of get_lidar_coor (view_transformer.py:128-146), made with the
    same torch calls on whatever device the calibration lives on."""
    B, N = trans.shape[:2]
    inv_post = torch.inverse(post_rots).reshape(B * N, 9)
    combine = rots.matmul(torch.inverse(cam2imgs)).reshape(B * N, 9)
    bda_n = bda.view(B, 1, 9).expand(B, N, 9).reshape(B * N, 9)
    return torch.cat((inv_post, combine, post_trans.reshape(B * N, 3), trans.reshape(B * N, 3), bda_n), 1).float().contiguous()


def ht_camera_block(lidar2img, img_aug):
    """(B*N, 24) float32: lidar2img 3x4 | img_aug 3x4 per camera-frame (get_projection)."""
    B, N = lidar2img.shape[:2]
    return torch.cat((lidar2img.reshape(B * N, 12), img_aug.reshape(B * N, 12)), 1).float().contiguous()


class _RankBuffers:
    """Grow-only output buffers of one preparation (rank vectors at their capacity)."""

    def __init__(self):
        self.cap_pts = self.cap_iv = 0

    def get(self, dev, n_pts, n_iv):
        if n_pts > self.cap_pts or n_iv > self.cap_iv or self.bufs[0].device != dev:
            self.cap_pts, self.cap_iv = max(n_pts, self.cap_pts), max(n_iv, self.cap_iv)
            self.bufs = [torch.empty(self.cap_pts, dtype=torch.int32, device=dev) for _ in range(3)] + \
                        [torch.empty(self.cap_iv, dtype=torch.int32, device=dev) for _ in range(2)] + \
                        [torch.zeros(2, dtype=torch.int32, device=dev)]
        return self.bufs


_HOST_FLOATS = {}


def _host_floats(values):
    """A host float32 tensor holding ``values`` that lives as long as the process: the C entry points read a few
    floats through HOST pointers (grid bounds, pc_range), and a recorded step (``_lib.StepRecorder``) replays the call
    with the pointer it saw."""
    key = tuple(float(v) for v in values)
    t = _HOST_FLOATS.get(key)
    if t is None:
        t = _HOST_FLOATS[key] = torch.tensor(key, dtype=torch.float32)
    return t


def _prep_tag(dev):
    """Scratch tag of the index preparations: per STREAM, because the scratch holds cross-workgroup state (the
    look-back words of the prefix sums) — two preparations in flight on different streams must not share it."""
    return f'index_prep@{_lib.stream_ptr(dev).value or 0}'


def _trim(bufs, counts_host):
    n_p, n_v = int(counts_host[0]), int(counts_host[1])
    if n_p < 0:
        raise _lib.OcrfHipError('index preparation: a look-back prefix sum gave up waiting (its scratch was overwritten '
                                'while in use — two preparations sharing one scratch buffer?); the rank vectors are not valid')
    if n_p == 0:
        return None, None, None, None, None
    return bufs[0][:n_p], bufs[1][:n_p], bufs[2][:n_p], bufs[3][:n_v], bufs[4][:n_v]


def voxel_pooling_prepare_v2_hip(frustum, cam_block, B, N, grid_lower_bound, grid_interval, grid_size,
                                 buffers=None, sync=True):
    """HIP ``get_lidar_coor`` + ``voxel_pooling_prepare_v2`` (view_transformer.py:108-147,197-255).

    ``frustum`` (D,H,W,3) and ``cam_block`` (B*N,33, ``lss_camera_block``) on the device; the grid
    tensors are the host tensors of ``grid_infos``.  Returns ranks_bev, ranks_depth, ranks_feat,
    interval_starts, interval_lengths (int32 views of ``buffers``, or five ``None``); with
    ``sync=False`` returns ``(buffers, counts)`` untrimmed and does not read the device."""
    _lib.require_cuda(frustum, cam_block)
    dev = frustum.device
    D, H, W, _ = frustum.shape
    gx, gy, gz = (int(v) for v in grid_size.tolist())
    lower = _host_floats(grid_lower_bound.detach().float().cpu().tolist())
    interval = _host_floats(grid_interval.detach().float().cpu().tolist())
    n_pts = B * N * D * H * W
    bufs = (buffers if buffers is not None else _RankBuffers()).get(dev, n_pts, min(n_pts, B * gz * gy * gx))
    L = _lib.lib()
    with _lib.on_device(dev):
        need = L.ocrf_lss_prepare_workspace_bytes(B, N, D, H, W, gx, gy, gz)
        ws = _lib.workspace.get(dev, need, _prep_tag(dev))
        _lib.check(L.ocrf_lss_prepare(
            B, N, D, H, W, _lib.ptr(frustum.contiguous()), _lib.ptr(cam_block), ctypes.c_void_p(lower.data_ptr()),
            ctypes.c_void_p(interval.data_ptr()), gx, gy, gz, *[_lib.ptr(b) for b in bufs], _lib.ptr(ws),
            ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev)), 'ocrf_lss_prepare')
    if not sync:
        return bufs[:5], bufs[5]
    return _trim(bufs, bufs[5].cpu())


def fast_sample_prepare_hip(ref_template, cam_block, B, N, pc_range, image_shapes, depth_range, W, H, D,
                            buffers=None, sync=True):
    """HIP ``get_sampling_point`` + ``fast_sample_prepare`` (view_transformer_ocrf.py:687-740,785-852).

    ``ref_template`` (Z,Nq,3): one sample of ``get_reference_points_3d`` (normalised, NOT yet scaled
    to metres); ``cam_block`` (B*N,24) from ``ht_camera_block``; ``image_shapes`` = (H_in, W_in);
    ``W``/``H``/``D``: feature-map size and depth bins.  Returns like ``voxel_pooling_prepare_v2_hip``."""
    _lib.require_cuda(ref_template, cam_block)
    dev = ref_template.device
    Z, Nq, _ = ref_template.shape
    pc = _host_floats(pc_range)
    n_pts = B * N * Z * Nq
    bufs = (buffers if buffers is not None else _RankBuffers()).get(dev, n_pts, B * Nq)
    L = _lib.lib()
    with _lib.on_device(dev):
        need = L.ocrf_ht_prepare_workspace_bytes(B, Nq)
        ws = _lib.workspace.get(dev, need, _prep_tag(dev))
        _lib.check(L.ocrf_ht_prepare(
            B, N, Z, Nq, int(W), int(H), int(D), _lib.ptr(ref_template.contiguous()), _lib.ptr(cam_block),
            ctypes.c_void_p(pc.data_ptr()), ctypes.c_float(image_shapes[1]), ctypes.c_float(image_shapes[0]),
            ctypes.c_float(depth_range[0]), ctypes.c_float(depth_range[1]), *[_lib.ptr(b) for b in bufs], _lib.ptr(ws),
            ctypes.c_size_t(ws.numel()), _lib.stream_ptr(dev)), 'ocrf_ht_prepare')
    if not sync:
        return bufs[:5], bufs[5]
    return _trim(bufs, bufs[5].cpu())


def geometry_blocks_hip(rots, trans, intrins, post_rots, post_trans, bda, c2w, image_shapes, znear=0.01, zfar=999.9):
    """The per-forward calibration algebra on the device (C ABI ``ocrf_geometry_blocks``): from the calibration
    tensors of the input tuple (all on the GPU) to ``lss_camera_block`` (B*N,33), ``ht_camera_block`` (B*N,24) and
    the rasteriser's camera rows (B,N,36) of every camera-frame — no host read, no synchronisation.  Values agree
    with the host formulation (``lss_camera_block`` / ``get_projection`` / ``camera_from_calibration``) to ~1 ulp,
    not bit for bit (double-precision cofactor inverses rounded once vs the float32 LAPACK chain)."""
    ts = [t.detach().float().contiguous() for t in (rots, trans, intrins, post_rots, post_trans, bda)]
    _lib.require_cuda(*ts)
    dev = ts[0].device
    B, N = ts[1].shape[:2]
    c2w_d = c2w.detach().float().contiguous().to(dev) if c2w is not None else None
    lss = torch.empty(B * N, 33, device=dev)
    ht = torch.empty(B * N, 24, device=dev)
    cam = torch.empty(B, N, 36, device=dev) if c2w_d is not None else None
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ocrf_geometry_blocks(
            B, N, *[_lib.ptr(t) for t in ts], _lib.ptr(c2w_d), int(image_shapes[0]), int(image_shapes[1]),
            ctypes.c_float(znear), ctypes.c_float(zfar), _lib.ptr(lss), _lib.ptr(ht), _lib.ptr(cam),
            _lib.stream_ptr(dev)), 'ocrf_geometry_blocks')
    return lss, ht, cam


def ht_project_hip(ref_template, cam_block, B, N, pc_range, image_shapes, depth_range):
    """HIP ``get_sampling_point`` outputs that the colour sampling needs (view_transformer_ocrf.py:
    687-740, 1057-1066): -> pix (B,N,Z,Nq,2) pixel coordinates, mask (B,N,Z,Nq) bool, voxel
    (B,Z,Nq,3) metric centres.  Inputs as ``fast_sample_prepare_hip``."""
    _lib.require_cuda(ref_template, cam_block)
    dev = ref_template.device
    Z, Nq, _ = ref_template.shape
    pc = torch.tensor([float(v) for v in pc_range], dtype=torch.float32)
    pix = torch.empty(B, N, Z, Nq, 2, device=dev)
    mask = torch.empty(B, N, Z, Nq, dtype=torch.bool, device=dev)
    voxel = torch.empty(B, Z, Nq, 3, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ocrf_ht_project(
            B, N, Z, Nq, _lib.ptr(ref_template.contiguous()), _lib.ptr(cam_block), ctypes.c_void_p(pc.data_ptr()),
            ctypes.c_float(image_shapes[1]), ctypes.c_float(image_shapes[0]), ctypes.c_float(depth_range[0]),
            ctypes.c_float(depth_range[1]), _lib.ptr(pix), _lib.ptr(mask), _lib.ptr(voxel), _lib.stream_ptr(dev)),
            'ocrf_ht_project')
    return pix, mask, voxel
