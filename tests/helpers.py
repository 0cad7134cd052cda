"""Shared input builders for the tests (seeded, synthetic)."""
import numpy as np

from oracle import index_prep as ip
from ocrfdet_amd import synthetic


def lss_ranks(cfg):
    """Reference-style LSS rank vectors for a config, from the numpy oracle."""
    r = synthetic.rig(cfg.n_cams, cfg.input_size, cfg.batch)
    fr = ip.create_frustum(cfg.grid['depth'], cfg.input_size, cfg.downsample)
    coor = ip.get_lidar_coor(fr, r['rots'], r['trans'], r['intrins'], r['post_rots'], r['post_trans'], r['bda'])
    lower = [cfg.grid[a][0] for a in 'xyz']
    interval = [cfg.grid[a][2] for a in 'xyz']
    return ip.voxel_pooling_prepare_v2(coor, lower, interval, cfg.bev_xyz)


def ht_ranks(cfg):
    r = synthetic.rig(cfg.n_cams, cfg.input_size, cfg.batch)
    X, Y, _ = cfg.bev_xyz
    ref = ip.get_reference_points_3d(Y, X, bs=cfg.batch, num_points_in_pillar=cfg.num_height)
    l2i, aug = ip.get_projection(r['rots'], r['trans'], r['intrins'], r['post_rots'], r['post_trans'], r['bda'])
    coor, mask, _ = ip.get_sampling_point(ref, cfg.pc_range, cfg.grid['depth'], l2i, aug, cfg.input_size)
    Hf, Wf = cfg.feat_hw
    return ip.fast_sample_prepare(coor, mask, Wf, Hf, cfg.D)


def pool_inputs(cfg, seed=0):
    """depth (B,N,D,H,W), feat channels-last (B,N,H,W,C) as numpy float32."""
    depth, feat = synthetic.depth_and_feat(cfg, seed)
    B, N = cfg.batch, cfg.n_cams
    H, W = cfg.feat_hw
    depth = depth.numpy().reshape(B, N, cfg.D, H, W)
    feat = np.ascontiguousarray(feat.numpy().reshape(B, N, cfg.channels, H, W).transpose(0, 1, 3, 4, 2))
    return depth, feat


def random_pool_problem(rng, n_points, n_voxels, c, n_depth=5000, n_feat=700, skew=True):
    """A random but valid bev_pool problem: sorted ranks_bev with (optionally skewed) interval
    lengths, random gather indices."""
    if skew:
        w = rng.pareto(1.2, n_voxels) + 0.05
    else:
        w = np.ones(n_voxels)
    vox = rng.choice(n_voxels, size=n_points, p=w / w.sum())
    rb = np.sort(vox).astype(np.int32)
    rd = rng.integers(0, n_depth, n_points).astype(np.int32)
    rf = rng.integers(0, n_feat, n_points).astype(np.int32)
    depth = rng.random(n_depth, dtype=np.float32)
    feat = rng.standard_normal((n_feat, c)).astype(np.float32)
    kept = np.ones(n_points, bool)
    kept[1:] = rb[1:] != rb[:-1]
    starts = np.nonzero(kept)[0].astype(np.int32)
    lengths = np.diff(np.append(starts, n_points)).astype(np.int32)
    return depth, feat, rd, rf, rb, starts, lengths


def simple_camera(W, H, fx=None, fy=None, cam_pos=(0.0, 0.0, 0.0)):
    """Camera at ``cam_pos`` looking down +z with identity rotation; returns the transposed view /
    full-projection matrices and tan(fov/2), built with the product's camera helpers."""
    import torch
    from ocrfdet_amd import gaussian_renderer as gr
    fx = fx or 0.8 * W
    fy = fy or fx
    K = np.array([[fx, 0, W / 2.0], [0, fy, H / 2.0], [0, 0, 1.0]])
    proj = gr.getProjectionMatrix(0.01, 999.9, K, H, W).transpose(0, 1)
    w2v = np.eye(4, dtype=np.float32)
    w2v[:3, 3] = -np.asarray(cam_pos, np.float32)
    view_t = torch.from_numpy(w2v).transpose(0, 1).contiguous()
    full = view_t.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    return view_t.numpy(), full.numpy(), W / (2 * fx), H / (2 * fy)


def random_gaussians(rng, n, z_range=(1.0, 30.0), xy_extent=12.0, scale=(0.05, 0.8)):
    xyz = np.stack([rng.uniform(-xy_extent, xy_extent, n), rng.uniform(-xy_extent * 0.4, xy_extent * 0.4, n),
                    rng.uniform(*z_range, n)], 1).astype(np.float32)
    scales = rng.uniform(*scale, (n, 3)).astype(np.float32)
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, (n, 1)).astype(np.float32)
    rgb = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    return xyz, rgb, opac, scales, q


CORE_CFG = dict(name='core_small_6cam_64x176_bev48x48', input_size=(64, 176),
                grid=dict(x=[-19.2, 19.2, 0.8], y=[-19.2, 19.2, 0.8], z=[-5.0, 3.0, 8.0], depth=[1.0, 60.0, 0.5]),
                pc_range=(-19.2, -19.2, -5.0, 19.2, 19.2, 3.0))


def core_fixture():
    """tests/golden/core_small.npz (reference ``OcRFViewTransformerFull.forward`` in eval mode on a
    small 6-camera configuration) -> (cfg, golden dict, state dict of numpy arrays)."""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'core_small.npz')))
    state = {k[len('state.'):]: v for k, v in g.items() if k.startswith('state.')}
    return synthetic.PathConfig(**CORE_CFG), g, state


def core_geometry(cfg, batch):
    """Pillar projections of the HT branch for the core fixture, from the numpy oracle:
    voxel centres (B,Zh,Q,3), pixel coordinates (B,N,Zh,Q,2), mask (B,N,Zh,Q,1)."""
    r = synthetic.rig(cfg.n_cams, cfg.input_size, batch)
    X, Y, _ = cfg.bev_xyz
    ref = ip.get_reference_points_3d(Y, X, bs=batch, num_points_in_pillar=cfg.num_height)
    l2i, aug = ip.get_projection(r['rots'], r['trans'], r['intrins'], r['post_rots'], r['post_trans'], r['bda'])
    coor, mask, extra = ip.get_sampling_point(ref, cfg.pc_range, cfg.grid['depth'], l2i, aug, cfg.input_size)
    pix = coor[..., :2].copy()
    pix[..., 0] *= np.float32(cfg.input_size[1])
    pix[..., 1] *= np.float32(cfg.input_size[0])
    return extra, pix, mask, r
