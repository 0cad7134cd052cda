"""Pins oracle/core.py (numpy restatement of SURVEY §8a rows a11-a15, a23, a27, a28) to vectors
dumped from the reference's own Python (tests/golden/{heads,color_cfg0,core_small}.npz)."""
import os

import numpy as np
import pytest

from oracle import core as oc
from oracle import hoa as ohoa
from tests import helpers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def core():
    return helpers.core_fixture()


def close(got, want, tol=1e-4, what=''):
    err = float(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max())
    assert err <= tol, f'{what}: max|err| {err:.3e} > {tol}'


def test_heads_and_lift_vs_reference():
    g = dict(np.load(os.path.join(GOLDEN, 'heads.npz')))
    p = {k.replace('S.', 'S_MLP.').replace('R.', 'R_MLP.').replace('A.', 'A_MLP.').replace('C.', 'C_MLP.'): v
         for k, v in g.items() if k.split('.')[0] in 'SRAC'}
    op, sc, rot, col = oc.gauss_heads(g['heads_feat'], g['heads_rgb'], p)
    close(op, g['A_out'], 1e-6, 'opacity'), close(sc, g['S_out'], 1e-6, 'scales')
    close(rot, g['R_out'], 1e-6, 'rotations'), close(col, g['C_out'], 1e-6, 'colour')
    pv = {k.replace('vfe.', 'ObtainVoxelFeature.'): v for k, v in g.items() if k.startswith('vfe.')}
    close(oc.voxel_lift(g['vfe_in'], pv), g['vfe_out'], 1e-6, 'voxel lift')


def test_colour_sampling_vs_reference():
    g = dict(np.load(os.path.join(GOLDEN, 'color_cfg0.npz')))
    imgs = g['imgs'].astype(np.float32)
    vals = oc.lidar_points_to_image_values(g['pix'], imgs, g['mask'])
    close(vals[0, :, :, ::37], g['img_values_slice'], 2e-4, 'img_values')       # values up to 255
    close(oc.color_voxels_avg(vals, g['mask']), g['avg_color'], 2e-4, 'avg colour')
    sparse = oc.retain_valid_pixels(imgs, g['pix'], g['mask'])
    assert np.array_equal(sparse.astype(np.uint8), g['sparse'])
    assert int((sparse != 255).any(2).sum()) == int(g['sparse_nkept'])


def test_prefilter_vs_reference(core):
    cfg, g, _ = core
    depth, fdepth, sem, ffeat = oc.prefilter(g['pre'], cfg.D, cfg.channels, 1.0 / cfg.D, 0.25)
    close(depth, g['depth'], 1e-6, 'depth softmax'), close(sem, g['semantic'], 1e-6, 'semantic')
    assert fdepth.shape == depth.shape and ffeat.shape == (g['pre'].shape[0], cfg.channels) + g['pre'].shape[2:]
    assert ((fdepth == 0) | (fdepth == depth)).all()


def test_bev_fusion_vs_reference(core):
    cfg, g, p = core
    B, (X, Y, _) = int(g['batch']), cfg.bev_xyz
    ch = oc.dual_feat_fusion(g['lss_feat'], g['ht_feat'], p)
    close(ch, g['channel_feat'], 1e-5, 'fuser')
    pos = oc.learned_positional_encoding(p, 'positional_encoding', B, Y, X)
    logit = oc.prob_net(pos + ch, p)
    close(logit, g['bev_mask_logit'], 1e-4, 'ProbNet')
    assert (oc.bev_geom_attention(ch, logit, p) * ch).shape == g['bev_feat'].shape


def test_sampling_heads_and_nerf_vs_reference(core):
    cfg, g, p = core
    B, (X, Y, _) = int(g['batch']), cfg.bev_xyz
    voxel, pix, mask, _ = helpers.core_geometry(cfg, B)
    raw = g['raw'].astype(np.float32)
    avg = oc.color_voxels_avg(oc.lidar_points_to_image_values(pix, raw, mask), mask)
    # white-noise images: a 1-ulp difference of a pixel coordinate (numpy vs torch matmul order) moves
    # a bilinear tap by up to 255 * 1.5e-5; compared in the 0..1 unit the colour enters the heads in
    close(avg / 255.0, g['colored_avg'] / 255.0, 2e-4, 'colored voxels')
    cams = [int(c) for c in g['cam_idx_list']]
    sparse = oc.retain_valid_pixels(raw, pix, mask)
    for b, c in enumerate(cams):
        assert np.array_equal(sparse[b, c].astype(np.uint8), g['sparse_sel'][b])
    lift = oc.voxel_lift(g['ht_feat'], p)
    x = g['x'].astype(np.float32)
    H, W = cfg.input_size
    for b, c in enumerate(cams):
        op, sc, rot, col = oc.gauss_heads(lift[b].reshape(-1, cfg.channels), (avg[b].reshape(-1, 3) / np.float32(255.0)), p)
        close(op[::5], g[f'gauss_opacity{b}'], 1e-5, 'opacity'), close(sc[::5], g[f'gauss_scales{b}'], 1e-5, 'scales')
        close(rot[::5], g[f'gauss_rot{b}'], 1e-5, 'rot'), close(col[::5], g[f'gauss_rgb{b}'], 1e-5, 'rgb')
        close(voxel[b].reshape(-1, 3)[::5], g[f'gauss_xyz{b}'], 1e-5, 'xyz')
        feat = oc.resize_network(x[b], p)                              # (6,80,H,W)
        alpha = oc.nerf_alpha(feat, p)                                 # (6,H,W)
        # the reference views the (6,H,W,1) stack as (1,6,1,W,H) (:1123): same memory, swapped extents
        alpha_img = alpha.reshape(1, 6, 1, W, H)
        vals = oc.lidar_points_to_image_values(pix[b:b + 1], alpha_img, mask[b:b + 1])
        close(oc.color_voxels_avg(vals, mask[b:b + 1]), g['alpha_lidar'][b:b + 1], 1e-4, 'alpha_lidar')
        img_n, dep_n = oc.nerf_render(feat[c], alpha[c], sparse[b, c], p)
        close(img_n, g['render_N'][b], 1e-5, 'render_N'), close(dep_n, g['render_depth_N'][b], 1e-5, 'render_depth_N')


def test_core_chain_vs_reference(core):
    """HOA-1 -> HOA-2 -> geometry attention -> HOA-3 from the fixture's intermediates to the
    final BEV feature and the returned opacity view."""
    cfg, g, p = core
    B, (X, Y, _) = int(g['batch']), cfg.bev_xyz
    lift = oc.voxel_lift(g['ht_feat'], p)
    ch = g['channel_feat']
    logit = g['bev_mask_logit']
    geom = oc.bev_geom_attention(ch, logit, p) * ch
    p_dca = {k.replace('defor_cross_attention.', 'dca.'): v for k, v in p.items()}
    p_v2b = {k.replace('OpacityVoxelToBEV.', 'v2b.'): v for k, v in p.items()}
    oas = []
    for b in range(B):
        op = oc.gauss_heads(lift[b].reshape(-1, cfg.channels), np.zeros((lift[b].size // cfg.channels, 3), np.float32), p)[0]
        alpha_lidar = g['alpha_lidar'][b].reshape(1, cfg.num_height, Y, X)
        up = ohoa.interpolate_bilinear_ac(op.reshape(1, cfg.num_height, Y, X), (Y // 6, X // 6))
        close(up, g['opacity_up'][b:b + 1], 1e-5, 'opacity_up')
        close(ohoa.interpolate_bilinear_ac(alpha_lidar, (Y // 6, X // 6)), g['alpha_up'][b:b + 1], 1e-5, 'alpha_up')
        oas.append(ohoa.hoa1(op, alpha_lidar, p_dca, cfg.num_height, Y, X)[0])
    pos1 = oc.learned_positional_encoding(p, 'positional_encoding1', B, Y, X)
    view = ohoa.opacity_voxel_to_bev(np.concatenate(oas, 0), pos1, p_v2b)
    close(view, g['opacity_alpha_view'], 1e-4, 'opacity_alpha_view')
    mask = ohoa.opacity_mask(geom, view, {'mask.conv.weight': p['ObatinOpacityMask.conv.weight']}, 'mask')
    close(geom * mask, g['bev_feat'], 1e-4, 'bev_feat')
