"""CPU: index preparation — numpy oracle (oracle/index_prep.py) AND the product's torch code
(ocrfdet_amd/index_prep.py, device-agnostic) against golden vectors dumped from the reference's
own Python (tests/golden/make_golden.py).  Voxel indices must be bit-exact: rank triples are
compared in canonical order (the reference's argsort is unstable), intervals directly."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import index_prep as oip
from ocrfdet_amd import index_prep as pip_
from ocrfdet_amd import synthetic

CASES = [('cfg0', 'cfg0_1cam_128x352_bev64x64x4', True), ('ref', 'ref_6cam_256x704_bev128x128x1', False),
         ('cfg1', 'cfg1_6cam_256x704_bev128x128x8', False), ('cfg2', 'cfg2_6cam_2frame_bev200x200_render_hoa', False),
         ('cfg4', 'cfg4_6cam_8frame_512x1408_bev200x200', False)]      # BASELINE configs[4]: 512x1408 -> 32x88 features


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def _rig_t(cfg):
    r = synthetic.rig(cfg.n_cams, cfg.input_size, cfg.batch)
    return r, [torch.from_numpy(r[k]) for k in ('rots', 'trans', 'intrins', 'post_rots', 'post_trans', 'bda')]


@pytest.mark.parametrize('tag,key,full', CASES)
def test_lss_product_matches_reference_vectors(golden, tag, key, full):
    cfg = synthetic.CONFIGS[key]
    g = golden(f'lss_{tag}.npz')
    _, args = _rig_t(cfg)
    frustum = pip_.create_frustum(cfg.grid['depth'], cfg.input_size, cfg.downsample)
    assert (sha(frustum.numpy()) == g['frustum_sha256']).all()
    coor = pip_.get_lidar_coor(frustum, *args)
    assert (sha(coor.numpy()) == g['coor_sha256']).all()
    lower, interval, size = pip_.grid_infos(cfg.grid)
    np.testing.assert_array_equal(lower.numpy(), g['grid_lower_bound'])
    np.testing.assert_array_equal(size.numpy(), g['grid_size'])
    rb, rd, rf, st, ln = pip_.voxel_pooling_prepare_v2(coor, lower, interval, size)
    assert rb.numel() == int(g['n_points'])
    tri = oip.canonical_triples(rb.numpy(), rd.numpy(), rf.numpy())
    assert (sha(tri) == g['triples_sha256']).all()
    np.testing.assert_array_equal(st.numpy(), g['interval_starts'])
    np.testing.assert_array_equal(ln.numpy(), g['interval_lengths'])
    if full:
        np.testing.assert_array_equal(tri, g['triples'])
        np.testing.assert_array_equal(coor.numpy(), g['coor'])


@pytest.mark.parametrize('tag,key,full', CASES)
def test_lss_oracle_matches_reference_vectors(golden, tag, key, full):
    cfg = synthetic.CONFIGS[key]
    g = golden(f'lss_{tag}.npz')
    r = synthetic.rig(cfg.n_cams, cfg.input_size, cfg.batch)
    # SID depth bins: torch's float32 exp/log vs numpy's differ by <= 1 ulp; x/y are exact
    fr = oip.create_frustum(cfg.grid['depth'], cfg.input_size, cfg.downsample)
    gd = g['frustum'][:, 0, 0, 2] if full else g['frustum_d']
    np.testing.assert_allclose(fr[:, 0, 0, 2], gd, rtol=2.4e-7, atol=0)
    gx = g['frustum'][0, 0, :, 0] if full else g['frustum_x']
    gy = g['frustum'][0, :, 0, 1] if full else g['frustum_y']
    np.testing.assert_array_equal(fr[0, 0, :, 0], gx)
    np.testing.assert_array_equal(fr[0, :, 0, 1], gy)
    # per-point path: bit-exact given the reference's depth bins and 3x3 products
    fr = oip.create_frustum(cfg.grid['depth'], cfg.input_size, cfg.downsample, depth_bins=gd)
    coor = oip.get_lidar_coor(fr, r['rots'], r['trans'], r['intrins'], r['post_rots'], r['post_trans'],
                              r['bda'], g['inv_post_rots'], g['combine'])
    assert (sha(coor) == g['coor_sha256']).all()
    rb, rd, rf, st, ln = oip.voxel_pooling_prepare_v2(coor, g['grid_lower_bound'], g['grid_interval'], g['grid_size'])
    assert (sha(oip.canonical_triples(rb, rd, rf)) == g['triples_sha256']).all()
    np.testing.assert_array_equal(st, g['interval_starts'])
    np.testing.assert_array_equal(ln, g['interval_lengths'])
    # the oracle's own LAPACK 3x3 products agree with torch's to a few ulp
    inv_k = np.linalg.inv(r['intrins'].astype(np.float32))
    np.testing.assert_allclose(np.matmul(r['rots'], inv_k), g['combine'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('tag,key,full', CASES)
def test_ht_product_matches_reference_vectors(golden, tag, key, full):
    cfg = synthetic.CONFIGS[key]
    g = golden(f'ht_{tag}.npz')
    _, args = _rig_t(cfg)
    X, Y, _ = cfg.bev_xyz
    lidar2img, img_aug, _, _ = pip_.get_projection(*args)
    np.testing.assert_array_equal(lidar2img.numpy(), g['lidar2img'])
    np.testing.assert_array_equal(img_aug.numpy(), g['img_aug'])
    ref = pip_.get_reference_points_3d(Y, X, bs=cfg.batch, num_points_in_pillar=cfg.num_height, device='cpu')
    coor, mask, _ = pip_.get_sampling_point(ref, list(cfg.pc_range), cfg.grid['depth'], lidar2img, img_aug, cfg.input_size)
    assert (sha(ref.numpy()) == g['voxel_sha256']).all()      # scaled in place to metres
    assert (sha(coor.numpy()) == g['coor_sha256']).all()
    assert (sha(mask.numpy()) == g['mask_sha256']).all()
    assert int(mask.sum()) == int(g['n_mask'])
    Hf, Wf = cfg.feat_hw
    rb, rd, rf, st, ln = pip_.fast_sample_prepare(coor, mask, Wf, Hf, cfg.D)
    assert (sha(oip.canonical_triples(rb.numpy(), rd.numpy(), rf.numpy())) == g['triples_sha256']).all()
    np.testing.assert_array_equal(st.numpy(), g['interval_starts'])
    np.testing.assert_array_equal(ln.numpy(), g['interval_lengths'])


@pytest.mark.parametrize('tag,key,full', CASES)
def test_ht_oracle_matches_reference_vectors(golden, tag, key, full):
    cfg = synthetic.CONFIGS[key]
    g = golden(f'ht_{tag}.npz')
    r = synthetic.rig(cfg.n_cams, cfg.input_size, cfg.batch)
    X, Y, _ = cfg.bev_xyz
    ref = oip.get_reference_points_3d(Y, X, bs=cfg.batch, num_points_in_pillar=cfg.num_height)
    if full:
        np.testing.assert_array_equal(ref, g['ref_norm'])
    else:
        np.testing.assert_array_equal(ref[0, :, 0, 2], g['ref_norm_z'])
        np.testing.assert_array_equal(ref[0, 0, :X, 0], g['ref_norm_x'])
    l2i, aug = oip.get_projection(r['rots'], r['trans'], r['intrins'], r['post_rots'], r['post_trans'], r['bda'])
    np.testing.assert_allclose(l2i, g['lidar2img'], rtol=1e-5, atol=2e-4)     # LAPACK vs torch.inverse
    np.testing.assert_array_equal(aug, g['img_aug'])
    coor, mask, voxel = oip.get_sampling_point(ref, cfg.pc_range, cfg.grid['depth'], g['lidar2img'], g['img_aug'], cfg.input_size)
    assert (sha(voxel) == g['voxel_sha256']).all()
    assert (sha(coor) == g['coor_sha256']).all()
    assert (sha(mask) == g['mask_sha256']).all()
    Hf, Wf = cfg.feat_hw
    rb, rd, rf, st, ln = oip.fast_sample_prepare(coor, mask, Wf, Hf, cfg.D)
    assert (sha(oip.canonical_triples(rb, rd, rf)) == g['triples_sha256']).all()
    np.testing.assert_array_equal(st, g['interval_starts'])
    np.testing.assert_array_equal(ln, g['interval_lengths'])
