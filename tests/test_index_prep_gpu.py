"""GPU: the fused HIP index preparation (C ABI ocrf_lss_prepare / ocrf_ht_prepare through
ocrfdet_amd.index_prep.*_hip) against the golden vectors dumped from the reference's own Python
(tests/golden/make_golden.py) — voxel indices bit-exact: rank triples in canonical order, intervals
directly — and against the product's torch formulation element for element (the HIP sort is stable,
like torch.sort(stable=True) there)."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import index_prep as oip
from ocrfdet_amd import index_prep as pip_
from ocrfdet_amd import synthetic

pytestmark = pytest.mark.gpu

CASES = [('cfg0', 'cfg0_1cam_128x352_bev64x64x4'), ('ref', 'ref_6cam_256x704_bev128x128x1'),
         ('cfg1', 'cfg1_6cam_256x704_bev128x128x8'), ('cfg2', 'cfg2_6cam_2frame_bev200x200_render_hoa'),
         ('cfg4', 'cfg4_6cam_8frame_512x1408_bev200x200')]       # BASELINE configs[4]: 512x1408 -> 32x88 features


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def _rig_t(cfg):
    r = synthetic.rig(cfg.n_cams, cfg.input_size, cfg.batch)
    return [torch.from_numpy(r[k]) for k in ('rots', 'trans', 'intrins', 'post_rots', 'post_trans', 'bda')]


@pytest.mark.parametrize('tag,key', CASES)
def test_lss_prepare_hip_matches_reference_vectors(cuda, golden, tag, key):
    cfg = synthetic.CONFIGS[key]
    g = golden(f'lss_{tag}.npz')
    args = _rig_t(cfg)                                    # calibration on the host: the 3x3 algebra is LAPACK's
    frustum = pip_.create_frustum(cfg.grid['depth'], cfg.input_size, cfg.downsample)
    lower, interval, size = pip_.grid_infos(cfg.grid)
    block = pip_.lss_camera_block(*args)
    B, N = args[1].shape[:2]
    rb, rd, rf, st, ln = pip_.voxel_pooling_prepare_v2_hip(frustum.to(cuda), block.to(cuda), B, N, lower, interval, size)
    torch.cuda.synchronize()
    assert rb.numel() == int(g['n_points'])
    tri = oip.canonical_triples(rb.cpu().numpy(), rd.cpu().numpy(), rf.cpu().numpy())
    assert (sha(tri) == g['triples_sha256']).all()
    np.testing.assert_array_equal(st.cpu().numpy(), g['interval_starts'])
    np.testing.assert_array_equal(ln.cpu().numpy(), g['interval_lengths'])
    # element for element against the torch formulation (stable order inside a voxel)
    coor = pip_.get_lidar_coor(frustum, *args)
    want = pip_.voxel_pooling_prepare_v2(coor, lower, interval, size)
    for got, w in zip((rb, rd, rf, st, ln), want):
        np.testing.assert_array_equal(got.cpu().numpy(), w.numpy())


@pytest.mark.parametrize('tag,key', CASES)
def test_ht_prepare_hip_matches_reference_vectors(cuda, golden, tag, key):
    cfg = synthetic.CONFIGS[key]
    g = golden(f'ht_{tag}.npz')
    args = _rig_t(cfg)
    X, Y, _ = cfg.bev_xyz
    Hf, Wf = cfg.feat_hw
    lidar2img, img_aug, _, _ = pip_.get_projection(*args)
    B, N = lidar2img.shape[:2]
    template = pip_.get_reference_points_3d(Y, X, bs=1, num_points_in_pillar=cfg.num_height, device='cpu')[0]
    block = pip_.ht_camera_block(lidar2img, img_aug)
    rb, rd, rf, st, ln = pip_.fast_sample_prepare_hip(template.to(cuda), block.to(cuda), B, N, list(cfg.pc_range),
                                                      cfg.input_size, cfg.grid['depth'], Wf, Hf, cfg.D)
    torch.cuda.synchronize()
    assert rb.numel() == int(g['n_mask'])
    tri = oip.canonical_triples(rb.cpu().numpy(), rd.cpu().numpy(), rf.cpu().numpy())
    assert (sha(tri) == g['triples_sha256']).all()
    np.testing.assert_array_equal(st.cpu().numpy(), g['interval_starts'])
    np.testing.assert_array_equal(ln.cpu().numpy(), g['interval_lengths'])
    ref = pip_.get_reference_points_3d(Y, X, bs=cfg.batch, num_points_in_pillar=cfg.num_height, device='cpu')
    coor, mask, _ = pip_.get_sampling_point(ref, list(cfg.pc_range), cfg.grid['depth'], lidar2img, img_aug, cfg.input_size)
    want = pip_.fast_sample_prepare(coor, mask, Wf, Hf, cfg.D)
    for got, w in zip((rb, rd, rf, st, ln), want):
        np.testing.assert_array_equal(got.cpu().numpy(), w.numpy())


def test_lss_prepare_hip_three_radix_passes_and_empty(cuda):
    """A grid of more than 2^18 voxels takes the third radix pass; a grid nothing falls into
    returns five None like the reference (view_transformer.py:238-244)."""
    cfg = synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4']
    args = _rig_t(cfg)
    frustum = pip_.create_frustum(cfg.grid['depth'], cfg.input_size, cfg.downsample)
    block = pip_.lss_camera_block(*args)
    B, N = args[1].shape[:2]
    grid = dict(x=[-51.2, 51.2, 0.1], y=[-51.2, 51.2, 0.1], z=[-5.0, 3.0, 8.0])      # 1024 x 1024 x 1 = 2^20 voxels
    lower, interval, size = pip_.grid_infos(grid)
    got = pip_.voxel_pooling_prepare_v2_hip(frustum.to(cuda), block.to(cuda), B, N, lower, interval, size)
    want = pip_.voxel_pooling_prepare_v2(pip_.get_lidar_coor(frustum, *args), lower, interval, size)
    for a, w in zip(got, want):
        np.testing.assert_array_equal(a.cpu().numpy(), w.numpy())
    far = dict(x=[1000.0, 1010.0, 1.0], y=[1000.0, 1010.0, 1.0], z=[-5.0, 3.0, 8.0])
    lower, interval, size = pip_.grid_infos(far)
    assert pip_.voxel_pooling_prepare_v2_hip(frustum.to(cuda), block.to(cuda), B, N, lower, interval, size) == (None,) * 5


def test_prepared_ranks_feed_the_pool(cuda, oracle_lib):
    """End to end: HIP-prepared ranks -> HIP pool == oracle pool on the torch-prepared ranks."""
    from ocrfdet_amd import bevpool
    cfg = synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4']
    args = _rig_t(cfg)
    frustum = pip_.create_frustum(cfg.grid['depth'], cfg.input_size, cfg.downsample)
    lower, interval, size = pip_.grid_infos(cfg.grid)
    B, N = args[1].shape[:2]
    rb, rd, rf, st, ln = pip_.voxel_pooling_prepare_v2_hip(frustum.to(cuda), pip_.lss_camera_block(*args).to(cuda), B, N,
                                                           lower, interval, size)
    depth, feat = synthetic.depth_and_feat(cfg, seed=3)        # (B*N,D,H,W), (B*N,C,H,W)
    Hf, Wf = cfg.feat_hw
    depth = depth.numpy().reshape(B, N, cfg.D, Hf, Wf)
    feat = np.ascontiguousarray(feat.numpy().reshape(B, N, cfg.channels, Hf, Wf).transpose(0, 1, 3, 4, 2))   # channels-last
    X, Y, Z = cfg.bev_xyz
    shape = (B, Z, Y, X, cfg.channels)
    got = bevpool.bev_pool_v2(torch.from_numpy(depth).to(cuda), torch.from_numpy(feat).to(cuda), rd, rf, rb, shape, st, ln)
    want = oracle_lib.bev_pool_v2(depth, feat, rd.cpu().numpy(), rf.cpu().numpy(), rb.cpu().numpy(), shape,
                                  st.cpu().numpy(), ln.cpu().numpy())
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-4)


@pytest.mark.parametrize('key', ['cfg0_1cam_128x352_bev64x64x4', 'cfg2_6cam_2frame_bev200x200_render_hoa'])
def test_per_step_path_without_host_reads_equals_cached_path(cuda, key):
    """HotPath(index_prep_mode='per_step'): HIP index preparation + pooling on capacity-sized
    vectors with device-side counts gives bitwise the pooled BEVs of the cached-rank path."""
    from ocrfdet_amd import hotpath
    cfg = synthetic.CONFIGS[key]
    cfg = synthetic.PathConfig(**{**cfg.__dict__, 'render': False, 'hoa': False})
    a = hotpath.HotPath(cfg, cuda, lss_pool_backend='tile', ht_pool_backend='tile')      # the per-step path pools with the tile kernel
    b = hotpath.HotPath(cfg, cuda, index_prep_mode='per_step')
    depth, feat = a.make_inputs(seed=1)
    want = a.step(depth, feat)
    got = b.step(depth, feat)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_per_step_path_on_two_preparation_streams_equals_cached_path(cuda):
    """The whole cfg2 step with the index preparation inside AND the calibration algebra on the device: the two
    preparations run on a stream each beside HOA-1/2, every pooling waits for its own ranks only — pooled BEVs, HOA
    outputs and renders equal the single-stream step's bit for bit, step after step (the look-back scratch is per
    stream, the buffers are reused)."""
    from ocrfdet_amd import hotpath
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    # the same step on ONE stream (device geometry may move a borderline HT sample by one cell against the host
    # algebra, tests/test_device_geometry_gpu.py: the reference here is the device-geometry step itself)
    a = hotpath.HotPath(cfg, cuda, index_prep_mode='per_step', device_geometry=True, overlap=False)
    b = hotpath.HotPath(cfg, cuda, index_prep_mode='per_step', device_geometry=True)
    assert hasattr(b, '_calib_dev')
    depth, feat = a.make_inputs(seed=3)
    want = a.step(depth, feat)
    torch.cuda.synchronize()
    for _ in range(3):
        got = b.step(depth, feat)
        torch.cuda.synchronize()
        assert b._prep_stream2 is not None
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])                 # LSS, HT
        for g, w in zip(got[2], want[2]):                                                    # renders per frame
            assert torch.equal(g['color'], w['color']) and torch.equal(g['depth'], w['depth'])
        assert torch.equal(got[3], want[3]) and torch.equal(got[4], want[4])                 # gated BEV, opacity BEV


SWEEP = [  # (B, N, input H, W, BEV Y, X, z cells, heights): one-tile grids, ragged tiles, several scan tiles, empty cameras
    (1, 1, 32, 48, 4, 4, 1, 5), (1, 2, 64, 96, 9, 17, 2, 6), (2, 3, 64, 176, 33, 31, 1, 13), (3, 1, 128, 160, 64, 64, 4, 7),
    (1, 6, 96, 128, 130, 70, 1, 13), (2, 2, 48, 64, 257, 3, 1, 8)]


@pytest.mark.parametrize('B,N,H,W,Y,X,Zc,Zh', SWEEP)
def test_prepare_hip_equals_the_torch_formulation_on_a_shape_sweep(cuda, B, N, H, W, Y, X, Zc, Zh):
    """Edge sizes of the single-launch prefix sums and of the interval kernel (fewer voxels than a workgroup, ragged last
    tiles, several look-back tiles, cameras that see nothing, samples with no points): rank vectors and intervals of
    both preparations, element for element, against the CPU torch formulation the reference's vectors pin."""
    rng = np.random.default_rng(B * 1000 + N * 100 + Y)
    r = synthetic.rig(N, (H, W), B)
    r['trans'] = (r['trans'] + rng.uniform(-0.3, 0.3, r['trans'].shape)).astype(np.float32)
    args = [torch.from_numpy(r[k]) for k in ('rots', 'trans', 'intrins', 'post_rots', 'post_trans', 'bda')]
    half_x, half_y = 0.4 * X, 0.4 * Y
    grid = dict(x=[-half_x, half_x, 0.8], y=[-half_y, half_y, 0.8], z=[-5.0, 3.0, 8.0 / Zc], depth=[1.0, 45.0, 1.0])
    pc_range = [-half_x, -half_y, -5.0, half_x, half_y, 3.0]
    frustum = pip_.create_frustum(grid['depth'], (H, W), 16)
    lower, interval, size = pip_.grid_infos(grid)
    D = frustum.shape[0]
    Hf, Wf = H // 16, W // 16
    block = pip_.lss_camera_block(*args)
    got = pip_.voxel_pooling_prepare_v2_hip(frustum.to(cuda), block.to(cuda), B, N, lower, interval, size)
    coor = pip_.get_lidar_coor(frustum, *args)
    want = pip_.voxel_pooling_prepare_v2(coor, lower, interval, size)
    for name, g_, w_ in zip(('ranks_bev', 'ranks_depth', 'ranks_feat', 'starts', 'lengths'), got, want):
        if w_ is None:
            assert g_ is None or g_.numel() == 0, f'lss {name}: expected nothing'
        else:
            np.testing.assert_array_equal(g_.cpu().numpy(), w_.numpy(), err_msg=f'lss {name}')
    lidar2img, img_aug, _, _ = pip_.get_projection(*args)
    template = pip_.get_reference_points_3d(Y, X, bs=1, num_points_in_pillar=Zh, device='cpu')[0]
    got = pip_.fast_sample_prepare_hip(template.to(cuda), pip_.ht_camera_block(lidar2img, img_aug).to(cuda), B, N, pc_range,
                                       (H, W), grid['depth'], Wf, Hf, D)
    ref = pip_.get_reference_points_3d(Y, X, bs=B, num_points_in_pillar=Zh, device='cpu')
    c, m, _ = pip_.get_sampling_point(ref, pc_range, grid['depth'], lidar2img, img_aug, (H, W))
    want = pip_.fast_sample_prepare(c, m, Wf, Hf, D)
    for name, g_, w_ in zip(('ranks_bev', 'ranks_depth', 'ranks_feat', 'starts', 'lengths'), got, want):
        if w_ is None:
            assert g_ is None or g_.numel() == 0, f'ht {name}: expected nothing'
        else:
            np.testing.assert_array_equal(g_.cpu().numpy(), w_.numpy(), err_msg=f'ht {name}')
