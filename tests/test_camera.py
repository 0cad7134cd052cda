"""Camera set-up of the render call: ``gaussian_renderer.camera_from_calibration``
against the vectors dumped from the reference's own code (view_transformer_ocrf.py:1135-1152 with
MVSGaussian/lib/utils/data_utils.py:703-733) — tests/golden/camera.npz (6 cameras of the 256x704 rig) and the
``cam_*`` entries of tests/golden/core_small.npz (the cameras the reference's whole forward handed to ``render``).
Host arithmetic (numpy float64 / torch float32), so the bar is bit-exact."""
import math

import numpy as np
import torch

from ocrfdet_amd import gaussian_renderer as gr
from ocrfdet_amd import synthetic
from tests import helpers


def test_camera_from_calibration_matches_reference_vectors(golden):
    g = golden('camera.npz')
    H, W = 256, 704
    for cam in range(6):
        d = gr.camera_from_calibration(g[f'cam{cam}_K'], g[f'cam{cam}_c2w'], H, W)
        np.testing.assert_array_equal(d['world_view_transform'].numpy(), g[f'cam{cam}_world_view'])
        np.testing.assert_array_equal(d['full_proj_transform'].numpy(), g[f'cam{cam}_full_proj'])
        np.testing.assert_array_equal(d['camera_center'].numpy(), g[f'cam{cam}_center'])
        fov = g[f'cam{cam}_fov']
        assert float(d['FovX']) == fov[0] and float(d['FovY']) == fov[1]
        # the projection itself (transposed, as the reference stores it)
        proj = gr.getProjectionMatrix(0.01, 999.9, g[f'cam{cam}_K'], H, W).transpose(0, 1)
        np.testing.assert_array_equal(proj.numpy(), g[f'cam{cam}_projection'])
        assert d['height'] == H and d['width'] == W


def test_camera_of_the_whole_forward_fixture():
    """The cameras the reference's forward built for its two samples (random camera per sample,
    view_transformer_ocrf.py:1081) == ours from the same calibration."""
    cfg, g, _ = helpers.core_fixture()
    B = int(g['batch'])
    r = synthetic.rig(cfg.n_cams, cfg.input_size, B)
    H, W = cfg.input_size
    for n, cam in enumerate(int(c) for c in g['cam_idx_list']):
        d = gr.camera_from_calibration(r['intrins'][n, cam], r['c2w'][n, cam], H, W)
        np.testing.assert_array_equal(d['world_view_transform'].numpy(), g[f'cam_world_view{n}'])
        np.testing.assert_array_equal(d['full_proj_transform'].numpy(), g[f'cam_full_proj{n}'])
        fov = g[f'cam_fov{n}']
        assert float(d['FovX']) == fov[0] and float(d['FovY']) == fov[1]
        assert math.tan(float(d['FovX']) * 0.5) > 0
