"""HOA modules of ocrfdet_amd.hoa: state_dict compatibility with the reference (strict load of the
reference's own weights from tests/golden/hoa.npz) on the CPU, numerics on the GPU (HIP kernels
through the C ABI) against the reference's outputs and the numpy oracle."""
import numpy as np
import pytest
import torch

from ocrfdet_amd import hoa


def _sd(g, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(prefix + '.')}


@pytest.fixture(scope='module')
def g(golden):
    return dict(golden('hoa.npz'))


def test_state_dicts_load_strictly(g):
    hoa.ObatinOpacityMask().load_state_dict(_sd(g, 'mask'), strict=True)
    for ch in (4, 8, 16):
        hoa.HeightAttention(ch, ch, 1).load_state_dict(_sd(g, f'ha{ch}'), strict=True)
    hoa.OpacityVoxelToBEVConverter(13).load_state_dict(_sd(g, 'v2b'), strict=True)
    hoa.DeformableAttention2D(dim=13, dim_head=8, heads=1, dropout=0.1, downsample_factor=4, offset_scale=4,
                              offset_groups=None, offset_kernel_size=6).load_state_dict(_sd(g, 'dca'), strict=True)


def test_deformable_attention_matches_reference_on_cpu(g):
    m = hoa.DeformableAttention2D(dim=13, dim_head=8, heads=1, dropout=0.1, downsample_factor=4, offset_scale=4,
                                  offset_groups=None, offset_kernel_size=6).eval()
    m.load_state_dict(_sd(g, 'dca'))
    with torch.no_grad():
        out = m(torch.from_numpy(g['dca_q']), torch.from_numpy(g['dca_kv']))
        full = hoa.hoa1(m, torch.from_numpy(g['hoa1_opacity'].astype(np.float32)),
                        torch.from_numpy(g['hoa1_alpha'].astype(np.float32)), 13, 128, 128)
    np.testing.assert_allclose(out.numpy(), g['dca_out'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(full[:, :, ::5, ::3].numpy(), g['hoa1_out_slice'], rtol=1e-5, atol=1e-6)


def test_cpu_tensors_are_refused(g):
    from ocrfdet_amd import _lib
    with pytest.raises(_lib.OcrfHipError):
        hoa.ObatinOpacityMask()(torch.zeros(1, 8, 4, 4), torch.zeros(1, 1, 4, 4))


@pytest.mark.gpu
def test_opacity_mask_gate_gpu(cuda, g):
    from oracle import hoa as ohoa
    m = hoa.ObatinOpacityMask().to(cuda)
    m.load_state_dict(_sd(g, 'mask'))
    x, ob = torch.from_numpy(g['mask_in_x']).to(cuda), torch.from_numpy(g['mask_in_opacity']).to(cuda)
    with torch.no_grad():                                                                         # the HIP kernels
        mask, gated = m.gate(x, ob)
        plain = m(x, ob)
    np.testing.assert_allclose(mask.cpu().numpy(), g['mask_out'], rtol=1e-4, atol=1e-5)          # reference
    np.testing.assert_allclose(gated.cpu().numpy(), g['mask_gated'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(plain.cpu().numpy(), ohoa.opacity_mask(g['mask_in_x'], g['mask_in_opacity'], g, 'mask'),
                               rtol=1e-4, atol=1e-5)                                             # oracle
    # full BEV size of the headline config, ragged against the 256-pixel workgroups
    rng = np.random.default_rng(0)
    xb = rng.standard_normal((2, 80, 200, 200)).astype(np.float32)
    obb = rng.standard_normal((2, 1, 200, 200)).astype(np.float32)
    want = ohoa.opacity_mask(xb, obb, g, 'mask')
    with torch.no_grad():
        mask, gated = m.gate(torch.from_numpy(xb).to(cuda), torch.from_numpy(obb).to(cuda))
    np.testing.assert_allclose(mask.cpu().numpy(), want, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gated.cpu().numpy(), xb * want, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_channel_statistics_as_an_op_of_their_own(cuda, g):
    """``hoa.channel_stats`` (the first of the gate's two kernels) handed to ``gate(..., stats=...)``: the same bits as the
    gate computing them itself; a mis-shaped ``stats`` is refused."""
    torch.manual_seed(3)
    m = hoa.ObatinOpacityMask().to(cuda)
    x = torch.randn(2, 80, 40, 56, device=cuda)
    ob = torch.randn(2, 1, 40, 56, device=cuda)
    with torch.no_grad():
        mask, gated = m.gate(x, ob)
        st = hoa.channel_stats(x)
        torch.testing.assert_close(st[:, 0], x.mean(1), rtol=1e-5, atol=1e-6)
        assert torch.equal(st[:, 1], x.amax(1))
        mask2, gated2 = m.gate(x, ob, stats=st)
    assert torch.equal(mask, mask2) and torch.equal(gated, gated2)
    with pytest.raises(Exception), torch.no_grad():
        m.gate(x, ob, stats=st[:, :1])


@pytest.mark.gpu
def test_height_attention_gpu(cuda, g):
    for ch in (4, 8, 16):
        m = hoa.HeightAttention(ch, ch, 1).to(cuda)
        m.load_state_dict(_sd(g, f'ha{ch}'))
        x = torch.from_numpy(g[f'ha{ch}_in']).to(cuda)
        with torch.no_grad():
            gate, gated = m(x), m.gate_apply(x)
        assert gate.shape == (2, ch, 1, 1)
        np.testing.assert_allclose(gate.cpu().numpy(), g[f'ha{ch}_out'], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(gated.cpu().numpy(), g[f'ha{ch}_out'] * g[f'ha{ch}_in'], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_opacity_voxel_to_bev_gpu(cuda, g):
    m = hoa.OpacityVoxelToBEVConverter(13).to(cuda).eval()
    m.load_state_dict(_sd(g, 'v2b'))
    with torch.no_grad():
        out = m(torch.from_numpy(g['v2b_in']).to(cuda), torch.from_numpy(g['v2b_pos']).to(cuda))
    np.testing.assert_allclose(out.cpu().numpy(), g['v2b_out'], rtol=1e-4, atol=3e-5)


@pytest.mark.gpu
def test_opacity_voxel_to_bev_fused_vs_blockwise_and_oracle(cuda, g):
    """Eval-mode fused path (5 block kernels) == training-style block-by-block path == numpy oracle,
    at the headline BEV size (200x200: ragged 16x16 tiles, H/4 = 50)."""
    from oracle import hoa as ohoa
    m = hoa.OpacityVoxelToBEVConverter(13).to(cuda).eval()
    m.load_state_dict(_sd(g, 'v2b'))
    rng = np.random.default_rng(1)
    x = rng.random((2, 13, 200, 200), dtype=np.float32)
    pos = (rng.standard_normal((2, 4, 200, 200)) * 0.1).astype(np.float32)
    xt, pt = torch.from_numpy(x).to(cuda), torch.from_numpy(pos).to(cuda)
    with torch.no_grad():
        fused = m(xt, pt)
        enc1 = m.ca1.gate_apply(m.encoder1(xt) + pt)                       # block-by-block (torch convs)
        enc2 = m.ca2.gate_apply(m.encoder2(m.pool(enc1)))
        mid = m.ca_bottleneck.gate_apply(m.bottleneck(m.pool(enc2)))
        dec2 = m.ca_dec2.gate_apply(m.decoder2(torch.cat((m.upconv2(mid), enc2), dim=1)))
        dec1 = m.ca_dec1.gate_apply(m.decoder1(torch.cat((m.upconv1(dec2), enc1), dim=1)))
        blockwise = m.output_conv(dec1)
    np.testing.assert_allclose(fused.cpu().numpy(), blockwise.cpu().numpy(), rtol=1e-4, atol=3e-5)
    want = ohoa.opacity_voxel_to_bev(x, pos, g, 'v2b')
    np.testing.assert_allclose(fused.cpu().numpy(), want, rtol=1e-4, atol=3e-5)
    # the one-call form (six launches, gates computed in the consumers' prologues) == the eleven-launch form (a
    # gate kernel between every two blocks), bit for bit: same reductions, same arithmetic
    with torch.no_grad():
        eleven = m._forward_blocks(xt, pt)
        again = m(xt, pt)
    assert torch.equal(fused, eleven) and torch.equal(fused, again)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W', [(1, 16, 16), (1, 36, 52), (3, 100, 60), (2, 128, 128), (1, 44, 236), (1, 264, 260)])
def test_opacity_voxel_to_bev_six_launch_form_on_other_map_sizes(cuda, g, B, H, W):
    """csrc/hoa_v2b.hip (16 x 4 pixel tiles, gates rebuilt from per-tile maxima, DPP reductions) against the block-wise
    eleven-launch form (16 x 16 tiles, gate kernels in between), bit for bit, on maps whose three levels end in ragged
    tiles in x, in y, in both, on the smallest map a tile covers and on one beyond 256 x 256 (more than four rounds of
    tile maxima per thread)."""
    m = hoa.OpacityVoxelToBEVConverter(13).to(cuda).eval()
    m.load_state_dict(_sd(g, 'v2b'))
    rng = np.random.default_rng(H * 1000 + W)
    xt = torch.from_numpy(rng.random((B, 13, H, W), dtype=np.float32)).to(cuda)
    pt = torch.from_numpy((rng.standard_normal((B, 4, H, W)) * 0.1).astype(np.float32)).to(cuda)
    with torch.no_grad():
        six = m(xt, pt)
        eleven = m._forward_blocks(xt, pt)
    assert torch.equal(six, eleven)


@pytest.mark.gpu
def test_hoa1_gpu(cuda, g):
    m = hoa.DeformableAttention2D(dim=13, dim_head=8, heads=1, dropout=0.1, downsample_factor=4, offset_scale=4,
                                  offset_groups=None, offset_kernel_size=6).to(cuda).eval()
    m.load_state_dict(_sd(g, 'dca'))
    with torch.no_grad():
        full = hoa.hoa1(m, torch.from_numpy(g['hoa1_opacity'].astype(np.float32)).to(cuda),
                        torch.from_numpy(g['hoa1_alpha'].astype(np.float32)).to(cuda), 13, 128, 128)
    np.testing.assert_allclose(full[:, :, ::5, ::3].cpu().numpy(), g['hoa1_out_slice'], rtol=1e-4, atol=2e-5)
    # fused kernels (eval) == the reference's op sequence (train-mode dispatch, dropout p = 0) at 200x200, B = 2
    rng = np.random.default_rng(4)
    op = torch.from_numpy(rng.random((2, 13, 200, 200), dtype=np.float32)).to(cuda)
    al = torch.from_numpy(rng.random((2, 13, 200, 200), dtype=np.float32)).to(cuda)
    with torch.no_grad():
        fused = hoa.hoa1(m, op.reshape(-1, 1), al, 13, 200, 200)
        m.train()
        m.dropout.p = 0.0
        ref = hoa.hoa1(m, op.reshape(-1, 1), al, 13, 200, 200)
        m.eval()
    np.testing.assert_allclose(fused.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('B,Y,X', [(1, 36, 36), (1, 48, 132), (3, 100, 60), (2, 128, 200), (1, 236, 44)])
def test_hoa1_fused_kernels_on_other_map_sizes(cuda, g, B, Y, X):
    """The two-launch HOA-1 (kv tokens with rebuilt q rows; attention + upsample per 16x16 output tile with a window of
    at most 5 x 5 tokens) against the reference's op sequence on maps that are not square, not multiples of the tile,
    the smallest legal size (Y/6 = 6 tokens) and a single-column kv grid."""
    m = hoa.DeformableAttention2D(dim=13, dim_head=8, heads=1, dropout=0.1, downsample_factor=4, offset_scale=4,
                                  offset_groups=None, offset_kernel_size=6).to(cuda).eval()
    m.load_state_dict(_sd(g, 'dca'))
    rng = np.random.default_rng(Y * 1000 + X)
    op = torch.from_numpy(rng.random((B, 13, Y, X), dtype=np.float32)).to(cuda)
    al = torch.from_numpy(rng.random((B, 13, Y, X), dtype=np.float32)).to(cuda)
    with torch.no_grad():
        fused = hoa.hoa1(m, op.reshape(-1, 1), al, 13, Y, X)
        m.train()
        m.dropout.p = 0.0
        ref = hoa.hoa1(m, op.reshape(-1, 1), al, 13, Y, X)
        m.eval()
    np.testing.assert_allclose(fused.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
def test_hoa_modules_keep_gradients_under_autograd():
    """The HIP kernels are forward-only: with autograd recording, every HOA module must fall back to
    differentiable torch ops (a silent gradient cut would break training as a drop-in) and agree with
    its fused forward."""
    import torch
    from ocrfdet_amd import hoa
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    ha = hoa.HeightAttention(8, 8, 1).to(dev)
    x = torch.randn(2, 8, 12, 10, device=dev, requires_grad=True)
    y = ha.gate_apply(x)
    y.sum().backward()
    assert x.grad is not None and ha.conv1[0].weight.grad is not None and float(x.grad.abs().sum()) > 0
    with torch.no_grad():
        assert torch.allclose(ha.gate_apply(x), y, atol=1e-6)
    om = hoa.ObatinOpacityMask().to(dev)
    f = torch.randn(2, 16, 12, 10, device=dev, requires_grad=True)
    ob = torch.randn(2, 1, 12, 10, device=dev, requires_grad=True)
    mask, gated = om.gate(f, ob)
    gated.sum().backward()
    assert f.grad is not None and ob.grad is not None and om.conv.weight.grad is not None
    with torch.no_grad():
        m2, g2 = om.gate(f, ob)
    assert torch.allclose(m2, mask, atol=1e-6) and torch.allclose(g2, gated, atol=1e-5)
    v2b = hoa.OpacityVoxelToBEVConverter(13).to(dev).eval()
    xin = torch.randn(1, 13, 16, 16, device=dev, requires_grad=True)
    pos = torch.randn(1, 4, 16, 16, device=dev)
    out = v2b(xin, pos)
    out.sum().backward()
    assert xin.grad is not None and float(xin.grad.abs().sum()) > 0
    with torch.no_grad():
        assert torch.allclose(v2b(xin, pos), out, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 13, 50, 70), (1, 4, 17, 5), (3, 16, 33, 129)])
def test_depthwise_conv3x3_forward_and_backward(shape):
    """ocrf_hoa_dw3x3 / ocrf_hoa_dw3x3_wgrad (the training path of the UNet blocks) against F.conv2d's own
    forward and autograd gradients; ragged planes (not multiples of the 64x4 tile / 16-row band)."""
    import torch.nn.functional as F
    B, C, Y, X = shape
    torch.manual_seed(C)
    dev = torch.device('cuda:0')
    conv = torch.nn.Conv2d(C, C, 3, padding=1, groups=C).to(dev)
    x = torch.randn(B, C, Y, X, device=dev, requires_grad=True)
    gy = torch.randn(B, C, Y, X, device=dev)
    want = F.conv2d(x, conv.weight, conv.bias, padding=1, groups=C)
    gx_w, gw_w, gb_w = torch.autograd.grad(want, (x, conv.weight, conv.bias), gy)
    got = hoa._DepthwiseConv3x3.apply(x, conv.weight, conv.bias)
    gx, gw, gb = torch.autograd.grad(got, (x, conv.weight, conv.bias), gy)
    assert torch.allclose(got, want, atol=1e-5)
    assert torch.allclose(gx, gx_w, atol=1e-5)
    assert torch.allclose(gw, gw_w, atol=2e-4 * float(gw_w.abs().max()) + 1e-4), float((gw - gw_w).abs().max())
    assert torch.allclose(gb, gb_w, atol=2e-4 * float(gb_w.abs().max()) + 1e-4)
    # the block wrapper takes this path and matches the plain Sequential, training-mode BatchNorm included
    v2b = hoa.OpacityVoxelToBEVConverter(13).to(dev).train()
    xin = torch.randn(2, 13, 16, 24, device=dev, requires_grad=True)
    a = v2b._block(v2b.encoder1, xin)
    b = v2b.encoder1(xin)
    assert torch.allclose(a, b, atol=1e-5)
    ga, = torch.autograd.grad(a.square().sum(), xin)
    gb2, = torch.autograd.grad(b.square().sum(), xin)
    assert torch.allclose(ga, gb2, atol=1e-4)


@pytest.mark.gpu
def test_full_size_hoa_is_per_sample_and_reproducible(cuda):
    """HOA-1 / HOA-2 / HOA-3 at the headline BEV size (B = 2, 13 x 200 x 200; 80 x 200 x 200): size-independent
    properties, bit for bit — a sample's result does not depend on the other samples of the batch (the fused
    kernels run the whole batch in one launch where the reference loops), two runs agree, and HOA-3's gated output
    is exactly x * mask."""
    torch.manual_seed(0)
    B, Zh, Y, X = 2, 13, 200, 200
    dca = hoa.DeformableAttention2D(dim=13, dim_head=8, heads=1, dropout=0.1, downsample_factor=4, offset_scale=4,
                                    offset_groups=None, offset_kernel_size=6).to(cuda).eval()
    v2b = hoa.OpacityVoxelToBEVConverter(13).to(cuda).eval()
    om = hoa.ObatinOpacityMask().to(cuda).eval()
    opacity = torch.rand(B * Zh * Y * X, 1, device=cuda)
    alpha = torch.rand(B, Zh, Y, X, device=cuda)
    pos = torch.randn(B, 4, Y, X, device=cuda)
    feat = torch.randn(B, 80, Y, X, device=cuda)
    with torch.no_grad():
        oa = hoa.hoa1(dca, opacity, alpha, Zh, Y, X)
        view = v2b(oa, pos)
        mask, gated = om.gate(feat, view)
        assert torch.equal(hoa.hoa1(dca, opacity, alpha, Zh, Y, X), oa) and torch.equal(v2b(oa, pos), view)
        assert torch.equal(gated, feat * mask)
        assert torch.isfinite(oa).all() and torch.isfinite(view).all() and float(mask.min()) >= 0 and float(mask.max()) <= 1
        n = Zh * Y * X
        for b in range(B):
            oa_b = hoa.hoa1(dca, opacity[b * n:(b + 1) * n], alpha[b:b + 1], Zh, Y, X)
            assert torch.equal(oa_b, oa[b:b + 1]), 'HOA-1 depends on the rest of the batch'
            view_b = v2b(oa_b, pos[b:b + 1])
            assert torch.equal(view_b, view[b:b + 1]), 'HOA-2 depends on the rest of the batch'
            mb, gb = om.gate(feat[b:b + 1], view_b)
            assert torch.equal(mb, mask[b:b + 1]) and torch.equal(gb, gated[b:b + 1])
