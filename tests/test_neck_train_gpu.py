"""GPU parity of csrc/neck_train.hip: the backward of the fused voxel lift + four Gaussian heads against autograd through
the reference's layer formulation (VoxelFeatureExtractor = Conv3d(1 -> Zh, k = 1) + BatchNorm3d + ReLU,
view_transformer_ocrf.py:520-531; S/R/A/C_MLP :272-320) in float64.  Tolerance: 1e-4 of the largest entry of each gradient
(BASELINE.json north_star's bar for network outputs), stated where it is applied."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

REL = 1e-4


def _modules(zh, dev, C=80):
    from ocrfdet_amd import view_transformer_ocrf as vto
    vfe = vto.VoxelFeatureExtractor(1, zh).to(dev)
    bn = vfe.conv[1]
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.2), bn.running_var.uniform_(0.5, 1.5)
    heads = [cls(C, 4, o).to(dev) for cls, o in ((vto.ScaleFactorMLP, 3), (vto.RotationFactorMLP, 4),
                                                 (vto.OpacityFactorMLP, 1), (vto.ColorFactorMLPGaussian, 3))]
    return vfe, heads


def _layers(vfe, heads, bev, rgb):
    """The reference's op sequence (:1051, :1130-1133) on the materialised voxel feature, plain torch layers."""
    B, zh = bev.shape[0], vfe.conv[0].out_channels
    vf = vfe.conv(bev.permute(0, 2, 3, 1).unsqueeze(1)).reshape(B, zh * bev.shape[2] * bev.shape[3], -1)
    s, r, a, c = heads
    plain = lambda m, x: m.act(m.fc2(torch.relu(m.fc1(x))))                                   # noqa: E731
    return (plain(a, vf), plain(s, vf), plain(r, vf), plain(c, torch.cat((vf, rgb.reshape(B, -1, 3) / 255.0), -1)))


def _named_grads(vfe, heads):
    out = {}
    for tag, m in (('vfe', vfe), ('S', heads[0]), ('R', heads[1]), ('A', heads[2]), ('C', heads[3])):
        for n, p in m.named_parameters():
            out[f'{tag}.{n}'] = p.grad
    return out


def _check(got, want, what, scale=None):
    want = want.float()
    bar = REL * float((want if scale is None else scale).abs().max()) + 1e-9
    err = float((got - want).abs().max())
    assert err <= bar, f'{what}: |err| {err:.3e} > {bar:.3e} (1e-4 of the largest entry)'


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('zh', [1, 2, 4, 6, 8, 13])
def test_gauss_heads_backward_every_height_count(zh, training):
    """Every register-tile instantiation, a plane that is no multiple of the 64-pillar tile, batch statistics (training) and
    running statistics (eval): gradients of the BEV map and of every parameter of the five modules."""
    from ocrfdet_amd import neck_ops
    torch.manual_seed(100 + zh)
    dev = torch.device('cuda:0')
    vfe, heads = _modules(zh, dev)
    vfe.train(training)
    B, Y, X = 2, 5, 27
    bev = torch.randn(B, 80, Y, X, device=dev).mul_(1.5).requires_grad_(True)
    rgb = torch.rand(B, zh, Y * X, 3, device=dev) * 255
    wts = [torch.randn(B, zh * Y * X, k, device=dev) for k in (1, 3, 4, 3)]

    ref_vfe, ref_heads = copy.deepcopy(vfe).double(), [copy.deepcopy(h).double() for h in heads]
    bev64 = bev.detach().double().requires_grad_(True)
    outs64 = _layers(ref_vfe, ref_heads, bev64, rgb.double())
    sum((o * w.double()).sum() for o, w in zip(outs64, wts)).backward()

    stats_before = (vfe.conv[1].running_mean.clone(), vfe.conv[1].running_var.clone())
    prm = neck_ops.pack_gauss_head_params_autograd(vfe, *heads, x=bev)
    outs = neck_ops.gauss_heads_train(bev, rgb, prm, zh)
    for o, o64, name in zip(outs, outs64, ('opacity', 'scales', 'rotations', 'colour')):
        assert float((o.detach() - o64.detach().float()).abs().max()) <= 1e-5, name
    sum((o * w).sum() for o, w in zip(outs, wts)).backward()

    _check(bev.grad, bev64.grad, f'd bev (Zh={zh})')
    want = _named_grads(ref_vfe, ref_heads)
    for name, g in _named_grads(vfe, heads).items():
        assert g is not None, name
        # With batch statistics the lifted feature does not depend on the convolution's weight and bias (BatchNorm removes
        # any affine map of its input; only eps leaks through): their gradients are the residue of two terms that cancel,
        # each of the size of the BatchNorm weight's gradient — which is the scale their error is held to.
        cancels = training and name.startswith('vfe.conv.0')
        _check(g, want[name], f'd {name} (Zh={zh})', scale=want['vfe.conv.1.weight'] if cancels else None)
    bn, bn64 = vfe.conv[1], ref_vfe.conv[1]
    if training:            # the momentum update of the running statistics happened, once, as BatchNorm3d does it
        assert not torch.equal(bn.running_mean, stats_before[0])
        assert int(bn.num_batches_tracked) == int(bn64.num_batches_tracked) == 1
    torch.testing.assert_close(bn.running_mean, bn64.running_mean.float(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, bn64.running_var.float(), rtol=1e-5, atol=1e-6)


def test_gauss_heads_backward_unused_outputs_and_determinism():
    """Outputs that took no part in the loss arrive as ``None`` (a NULL pointer at the C ABI = a zero gradient); the same
    call twice gives the same bits (fixed-order sums, no atomics)."""
    from ocrfdet_amd import neck_ops
    torch.manual_seed(5)
    dev = torch.device('cuda:0')
    vfe, heads = _modules(13, dev)
    vfe.eval()
    bev = torch.randn(1, 80, 9, 31, device=dev)
    rgb = torch.rand(1, 13, 279, 3, device=dev) * 255
    prm = neck_ops.pack_gauss_head_params(vfe, *heads)
    g_op = torch.randn(1, 13 * 279, 1, device=dev)
    a = neck_ops.gauss_heads_backward(bev, rgb, prm, 13, g_opacity=g_op)
    b = neck_ops.gauss_heads_backward(bev, rgb, prm, 13, g_op, torch.zeros(1, 13 * 279, 3, device=dev),
                                      torch.zeros(1, 13 * 279, 4, device=dev), torch.zeros(1, 13 * 279, 3, device=dev))
    c = neck_ops.gauss_heads_backward(bev, rgb, prm, 13, g_opacity=g_op)
    torch.cuda.synchronize()
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert float(a[0].abs().max()) > 0
    # only the opacity head (and what feeds it) has a gradient: S, R, Col second layers are exactly zero
    C, L = 80, prm.numel()
    small = a[1][2 * 13 + 16 * C:]
    assert L == 2 * 13 + 16 * C + 83 and small.numel() == 83
    assert float(small[12 + 16:12 + 16 + 35].abs().max()) == 0 and float(small[-15:].abs().max()) == 0
    assert float(small[12 + 16 + 35:12 + 16 + 40].abs().max()) > 0


def test_gauss_heads_backward_refuses_what_it_cannot_do():
    from ocrfdet_amd import _lib, neck_ops
    dev = torch.device('cuda:0')
    vfe, heads = _modules(13, dev)
    prm = neck_ops.pack_gauss_head_params(vfe.eval(), *heads)
    bev, rgb = torch.zeros(1, 80, 4, 4, device=dev), torch.zeros(1, 13, 16, 3, device=dev)
    with pytest.raises(ValueError):
        neck_ops.gauss_heads_backward(bev, rgb, prm[:-1], 13)
    with pytest.raises(ValueError):
        neck_ops.gauss_heads_backward(bev, rgb, prm, 13, g_opacity=torch.zeros(5, device=dev))
    with pytest.raises(_lib.OcrfHipError):
        neck_ops.gauss_heads_backward(bev.cpu(), rgb, prm, 13)
    assert _lib.lib().ocrf_gauss_heads_backward_workspace_bytes(1, 80, 5, 16) == 0          # no register tile for 5 heights
