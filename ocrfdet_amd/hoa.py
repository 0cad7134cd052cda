"""Height-aware Opacity-based Attention (HOA) — modules with the reference's class names,
constructor arguments, forward signatures and ``state_dict`` keys, so checkpoints and the config's
per-layer options (configs/ocrfdet/ocrfdet.py:259-337) carry over:

``HeightAttention(input_channel, output_channel, ratio=16).forward(x) -> (B,C,1,1)``
    view_transformer_ocrf.py:421-461
``OpacityVoxelToBEVConverter(input_channel=13).forward(x, position) -> (B,1,Y,X)``   (HOA-2)
    view_transformer_ocrf.py:463-518
``ObatinOpacityMask(kernel_size=7).forward(x, opacity_bev) -> (B,1,Y,X)``            (HOA-3)
    view_transformer_ocrf.py:230-242; ``.gate(x, opacity_bev)`` also returns ``x * mask`` (:1199)
``DeformableAttention2D(dim, dim_head, heads, dropout, downsample_factor, offset_scale,
offset_groups, offset_kernel_size).forward(x_q, x_kv)``                               (HOA-1)
    mmdet3d/ops/cross_attention_2d.py:93-220 and ``hoa1`` = the glue at
    view_transformer_ocrf.py:1159-1161

The channel reductions and gates run as HIP kernels (``csrc/hoa.hip`` through the C ABI):
channel mean/max + 7x7 conv + sigmoid + gate for HOA-3, global max + quarter MLPs + sigmoid (+ the
``ca(x) * x`` multiply) for HeightAttention.  The small dense convolutions of the UNet and of the
deformable attention stay PyTorch-ROCm ops (plumbing; SURVEY.md 8a rows a24-a25 note they are
~40 tiny launches, not custom-kernel targets of the reference either).  GPU tensors only.
"""
import ctypes

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .neck_ops import tall_linear

__all__ = ['HeightAttention', 'OpacityVoxelToBEVConverter', 'ObatinOpacityMask', 'DeformableAttention2D',
           'CPB', 'hoa1', 'spatial_gate']


def _f32c(t):
    return t.contiguous().float()


class _LaunchCache(dict):
    """Per-module cache of launch plans (ctypes pointers into the module's own buffers).  It is not
    part of the module's state: ``copy.deepcopy(model)`` / pickling (EMA hooks, checkpointing of whole
    modules) get an empty cache and rebuild it on the next forward."""

    def __deepcopy__(self, memo):
        return _LaunchCache()

    def __reduce__(self):
        return (_LaunchCache, ())


class HeightAttention(nn.Module):
    def __init__(self, input_channel, output_channel, ratio=16):
        super().__init__()
        qi, qo = input_channel // 4, output_channel // 4
        self.q_in, self.q_out, self.hid = qi, qo, qi // ratio

        def branch():
            return nn.Sequential(nn.Conv2d(qi, qi // ratio, 1, bias=False), nn.ReLU(inplace=True),
                                 nn.Conv2d(qi // ratio, qo, 1, bias=False))
        self.max_pool1, self.conv1 = nn.AdaptiveMaxPool2d(1), branch()
        self.max_pool2, self.conv2 = nn.AdaptiveMaxPool2d(1), branch()
        self.max_pool3, self.conv3 = nn.AdaptiveMaxPool2d(1), branch()
        self.max_pool4, self.conv4 = nn.AdaptiveMaxPool2d(1), branch()
        self.tanh = nn.Sigmoid()           # the reference names its sigmoid `tanh` (:445)

    def _packed(self):
        """w1 [4][hid][q], w2 [4][q][hid] for the C ABI; cached until a weight changes."""
        convs = (self.conv1, self.conv2, self.conv3, self.conv4)
        ps = [c[i].weight for c in convs for i in (0, 2)]
        key = tuple((p._version, p.data_ptr()) for p in ps)
        if getattr(self, '_pack_key', None) != key:
            w1 = torch.stack([c[0].weight.detach().reshape(self.hid, self.q_in) for c in convs]).contiguous().float()
            w2 = torch.stack([c[2].weight.detach().reshape(self.q_out, self.hid) for c in convs]).contiguous().float()
            self._pack_key, self._pack = key, (w1, w2)
        return self._pack

    def _run(self, x, want_gated):
        _lib.require_cuda(x)
        B, C, Y, X = x.shape
        if C != 4 * self.q_in or self.q_in != self.q_out:
            raise _lib.OcrfHipError(f'HeightAttention built for {4 * self.q_in}->{4 * self.q_out} channels, got {C}')
        x = _f32c(x)
        w1, w2 = self._packed()
        gate = torch.empty(B, C, device=x.device)
        gated = torch.empty_like(x) if want_gated else None
        L = _lib.lib()
        with _lib.on_device(x.device):
            ws = _lib.workspace.get(x.device, L.ocrf_hoa_height_attention_workspace_bytes(B, C), 'hoa')
            _lib.check(L.ocrf_hoa_height_attention(
                _lib.ptr(x), B, C, self.hid, Y, X, _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(gate), _lib.ptr(gated),
                _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_ptr(x.device)), 'ocrf_hoa_height_attention')
        return gate.view(B, C, 1, 1), gated

    def _needs_grad(self, x):
        """The HIP kernels are forward-only: whenever autograd is recording something that reaches this
        module, the reference's op sequence runs as differentiable torch ops instead."""
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _forward_torch(self, x):
        """The differentiable formulation (view_transformer_ocrf.py:499-514): per quarter of the channels a spatial maximum
        and conv1x1 -> ReLU -> conv1x1 on the pooled (B,q,1,1) vector.  The eight 1 x 1 convolutions on 1 x 1 maps are two
        batched matrix products here: as convolutions each took MIOpen's naive kernels — three launches forward + backward,
        ~50 us of host time apiece, ~140 launches per training iteration of the neck."""
        B, q = x.shape[0], self.q_in
        convs = (self.conv1, self.conv2, self.conv3, self.conv4)
        if not all(len(c) == 3 and c[0].bias is None and c[2].bias is None for c in convs):
            outs = [conv(x[:, i * q:(i + 1) * q].flatten(2).max(-1)[0].unsqueeze(-1).unsqueeze(-1)) for i, conv in enumerate(convs)]
            return self.tanh(torch.cat(outs, dim=1))
        # (a maximum WITH its index, like the reference's nn.AdaptiveMaxPool2d(1), :429-456: the backward hands the gradient to
        # ONE maximum of the plane — ``amax`` would spread it over ties, e.g. over a whole plane that a ReLU left at zero.
        # As ``max`` over the flattened plane: torch's adaptive max pooling to 1 x 1 scans a plane with ONE thread — 12 ms
        # of kernels per training iteration at 200 x 200, measured)
        pooled = x.flatten(2).max(-1)[0].reshape(B, 4, q)                                    # (B,4,q)
        w1 = torch.stack([c[0].weight.reshape(self.hid, q) for c in convs])                  # (4,hid,q)
        w2 = torch.stack([c[2].weight.reshape(self.q_out, self.hid) for c in convs])         # (4,q_out,hid)
        hidden = torch.relu(torch.einsum('bgq,ghq->bgh', pooled, w1))
        return self.tanh(torch.einsum('bgh,goh->bgo', hidden, w2).reshape(B, 4 * self.q_out, 1, 1))

    def forward(self, x):
        _lib.require_cuda(x)               # GPU only, also for the differentiable formulation
        if self._needs_grad(x):
            return self._forward_torch(x)
        return self._run(x, False)[0]

    def gate_apply(self, x):
        """``self(x) * x`` in one pass (view_transformer_ocrf.py:499-514)."""
        _lib.require_cuda(x)
        if self._needs_grad(x):
            return self._forward_torch(x) * x
        return self._run(x, True)[1]


def _dw3x3(x, w9, bias):
    B, C, Y, X = x.shape
    y = torch.empty_like(x)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().ocrf_hoa_dw3x3(_lib.ptr(x), _lib.ptr(w9), _lib.ptr(bias), B, C, Y, X, _lib.ptr(y),
                                             _lib.stream_ptr(x.device)), 'ocrf_hoa_dw3x3')
    return y


class _DepthwiseConv3x3(torch.autograd.Function):
    """Depthwise 3x3 convolution (padding 1) with HIP forward and backward (``ocrf_hoa_dw3x3``,
    ``ocrf_hoa_dw3x3_wgrad``): the training path of HOA-2's UNet blocks.  MIOpen's depthwise backward-weight
    takes 2-10 ms per call on these 4..16-channel BEV maps; the weight gradient here is a band-wise partial
    reduction summed in a fixed order (deterministic)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias):
        _lib.require_cuda(x, weight)
        x = _f32c(x)
        C = x.shape[1]
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _dw3x3(x, _f32c(weight.detach()).reshape(C, 9), _f32c(bias.detach()) if bias is not None else None)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        B, C, Y, X = x.shape
        gy = _f32c(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _dw3x3(gy, _f32c(weight.detach().flip(2, 3)).reshape(C, 9), None)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            L = _lib.lib()
            nb = L.ocrf_hoa_dw3x3_wgrad_bands(Y)
            partial = torch.empty(B, C, nb, 10, device=x.device)
            with _lib.on_device(x.device):
                _lib.check(L.ocrf_hoa_dw3x3_wgrad(_lib.ptr(x), _lib.ptr(gy), B, C, Y, X, _lib.ptr(partial),
                                                  _lib.stream_ptr(x.device)), 'ocrf_hoa_dw3x3_wgrad')
            sums = partial.sum((0, 2))                                   # (C, 10)
            gw = sums[:, :9].reshape(C, 1, 3, 3).to(weight.dtype)
            gb = sums[:, 9] if ctx.has_bias else None
        return gx, gw, gb


def _is_dw3x3(conv):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.padding == (1, 1)
            and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.padding_mode == 'zeros'
            and conv.groups == conv.in_channels == conv.out_channels)


class OpacityVoxelToBEVConverter(nn.Module):
    def __init__(self, input_channel=13):
        super().__init__()
        self.encoder1 = self.conv_block(input_channel, 4)
        self.ca1 = HeightAttention(4, 4, 1)
        self.encoder2 = self.conv_block(4, 8)
        self.ca2 = HeightAttention(8, 8, 1)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.bottleneck = self.conv_block(8, 16)
        self.ca_bottleneck = HeightAttention(16, 16, 1)
        self.upconv2 = self.upconv(16, 8)
        self.decoder2 = self.conv_block(16, 8)
        self.ca_dec2 = HeightAttention(8, 8, 1)
        self.upconv1 = self.upconv(8, 4)
        self.decoder1 = self.conv_block(8, 4)
        self.ca_dec1 = HeightAttention(4, 4, 1)
        self.output_conv = nn.Conv2d(4, 1, kernel_size=1)

    @staticmethod
    def conv_block(in_channels, out_channels):
        return nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, padding=1, groups=in_channels),
                             nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels),
                             nn.ReLU(inplace=True))

    @staticmethod
    def upconv(in_channels, out_channels):
        return nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)

    def forward(self, x, position):
        if not self.training and x.is_cuda and not (torch.is_grad_enabled() and (
                x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self._forward_fused(x, position)
        # training mode (BatchNorm batch statistics): block by block; the depthwise layers (forward and
        # backward) and the gates run in HIP, the 1x1 layers / BatchNorm / pooling are torch ops
        blk = self._block
        enc1 = self.ca1.gate_apply(blk(self.encoder1, x) + position)
        enc2 = self.ca2.gate_apply(blk(self.encoder2, self.pool(enc1)))
        mid = self.ca_bottleneck.gate_apply(blk(self.bottleneck, self.pool(enc2)))
        dec2 = self.ca_dec2.gate_apply(blk(self.decoder2, torch.cat((self.upconv2(mid), enc2), dim=1)))
        dec1 = self.ca_dec1.gate_apply(blk(self.decoder1, torch.cat((self.upconv1(dec2), enc1), dim=1)))
        return self.output_conv(dec1)

    @staticmethod
    def _block(block, x):
        dw = block[0]
        if x.is_cuda and _is_dw3x3(dw):
            x = _DepthwiseConv3x3.apply(x, dw.weight, dw.bias)
            for layer in list(block)[1:]:
                x = layer(x)
            return x
        return block(x)

    def _folded(self, block):
        """(dw_w (Cin,9), dw_b, pw_w (Cout,Cin) with BatchNorm folded in, pw_b) of a conv_block;
        cached until one of the tensors involved changes."""
        dw, pw, bn = block[0], block[1], block[2]
        ts = (dw.weight, dw.bias, pw.weight, pw.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t._version, t.data_ptr()) for t in ts)
        cache = self.__dict__.setdefault('_fold_cache', {})
        hit = cache.get(id(block))
        if hit is None or hit[0] != key:
            with torch.no_grad():
                s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                pw_w = pw.weight.reshape(pw.out_channels, pw.in_channels) * s[:, None]
                pw_b = (pw.bias - bn.running_mean) * s + bn.bias
                vals = [t.detach().contiguous().float()
                        for t in (dw.weight.reshape(dw.in_channels, 9), dw.bias, pw_w, pw_b)]
            cache[id(block)] = hit = (key, vals)
        return hit[1]

    def _plan(self, B, H, W, dev):
        """Everything of the fused forward that does not depend on the input values, built once per
        (shape, device, weight version): folded / packed weights, the intermediate buffers and the
        ctypes argument lists of the 11 launches.  The per-call host work is then one version check,
        one allocation (the result) and the launches themselves (the block-by-block Python of the
        first version cost ~0.3 ms per call, more than the kernels)."""
        cache = self.__dict__.get('_plan_cache')
        if cache is None:
            cache = self.__dict__['_plan_cache'] = _LaunchCache()
        tensors = cache.get('tensors')
        if tensors is None:
            tensors = cache['tensors'] = list(self.parameters()) + list(self.buffers())
        key = (B, H, W, dev, tuple(t._version for t in tensors), tuple(t.data_ptr() for t in tensors))
        plan = cache.get('plan')
        if plan is not None and plan['key'] == key:
            return plan
        L = _lib.lib()
        keep, calls = [], []                      # keep: tensors the pointers below refer to

        def dev_f32(t):
            t = t.detach().contiguous().float()
            keep.append(t)
            return t

        def block(src0, gate0, mode, up, src1, gate1, conv, ca, cout, h, w, addend_slot=False):
            dw_w, dw_b, pw_w, pw_b = (dev_f32(t) for t in self._folded(conv))
            up_w = dev_f32(up.weight) if up is not None else None
            up_b = dev_f32(up.bias) if up is not None else None
            cup = up.out_channels if up is not None else 0
            out = torch.empty(B, cout, h, w, device=dev)
            n_tiles = L.ocrf_hoa_unet_tiles(h, w)
            pmax = torch.empty(B * cout, n_tiles, device=dev)
            gate = torch.empty(B, cout, device=dev)
            w1, w2 = (dev_f32(t) for t in ca._packed())
            keep.extend((out, pmax, gate))
            c0 = src0.shape[1] if src0 is not None else self.encoder1[0].in_channels
            h0, w0 = (src0.shape[2], src0.shape[3]) if src0 is not None else (H, W)
            args = [_lib.ptr(src0), _lib.ptr(gate0), c0, h0, w0, mode, _lib.ptr(up_w), _lib.ptr(up_b), cup,
                    _lib.ptr(src1), _lib.ptr(gate1), src1.shape[1] if src1 is not None else 0, _lib.ptr(dw_w),
                    _lib.ptr(dw_b), _lib.ptr(pw_w), _lib.ptr(pw_b), cout, _lib.ptr(None), _lib.ptr(out), _lib.ptr(pmax),
                    B, h, w]
            calls.append(('block', args, addend_slot))
            calls.append(('gate', [B, cout, ca.hid, n_tiles, _lib.ptr(pmax), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(gate)]))
            return out, gate
        with torch.no_grad():
            e1, g1 = block(None, None, 0, None, None, None, self.encoder1, self.ca1, 4, H, W, addend_slot=True)
            e2, g2 = block(e1, g1, 1, None, None, None, self.encoder2, self.ca2, 8, H // 2, W // 2)
            bt, gb = block(e2, g2, 1, None, None, None, self.bottleneck, self.ca_bottleneck, 16, H // 4, W // 4)
            d2, gd2 = block(bt, gb, 2, self.upconv2, e2, g2, self.decoder2, self.ca_dec2, 8, H // 2, W // 2)
            d1, gd1 = block(d2, gd2, 2, self.upconv1, e1, g1, self.decoder1, self.ca_dec1, 4, H, W)
            ow, ob = dev_f32(self.output_conv.weight.reshape(-1)), dev_f32(self.output_conv.bias)
        plan = dict(key=key, keep=keep, calls=calls,
                    out_args=[_lib.ptr(d1), _lib.ptr(gd1), B, 4, H, W, _lib.ptr(ow), _lib.ptr(ob)])
        cache['plan'] = plan
        return plan

    def _packed_v2b(self):
        """All weights of the converter as ONE float vector in the order ``ocrf_hoa_v2b_forward`` reads them
        (csrc/hoa_v2b.hip ``v2b_offsets``): per block (encoder1, encoder2, bottleneck, decoder2, decoder1) the folded
        depthwise / pointwise weights and that block's HeightAttention w1, w2; the two up-convolutions; the output
        conv.  Cached until a parameter or buffer changes."""
        cache = self.__dict__.get('_plan_cache')
        if cache is None:
            cache = self.__dict__['_plan_cache'] = _LaunchCache()
        tensors = cache.get('tensors')
        if tensors is None:
            tensors = cache['tensors'] = list(self.parameters()) + list(self.buffers())
        key = (tuple(t._version for t in tensors), tuple(t.data_ptr() for t in tensors))
        hit = cache.get('packed')
        if hit is not None and hit[0] == key:
            return hit[1]
        parts = []
        with torch.no_grad():
            for blk, ca in ((self.encoder1, self.ca1), (self.encoder2, self.ca2), (self.bottleneck, self.ca_bottleneck),
                            (self.decoder2, self.ca_dec2), (self.decoder1, self.ca_dec1)):
                parts.extend(t.reshape(-1) for t in self._folded(blk))
                parts.extend(t.reshape(-1) for t in ca._packed())
            for up in (self.upconv2, self.upconv1):
                parts.extend((up.weight.detach().float().reshape(-1), up.bias.detach().float().reshape(-1)))
            parts.extend((self.output_conv.weight.detach().float().reshape(-1), self.output_conv.bias.detach().float().reshape(-1)))
            packed = torch.cat([t.float() for t in parts])
            n = _lib.lib().ocrf_hoa_v2b_weights_len()           # the vector zero-padded to whole 16-byte words
            assert 0 <= n - packed.numel() < 4, (n, packed.numel())
            packed = torch.cat((packed, packed.new_zeros(n - packed.numel()))).contiguous()
        cache['packed'] = (key, packed)
        return packed

    def _is_reference_architecture(self):
        e1 = self.encoder1[0]
        return (e1.in_channels == 13 and self.ca1.hid == self.ca1.q_in == 1 and self.ca_bottleneck.hid == 4)

    def _forward_fused(self, x, position):
        """Eval-mode forward as ONE C call of six launches (csrc/hoa_v2b.hip, ``ocrf_hoa_v2b_forward``): five fused
        block kernels, each computing the HeightAttention gates of its producers in its own prologue, and the gated
        output conv; no intermediate pooled / upsampled / concatenated / gated tensor is ever written.  The
        intermediates live in the library's 'hoa_v2b' scratch buffer: one forward at a time per stream."""
        _lib.require_cuda(x, position)
        x, position = _f32c(x), _f32c(position)
        B, _, H, W = x.shape
        dev = x.device
        L = _lib.lib()
        if self._is_reference_architecture() and H % 4 == 0 and W % 4 == 0:
            w = self._packed_v2b()
            out = torch.empty(B, 1, H, W, device=dev)
            with _lib.on_device(dev):
                ws = _lib.workspace.get(dev, L.ocrf_hoa_v2b_workspace_bytes(B, H, W), 'hoa_v2b')
                _lib.check(L.ocrf_hoa_v2b_forward(_lib.ptr(x), _lib.ptr(position), _lib.ptr(w), B, H, W, _lib.ptr(ws),
                                                  ctypes.c_size_t(ws.numel()), _lib.ptr(out), _lib.stream_ptr(dev)),
                           'ocrf_hoa_v2b_forward')
            return out
        return self._forward_blocks(x, position)

    def _forward_blocks(self, x, position):
        """Any other channel configuration: block by block (5 block kernels + 5 gate kernels + the output conv)."""
        B, _, H, W = x.shape
        dev = x.device
        L = _lib.lib()
        plan = self._plan(B, H, W, dev)
        out = torch.empty(B, 1, H, W, device=dev)
        with _lib.on_device(dev):
            st = _lib.stream_ptr(dev)
            for call in plan['calls']:
                if call[0] == 'block':
                    args = call[1]
                    if call[2]:                   # the first block reads the caller's tensors
                        args = list(args)
                        args[0], args[17] = _lib.ptr(x), _lib.ptr(position)
                    _lib.check(L.ocrf_hoa_unet_block(*args, st), 'ocrf_hoa_unet_block')
                else:
                    _lib.check(L.ocrf_hoa_height_gate_from_tiles(*call[1], st), 'ocrf_hoa_height_gate_from_tiles')
            _lib.check(L.ocrf_hoa_gated_conv1x1(*plan['out_args'], _lib.ptr(out), st), 'ocrf_hoa_gated_conv1x1')
        return out


def channel_stats(x):
    """[mean_c(x), max_c(x)] (B,2,Y,X): the first of the two HIP kernels of ``spatial_gate``.  It reads nothing but x, so a
    caller may issue it early, on another stream, and hand the result to ``spatial_gate(..., stats=...)``."""
    _lib.require_cuda(x)
    B, C, Y, X = x.shape
    x = _f32c(x)
    stats = torch.empty(B, 2, Y, X, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().ocrf_hoa_channel_stats(_lib.ptr(x), B, C, Y, X, _lib.ptr(stats), _lib.stream_ptr(x.device)),
                   'ocrf_hoa_channel_stats')
    return stats


def spatial_gate(weight, x, addend, want_gated, stats=None, in_place=False):
    """mask = sigmoid(conv_kxk([mean_c(x), max_c(x)]; weight (1,2,k,k)) + addend (B,1,Y,X)) and,
    optionally, x * mask — the shared form of ObatinOpacityMask (view_transformer_ocrf.py:230-242,
    :1197-1199) and BEVGeomAttention (:215-228, :1190), as two HIP kernels (csrc/hoa.hip).  ``stats``: the result of
    ``channel_stats(x)`` if the caller already has it (ordered before this call on the current stream) — or the statistics
    of a LARGER channel set x is a slice of (camera-frame sharding: a rank gates its plane block with the group's
    combined statistics).  ``in_place``: x * mask overwrites x (every element is read by the thread that writes it, before
    it writes: the kernel's x and gated may be the same tensor)."""
    _lib.require_cuda(x, addend)
    B, C, Y, X = x.shape
    if in_place and (x.dtype != torch.float32 or not x.is_contiguous()):
        raise _lib.OcrfHipError('in_place gating needs a contiguous fp32 tensor')
    x, ob = _f32c(x), _f32c(addend)
    w = _f32c(weight)
    k = w.shape[-1]
    if stats is not None and (tuple(stats.shape) != (B, 2, Y, X) or stats.dtype != torch.float32 or not stats.is_contiguous()):
        raise _lib.OcrfHipError('stats must be the contiguous fp32 (B,2,Y,X) result of channel_stats(x)')
    mask = torch.empty(B, 1, Y, X, device=x.device)
    gated = (x if in_place else torch.empty_like(x)) if want_gated else None
    L = _lib.lib()
    with _lib.on_device(x.device):
        st = _lib.stream_ptr(x.device)
        if stats is None:
            stats = torch.empty(B, 2, Y, X, device=x.device)
            _lib.check(L.ocrf_hoa_channel_stats(_lib.ptr(x), B, C, Y, X, _lib.ptr(stats), st), 'ocrf_hoa_channel_stats')
        _lib.check(L.ocrf_hoa_opacity_mask_gate(_lib.ptr(x), _lib.ptr(stats), _lib.ptr(ob), _lib.ptr(w), k, B, C,
                                                Y, X, _lib.ptr(mask), _lib.ptr(gated), st),
                   'ocrf_hoa_opacity_mask_gate')
    return mask, gated


class ObatinOpacityMask(nn.Module):          # sic: the reference's spelling
    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.sigmoid = nn.Sigmoid()

    def _run(self, x, opacity_bev, want_gated, stats=None, in_place=False):
        _lib.require_cuda(x, opacity_bev)
        if torch.is_grad_enabled() and (x.requires_grad or opacity_bev.requires_grad or self.conv.weight.requires_grad):
            # forward-only HIP kernels: under autograd the reference's ops (:236-242), differentiable
            stats = torch.cat((x.mean(1, keepdim=True), torch.max(x, dim=1, keepdim=True)[0]), 1)     # (:225, :238)
            mask = self.sigmoid(self.conv(stats) + opacity_bev)
            return mask, (x * mask if want_gated else None)
        return spatial_gate(self.conv.weight, x, opacity_bev, want_gated, stats=stats, in_place=in_place)

    def forward(self, x, opacity_bev):
        return self._run(x, opacity_bev, False)[0]

    def gate(self, x, opacity_bev, stats=None, in_place=False):
        """-> (mask, x * mask): view_transformer_ocrf.py:1197-1199 in two HBM passes.  ``stats``: ``channel_stats`` of x
        — or of the whole channel set x is a block of — if the caller has it already; ``in_place``: x is overwritten."""
        return self._run(x, opacity_bev, True, stats=stats, in_place=in_place)


# ------------------------------------------------------------------------------------------------
# HOA-1: deformable cross attention on the (1,13,21,21) opacity / alpha maps
# ------------------------------------------------------------------------------------------------
def _grid_like(t, dim=0):
    h, w = t.shape[-2:]
    xs = torch.arange(w, device=t.device)
    ys = torch.arange(h, device=t.device)
    gx, gy = torch.meshgrid(xs, ys, indexing='xy')
    return torch.stack((gx, gy), dim=dim).type_as(t)


def _normalize_grid(grid, dim=1, out_dim=-1):
    # as written in the reference (cross_attention_2d.py:30-38): channel 0 over (h-1), 1 over (w-1)
    h, w = grid.shape[-2:]
    g0, g1 = grid.unbind(dim=dim)
    return torch.stack((2.0 * g0 / max(h - 1, 1) - 1.0, 2.0 * g1 / max(w - 1, 1) - 1.0), dim=out_dim)


class _Scale(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.scale = scale

    def forward(self, x):
        return x * self.scale


class CPB(nn.Module):
    """Continuous positional bias MLP (cross_attention_2d.py:49-89)."""

    def __init__(self, dim, *, heads, offset_groups, depth):
        super().__init__()
        self.heads, self.offset_groups = heads, offset_groups
        self.mlp = nn.ModuleList([nn.Sequential(nn.Linear(2, dim), nn.ReLU())])
        for _ in range(depth - 1):
            self.mlp.append(nn.Sequential(nn.Linear(dim, dim), nn.ReLU()))
        self.mlp.append(nn.Linear(dim, heads // offset_groups))

    def forward(self, grid_q, grid_kv):
        gq = grid_q.reshape(1, -1, grid_q.shape[-1])
        gk = grid_kv.reshape(grid_kv.shape[0], -1, grid_kv.shape[-1])
        pos = gq[:, :, None, :] - gk[:, None, :, :]
        bias = torch.sign(pos) * torch.log(pos.abs() + 1)
        # (B x queries x keys rows — 139 392 at cfg2 — through Linear layers 2 / 3 columns wide: under autograd their weight
        # gradients are tall reductions that hipBLASLt runs at 0.4-0.5 ms each; neck_ops.tall_linear splits K)
        for layer in self.mlp:
            bias = torch.relu(tall_linear(layer[0], bias)) if isinstance(layer, nn.Sequential) else tall_linear(layer, bias)
        bg, i, j, o = bias.shape
        g = self.offset_groups
        return bias.view(bg // g, g, i, j, o).permute(0, 1, 4, 2, 3).reshape(bg // g, g * o, i, j)


class DeformableAttention2D(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8, dropout=0., downsample_factor=4, offset_scale=None,
                 offset_groups=None, offset_kernel_size=6, group_queries=True, group_key_values=True):
        super().__init__()
        offset_scale = downsample_factor if offset_scale is None else offset_scale
        assert offset_kernel_size >= downsample_factor
        assert (offset_kernel_size - downsample_factor) % 2 == 0
        offset_groups = heads if offset_groups is None else offset_groups
        assert heads % offset_groups == 0
        inner = dim_head * heads
        self.scale = dim_head ** -0.5
        self.heads, self.offset_groups, self.downsample_factor = heads, offset_groups, downsample_factor
        od = inner // offset_groups
        self.to_offsets = nn.Sequential(
            nn.Conv2d(od, od, offset_kernel_size, groups=od, stride=downsample_factor,
                      padding=(offset_kernel_size - downsample_factor) // 2),
            nn.GELU(), nn.Conv2d(od, 2, 1, bias=False), nn.Tanh(), _Scale(offset_scale))
        self.rel_pos_bias = CPB(dim // 4, offset_groups=offset_groups, heads=heads, depth=2)
        self.dropout = nn.Dropout(dropout)
        self.to_q = nn.Conv2d(dim, inner, 1, groups=offset_groups if group_queries else 1, bias=False)
        self.to_k = nn.Conv2d(dim, inner, 1, groups=offset_groups if group_key_values else 1, bias=False)
        self.to_v = nn.Conv2d(dim, inner, 1, groups=offset_groups if group_key_values else 1, bias=False)
        self.to_out = nn.Conv2d(inner, dim, 1)

    def forward(self, x_q, x_kv, return_vgrid=False):
        b, _, h, w = x_q.shape
        g, heads = self.offset_groups, self.heads
        q = self.to_q(x_q)

        def grp(t):                               # 'b (g d) ... -> (b g) d ...'
            return t.reshape(t.shape[0] * g, t.shape[1] // g, *t.shape[2:])
        offsets = self.to_offsets(grp(q))
        vgrid = _grid_like(offsets) + offsets
        vgrid_scaled = _normalize_grid(vgrid)
        kv = F.grid_sample(grp(x_kv), vgrid_scaled, mode='bilinear', padding_mode='zeros', align_corners=False)
        kv = kv.reshape(b, -1, *kv.shape[2:])
        k, v = self.to_k(kv), self.to_v(kv)
        q = q * self.scale

        def split(t):                             # 'b (h d) ... -> b h (...) d'
            return t.reshape(t.shape[0], heads, t.shape[1] // heads, -1).transpose(2, 3)
        q, k, v = split(q), split(k), split(v)
        sim = q @ k.transpose(-1, -2)
        grid_scaled = _normalize_grid(_grid_like(x_kv), dim=0)
        sim = sim + self.rel_pos_bias(grid_scaled, vgrid_scaled)
        sim = sim - sim.amax(dim=-1, keepdim=True).detach()
        attn = self.dropout(sim.softmax(dim=-1))
        out = (attn @ v).transpose(2, 3).reshape(b, -1, h, w)
        out = self.to_out(out)
        return (out, vgrid) if return_vgrid else out


def _hoa1_packed(m):
    """The module's weights in the layout of ``ocrf_hoa1_forward`` (csrc/hoa.hip); cached."""
    ps = [m.to_q.weight, m.to_offsets[0].weight, m.to_offsets[0].bias, m.to_offsets[2].weight, m.to_k.weight,
          m.to_v.weight, m.to_out.weight, m.to_out.bias, m.rel_pos_bias.mlp[0][0].weight, m.rel_pos_bias.mlp[0][0].bias,
          m.rel_pos_bias.mlp[1][0].weight, m.rel_pos_bias.mlp[1][0].bias, m.rel_pos_bias.mlp[2].weight,
          m.rel_pos_bias.mlp[2].bias]
    key = tuple((p._version, p.data_ptr()) for p in ps)
    if getattr(m, '_hoa1_key', None) != key:
        with torch.no_grad():
            m._hoa1_pack = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
        m._hoa1_key = key
    return m._hoa1_pack


def _hoa1_fusable(m, x):
    return (x.is_cuda and not m.training and not (torch.is_grad_enabled() and (x.requires_grad or any(
        p.requires_grad for p in m.parameters()))) and m.heads == 1 and m.offset_groups == 1 and m.downsample_factor == 4
            and m.to_q.in_channels == 13 and m.to_q.out_channels == 8 and m.to_offsets[0].kernel_size == (6, 6)
            and len(m.rel_pos_bias.mlp) == 3 and m.rel_pos_bias.mlp[0][0].out_features == 3)


def hoa1(defor_cross_attention, opacity, alpha_lidar, heights, Y, X):
    """view_transformer_ocrf.py:1159-1161: opacity (B*heights*Y*X, 1) from A_MLP (B = 1 in the
    reference's per-sample loop), alpha_lidar (B,heights,Y,X) -> opacity_alpha (B,heights,Y,X).
    Eval mode on the GPU runs the two fused HIP kernels; otherwise the reference's op sequence."""
    m = defor_cross_attention
    o = opacity.view(-1, heights, Y, X)
    if _hoa1_fusable(m, o) and heights == 13:
        _lib.require_cuda(o, alpha_lidar)
        o32, a32 = _f32c(o), _f32c(alpha_lidar)
        B = o32.shape[0]
        L = _lib.lib()
        w = _hoa1_packed(m)
        assert w.numel() == L.ocrf_hoa1_weights_len()
        att = torch.empty(B * ((heights + 8) * (Y // 6) * (X // 6) + 18 * 128), device=o.device)   # att | q | kv
        out = torch.empty_like(o32)
        with _lib.on_device(o.device):
            _lib.check(L.ocrf_hoa1_forward(_lib.ptr(o32), _lib.ptr(a32), _lib.ptr(w), B, Y, X,
                                           ctypes.c_float(float(m.to_offsets[4].scale)), _lib.ptr(att), _lib.ptr(out),
                                           _lib.stream_ptr(o.device)), 'ocrf_hoa1_forward')
        return out
    size = (int(Y / 6), int(X / 6))
    o_up = F.interpolate(o, size=size, mode='bilinear', align_corners=True)
    a_up = F.interpolate(alpha_lidar, size=size, mode='bilinear', align_corners=True)
    att = m(o_up, a_up)
    return F.interpolate(att, size=(Y, X), mode='bilinear', align_corners=True) + o
