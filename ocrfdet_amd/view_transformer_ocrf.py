"""``OcRFViewTransformerFull`` — the reference's neck (mmdet3d/models/necks/view_transformer_ocrf.py:
577-648 ctor, :1040-1201 ``view_transform_core``, :1319-1334 ``forward``) with the same constructor
keywords, sub-module attribute names and ``state_dict`` keys (the config addresses ``S_MLP``, ``R_MLP``,
``A_MLP``, ``C_MLP``, ``C_MLP_nerf``, ``D_MLP_nerf``, ``sigma``, ``img_feat_resize1/2`` by name,
configs/ocrfdet/ocrfdet.py:259-337), assembled MI355X-first:

  eval mode (inference; everything below runs as HIP kernels behind include/ocrf_hip.h unless noted)
    pre-filter + channels-last feature ............ ocrf_prefilter
    rank vectors of both poolings ................. ocrf_lss_prepare / ocrf_ht_prepare (cached when
                                                    ``accelerate=True``)
    LSS pool, HT pool ............................. ocrf_bev_pool_v2_nchw
    voxel colours, alpha volume ................... ocrf_pillar_sample_mean
    sparse ground-truth image ..................... ocrf_retain_valid_pixels (selected camera only)
    voxel lift + 4 Gaussian heads ................. ocrf_gauss_heads (no (B,13,Y,X,80) tensor)
    NeRF branch ................................... 3 small MIOpen convolutions batched over all
                                                    cameras, then ocrf_nerf_alpha / ocrf_nerf_render on
                                                    composed maps (no (80,H,W) feature images)
    Gaussian render ............................... ocrf_rasterize_forward
    HOA-1 / HOA-2 / HOA-3, geometry attention ..... ocrf_hoa1_forward, ocrf_hoa_unet_block, ...,
                                                    ocrf_hoa_opacity_mask_gate
    DualFeatFusion / ProbNet convolutions ......... PyTorch-ROCm (MIOpen), SURVEY 8a row a27
  training mode: the reference's op sequence as differentiable torch ops around the differentiable
    HIP ops (bev_pool_v2 and the rasteriser have HIP backwards).

``depth_net`` (a CNN outside the path, view_transformer.py:463-630) is injected: pass
``depth_net=module``; when the reference's mmdet3d is importable its ``DepthNet`` is built from
``depthnet_cfg`` exactly as the reference does (:588-589).
"""
import contextlib
import math
import random

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, bevpool, gaussian_renderer, hoa, index_prep, neck_ops
from .diff_gaussian_rasterization import pack_cameras, rasterize_packed_autograd, rasterize_sets

__all__ = ['OcRFViewTransformerFull', 'GraphedNeck', 'MS_CAM', 'ChannelAttention', 'SpatialAttention', 'ResCBAMBlock', 'ProbNet',
           'DualFeatFusion', 'BEVGeomAttention', 'ScaleFactorMLP', 'RotationFactorMLP', 'OpacityFactorMLP',
           'ColorFactorMLPGaussian', 'ColorFactorMLPNerf', 'DepthFactorMLPNerf', 'ImgFeatResize1', 'ImgFeatResize2',
           'VoxelFeatureExtractor', 'ResizeNetwork', 'LinearWeightedImage', 'LinearWeightedDepth',
           'LearnedPositionalEncoding', 'DiceLoss']


# ------------------------------------------------------------------------------------------------
# small building blocks (structure fixed by the reference's state_dict; :36-228, :272-380, :520-575)
# ------------------------------------------------------------------------------------------------
def _conv_bn(cin, cout, k=1, relu=False):
    layers = [nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=k // 2), nn.BatchNorm2d(cout)]
    return layers + [nn.ReLU(inplace=True)] if relu else layers


class MS_CAM(nn.Module):
    """Multi-scale channel attention (:36-66).  The pooling modules are kept for the Sequential's
    indices (state_dict keys); the reductions themselves are ``mean`` / ``amax`` — PyTorch's
    adaptive-pool kernels run one thread per output element, 3.4 ms per call on a 200x200 map."""

    def __init__(self, input_channel=64, output_channel=64, r=4):
        super().__init__()
        mid = int(input_channel // r)
        self.local_att = nn.Sequential(*_conv_bn(input_channel, mid, relu=True), *_conv_bn(mid, output_channel))
        self.global_att = nn.Sequential(nn.AdaptiveAvgPool2d(1), *_conv_bn(input_channel, mid, relu=True),
                                        *_conv_bn(mid, output_channel))
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        g = x.mean((2, 3), keepdim=True)
        for layer in list(self.global_att)[1:]:
            g = layer(g)
        return self.sigmoid(self.local_att(x) + g)


class ChannelAttention(nn.Module):
    def __init__(self, input_channel, output_channel, ratio=16):
        super().__init__()
        self.avg_pool, self.max_pool = nn.AdaptiveAvgPool2d(1), nn.AdaptiveMaxPool2d(1)
        self.fc = nn.Sequential(nn.Conv2d(input_channel, input_channel // ratio, 1, bias=False), nn.ReLU(),
                                nn.Conv2d(input_channel // ratio, output_channel, 1, bias=False))
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        # (the maximum with its index, as the reference's AdaptiveMaxPool2d(1), :72-83 — see hoa.HeightAttention._forward_torch)
        peak = x.flatten(2).max(-1)[0].unsqueeze(-1).unsqueeze(-1)
        return self.sigmoid(self.fc(x.mean((2, 3), keepdim=True)) + self.fc(peak))


def _tensors_of(owner, name, modules, buffers=True):
    """The parameter (+ buffer) tensors of ``modules`` as a list cached on ``owner``: enumerating them through
    ``Module.parameters()`` costs ~0.1 ms per call, a third of the host time of an eager forward.  The weight-pack
    caches key on these tensors' versions / addresses, which ``load_state_dict``, optimizer steps and ``.to()``
    all change while the tensor OBJECTS stay; ``OcRFViewTransformerFull`` drops the lists on ``_apply`` /
    ``load_state_dict`` / ``train`` anyway.  Assigning a NEW ``nn.Parameter`` to a sub-module needs
    ``invalidate_packs()``."""
    cache = owner.__dict__.setdefault('_tensor_lists', {})
    lst = cache.get(name)
    if lst is None:
        lst = cache[name] = [t for m in modules
                             for t in list(m.parameters()) + (list(m.buffers()) if buffers else [])]
    return lst


def _drop_tensor_lists(root):
    for m in root.modules():
        m.__dict__.pop('_tensor_lists', None)


_ZEROS = {}


def _zeros(shape, like):
    """A persistent all-zero tensor per (shape, device, dtype): read-only operand of fused kernels.  Not
    ``torch.zeros`` per call: that is a memset, and memset NODES make a captured hipGraph of the step
    fault on replay after any intervening copy (csrc/launch.h, zero_async)."""
    key = (tuple(shape), like.device, like.dtype)
    t = _ZEROS.get(key)
    if t is None:
        t = _ZEROS[key] = torch.zeros(shape, device=like.device, dtype=like.dtype)
    return t


def _spatial_logits(x, conv):
    return conv(torch.cat((x.mean(1, keepdim=True), torch.max(x, dim=1, keepdim=True)[0]), 1))    # (:94-96: max with indices)


class SpatialAttention(nn.Module):
    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv1 = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return self.sigmoid(_spatial_logits(x, self.conv1))


class ResCBAMBlock(nn.Module):
    """:100-137.  Note the reference feeds ``conv2`` with ``inplanes`` channels (== planes here)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.ca, self.sa = ChannelAttention(planes, planes), SpatialAttention()
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        out = self.ca(out) * out
        if out.is_cuda and not torch.is_grad_enabled():
            out = hoa.spatial_gate(self.sa.conv1.weight, out, _zeros((out.shape[0], 1) + out.shape[2:], out), True)[1]
        else:
            out = self.sa(out) * out
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(out + res)


class DiceLoss(nn.Module):
    def __init__(self, use_sigmoid=True, loss_weight=1.):
        super().__init__()
        self.use_sigmoid, self.loss_weight = use_sigmoid, loss_weight

    def forward(self, inputs, targets, smooth=1e-5):
        p = (torch.sigmoid(inputs) if self.use_sigmoid else inputs).reshape(-1)
        t = targets.reshape(-1)
        dice = (2. * (p * t).sum() + smooth) / (p.sum() + t.sum() + smooth)
        return self.loss_weight * (1 - dice)


class ProbNet(nn.Module):
    """BEV foreground-probability head (:139-201); the losses keep their buffers so checkpoints load."""

    def __init__(self, in_channels=512, scale_factor=1, with_centerness=False, loss_weight=6.0, bev_size=None):
        super().__init__()
        self.loss_weight, self.loss_weight_opacity = loss_weight, 6.0
        mid = in_channels // 2
        self.base_conv = nn.Sequential(*_conv_bn(in_channels, mid, k=3, relu=True))
        self.prob_conv = nn.Sequential(ResCBAMBlock(mid, mid))
        self.mask_net = nn.Conv2d(mid, 1, kernel_size=1, padding=0, stride=1)
        self.with_centerness = with_centerness
        if with_centerness:
            n = bev_size[0]
            g = torch.stack(torch.meshgrid(torch.arange(n), torch.arange(n), indexing='ij'), -1)
            g = (g - n // 2) / (n // 2)
            self.centerness = ((g[..., 0] ** 2 + g[..., 1] ** 2) / 2).sqrt() + 1       # plain attribute (:167)
        self.dice_loss = DiceLoss(use_sigmoid=True, loss_weight=self.loss_weight)
        self.dice_loss_opacity = DiceLoss(use_sigmoid=True, loss_weight=self.loss_weight_opacity)
        self.ce_loss = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([2.13]))

    def _fusable(self, x):
        blk = self.prob_conv[0]
        return (x.is_cuda and not self.training and not torch.is_grad_enabled() and len(self.prob_conv) == 1
                and blk.downsample is None and blk.stride == 1 and self.mask_net.out_channels == 1)

    def forward(self, input):
        if self._fusable(input):
            # eval on the GPU: MIOpen's three convolutions with the BatchNorms folded + six HIP launches
            # (neck_ops.probnet_forward) instead of ~33 launches
            ts = _tensors_of(self, 'prob', (self,))
            key = tuple((t._version, t.data_ptr()) for t in ts)
            if self.__dict__.get('_pack_key') != key:
                self.__dict__['_pack'] = neck_ops.pack_probnet(self)
                self.__dict__['_pack_key'] = key
            return neck_ops.probnet_forward(input, self.__dict__['_pack'])
        return self.mask_net(self.prob_conv(self.base_conv(input)))


class DualFeatFusion(nn.Module):
    """:203-213.  Eval mode on the GPU: one HIP pass (``ocrf_dual_feat_fusion``) instead of cat + two
    1x1 convolutions + two BatchNorms + ReLU + seven elementwise passes over the BEV."""

    def __init__(self, input_channel, output_channel):
        super().__init__()
        self.ca = MS_CAM(input_channel, output_channel)

    def _fusable(self, x1):
        la = self.ca.local_att
        c, m = la[3].out_channels, la[0].out_channels
        return (x1.is_cuda and not self.training and not torch.is_grad_enabled() and la[0].in_channels == 2 * c
                and (c, m) in ((80, 40), (64, 32)))

    def forward(self, x1, x2, addend=None):
        """``addend`` (eval mode on the GPU only): also return ``addend + out``, written by the same pass."""
        if self._fusable(x1):
            ca = self.ca
            ts = _tensors_of(self, 'fuser', (ca.local_att, ca.global_att))
            key = tuple((t._version, t.data_ptr()) for t in ts)
            if self.__dict__.get('_pack_key') != key:
                self.__dict__['_pack'] = (neck_ops.pack_fusion_params(ca), neck_ops.pack_global_att(ca))
                self.__dict__['_pack_key'] = key
            local, glob = self.__dict__['_pack']
            g = neck_ops.global_att_vector(x1, x2, glob)
            return neck_ops.dual_feat_fusion(x1, x2, local, g, ca.local_att[0].out_channels, addend=addend)
        cf = self.ca(torch.cat((x1, x2), 1))
        out = cf * x1 + (1 - cf) * x2
        return out if addend is None else (out, addend + out)


class BEVGeomAttention(nn.Module):
    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv1 = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x, bev_prob):
        if x.is_cuda and not torch.is_grad_enabled():
            return hoa.spatial_gate(self.conv1.weight, x, bev_prob, False)[0]
        return self.sigmoid(_spatial_logits(x, self.conv1) + bev_prob)

    def gate(self, x, bev_prob):
        """-> x * forward(x, bev_prob) (:1190) in two HBM passes."""
        if x.is_cuda and not torch.is_grad_enabled():
            return hoa.spatial_gate(self.conv1.weight, x, bev_prob, True)[1]
        return self.forward(x, bev_prob) * x


_TallLinear, _linear = neck_ops.TallLinear, neck_ops.tall_linear


class _Head(nn.Module):
    """fc1 -> ReLU -> fc2 -> ``act``; ``extra`` widens fc1's input by the sampled RGB (:272-380)."""
    extra = 0

    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__()
        self.fc1 = nn.Linear(input_dim + self.extra, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)

    def act(self, x):
        return x

    def forward(self, x):
        return self.act(_linear(self.fc2, torch.relu(_linear(self.fc1, x))))


class ScaleFactorMLP(_Head):
    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__(input_dim, hidden_dim, output_dim)
        self.softplus = nn.Softplus()

    def act(self, x):
        return self.softplus(x)


class RotationFactorMLP(_Head):
    def act(self, x):
        return F.normalize(x, dim=-1)


class OpacityFactorMLP(_Head):
    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__(input_dim, hidden_dim, output_dim)
        self.sigmoid = nn.Sigmoid()

    def act(self, x):
        return self.sigmoid(x)


class ColorFactorMLPGaussian(OpacityFactorMLP):
    extra = 3


class ColorFactorMLPNerf(OpacityFactorMLP):
    extra = 3


class DepthFactorMLPNerf(_Head):
    extra = 3

    def act(self, x):
        return torch.relu(x)


class ImgFeatResize1(DepthFactorMLPNerf):
    pass


class ImgFeatResize2(DepthFactorMLPNerf):
    pass


class VoxelFeatureExtractor(nn.Module):
    def __init__(self, in_planes=1, out_planes=13):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(in_planes, out_planes, kernel_size=1), nn.BatchNorm3d(out_planes),
                                  nn.ReLU(inplace=True))

    def forward(self, x):
        conv, bn = self.conv[0], self.conv[1]
        if (x.is_cuda and x.dim() == 5 and x.shape[1] == 1 and conv.in_channels == 1 and conv.kernel_size == (1, 1, 1)
                and type(bn) is nn.BatchNorm3d and bn.affine and bn.track_running_stats):
            # (a converted nn.SyncBatchNorm needs cross-rank statistics: it takes the layer path below)
            return self._lift(x)
        return self.conv(x)

    def _lift(self, x):
        """Conv3d(1 -> Zh, k = 1) + BatchNorm3d + ReLU in closed form: channel h of the convolution is
        ``w_h x + b_h``, so its batch statistics are ``w_h mean(x) + b_h`` and ``w_h^2 var(x)`` and the three layers
        are ONE broadcast multiply-add + ReLU.  MIOpen runs the 1-channel Conv3d and the BatchNorm over the
        (B,Zh,Y,X,C) tensor at 0.8 + 2.0 ms and 4.0 + 2.8 ms (forward + backward) at 200x200."""
        conv, bn = self.conv[0], self.conv[1]
        w, b = conv.weight.reshape(-1), conv.bias
        if b is None:
            b = torch.zeros_like(w)
        if bn.training:
            n = x.numel()
            var, mean = torch.var_mean(x, unbiased=False)
            mean_h, var_h = w * mean + b, w * w * var
            with torch.no_grad():
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - mom).add_(mean_h, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var_h * (n / max(n - 1, 1)), alpha=mom)
        else:
            mean_h, var_h = bn.running_mean, bn.running_var
        a = bn.weight * torch.rsqrt(var_h + bn.eps)
        shape = (1, -1, 1, 1, 1)
        out = torch.addcmul((a * (b - mean_h) + bn.bias).view(shape), x, (a * w).view(shape))
        return out.relu_()


class ResizeNetwork(nn.Module):
    """256 -> output_channel, x16 up-sampling, purely linear (:534-554)."""

    def __init__(self, output_channel):
        super().__init__()
        self.conv1 = nn.Conv2d(256, 128, kernel_size=3, padding=1)
        self.upsample1 = nn.ConvTranspose2d(128, 64, kernel_size=2, stride=2)
        self.conv2 = nn.Conv2d(64, 32, kernel_size=3, padding=1)
        self.upsample2 = nn.ConvTranspose2d(32, output_channel, kernel_size=2, stride=2)
        self.upsample3 = nn.ConvTranspose2d(output_channel, output_channel, kernel_size=4, stride=4)

    def stem(self, x):
        """Everything up to conv2's output — what the composed NeRF kernels consume."""
        return self.conv2(self.upsample1(self.conv1(x)))

    def forward(self, x):
        return self.upsample3(self.upsample2(self.stem(x)))


class _LinearWeighted(nn.Module):
    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.tensor(0.5))

    def forward(self, a1, a2):
        if a1.is_cuda and not torch.is_grad_enabled():
            return torch.lerp(a2, a1, self.w)       # a2 + w (a1 - a2): one launch instead of four
        return self.w * a1 + (1 - self.w) * a2


class LinearWeightedImage(_LinearWeighted):
    pass


class LinearWeightedDepth(_LinearWeighted):
    pass


class LearnedPositionalEncoding(nn.Module):
    """mmdet 2.x ``LearnedPositionalEncoding`` (not in the reference tree; imported at :12):
    ``forward(mask (B,H,W)) -> (B, 2*num_feats, H, W)`` = cat(col_embed(x) over rows, row_embed(y)
    over columns)."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50):
        super().__init__()
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        self.num_feats, self.row_num_embed, self.col_num_embed = num_feats, row_num_embed, col_num_embed
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, mask):
        h, w = mask.shape[-2:]
        xe = self.col_embed(torch.arange(w, device=mask.device))
        ye = self.row_embed(torch.arange(h, device=mask.device))
        pos = torch.cat((xe.unsqueeze(0).expand(h, -1, -1), ye.unsqueeze(1).expand(-1, w, -1)), -1)
        return pos.permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)


# ------------------------------------------------------------------------------------------------
# the neck
# ------------------------------------------------------------------------------------------------
class _Geometry:
    """Everything ``view_transform_core`` derives from the calibration alone: both rank-vector sets,
    voxel centres, pillar projections and their validity (view_transformer.py:108-147,197-255;
    view_transformer_ocrf.py:651-740,785-852)."""
    __slots__ = ('lss', 'ht', 'voxel', 'pix', 'mask', 'calib', 'c2w', 'cam_rows', 'plans', 'cam_rows_dev', 'rank_vectors',
                 'raster_plan')


class OcRFViewTransformerFull(nn.Module):
    def __init__(self, pc_range, bev_h=128, bev_w=128, num_height=13, collapse_z=True, loss_semantic_weight=25,
                 depth_threshold=1, semantic_threshold=0.25, depthnet_cfg=dict(), grid_config=None, input_size=None,
                 downsample=16, in_channels=512, out_channels=64, accelerate=False, loss_depth_weight=3.0,
                 depth_net=None):
        super().__init__()
        if grid_config is None or input_size is None:
            raise TypeError('grid_config and input_size are required (view_transformer.py:36-44)')
        # LSSViewTransformer (view_transformer.py:36-76)
        self.grid_config, self.downsample, self.input_size = grid_config, downsample, input_size
        self.grid_lower_bound, self.grid_interval, self.grid_size = index_prep.grid_infos(grid_config)
        self.frustum = index_prep.create_frustum(grid_config['depth'], input_size, downsample)
        self.D = self.frustum.shape[0]
        self.in_channels, self.out_channels = in_channels, out_channels
        self.accelerate, self.initial_flag = accelerate, True
        self.loss_depth_weight = loss_depth_weight
        # :581-589
        self.loss_semantic_weight = loss_semantic_weight
        self.depth_threshold = depth_threshold / self.D
        self.semantic_threshold = semantic_threshold
        self.pc_range, self.bev_h, self.bev_w = pc_range, bev_h, bev_w
        self.num_height, self.collapse_z = num_height, collapse_z
        self.depth_net = depth_net if depth_net is not None else self._reference_depth_net(depthnet_cfg)
        self.fuser = DualFeatFusion(2 * out_channels, out_channels)
        self.geom_att = BEVGeomAttention()
        self.ObatinOpacityMask = hoa.ObatinOpacityMask()
        self.prob = ProbNet(in_channels=out_channels, with_centerness=True, bev_size=(bev_h, bev_w))
        self.positional_encoding = LearnedPositionalEncoding(out_channels // 2, bev_h, bev_w)
        self.positional_encoding1 = LearnedPositionalEncoding(8 // 4, bev_h, bev_w)
        # NeRF branch (:598-610)
        dim, hidden = 80, 4
        self.image_feat_resize = ResizeNetwork(dim)
        self.sigma = nn.Sequential(nn.Linear(dim, hidden), nn.Linear(hidden, 1), nn.Softplus())
        self.C_MLP_nerf = ColorFactorMLPNerf(dim, hidden, 3)
        self.D_MLP_nerf = DepthFactorMLPNerf(dim, hidden, 1)
        self.img_feat_resize1 = ImgFeatResize1(dim, hidden, 3)
        self.img_feat_resize2 = ImgFeatResize2(dim, hidden, 1)
        # Gaussian heads (:612-623)
        self.S_MLP = ScaleFactorMLP(dim, hidden, 3)
        self.R_MLP = RotationFactorMLP(dim, hidden, 4)
        self.A_MLP = OpacityFactorMLP(dim, hidden, 1)
        self.C_MLP = ColorFactorMLPGaussian(dim, hidden, 3)
        self.OpacityVoxelToBEV = hoa.OpacityVoxelToBEVConverter(input_channel=13)
        self.color_crit = nn.MSELoss(reduction='mean')
        self.zfar, self.znear, self.trans, self.scale = 999.9, 0.01, [0.0, 0.0, 0.0], 1.0
        self.ObtainVoxelFeature = VoxelFeatureExtractor()
        self.LinearWeightedImage, self.LinearWeightedDepth = LinearWeightedImage(), LinearWeightedDepth()
        self.defor_cross_attention = hoa.DeformableAttention2D(
            dim=13, dim_head=8, heads=1, dropout=0.1, downsample_factor=4, offset_scale=4, offset_groups=None,
            offset_kernel_size=6)
        self._geo, self._tmpl, self._bg = None, None, None
        # cached geometry (accelerate=True): render through a static plan with an on-device guard (see _render_sets);
        # the plan's extent bound = margin x the largest Gaussian extent of the forward that builds it
        self.render_plan, self.render_plan_margin = True, 2.0
        # forward-only mode: per-forward calibration algebra on the GPU (ocrf_geometry_blocks) when the calibration
        # tensors arrive there — no device read-back, no synchronisation in the forward (see ``_geometry``).
        # None (default, round 6): automatically, whenever the seven calibration tensors are CUDA tensors — eval-mode /
        # forward-only (what a reference config gets without edits: 1.75 -> 1.09 ms per forward at cfg2) and, with
        # per-forward geometry, under autograd as well (training: no read-back per iteration, 17.3 -> 15.2 ms); False: the host formulation, whose rank vectors are the reference's bit for bit (the device algebra is
        # ~1 ulp off: a 1e-5 fraction of border points may change cell, tests/test_device_geometry_gpu.py); True: as None
        self.device_geometry = None
        # guard of the cached render plan (accelerate=True): 'device' — exact whatever the scale head emits, the per-call
        # pipeline armed behind every render (four near-empty launches, ~24 us at cfg2); 'host' — only a status bit is
        # raised, ``check_render()`` (one synchronising read) turns it into an exception: the caller decides when to pay
        self.render_guard = 'device'
        # training mode: the voxel lift + four Gaussian heads by the fused forward / backward kernels (csrc/neck_train.hip)
        # instead of torch layers on the (B,Zh,Y,X,C) voxel feature; False keeps the reference's layer formulation
        self.fused_heads_training = True
        # eval-mode strands on side HIP streams (see _core_fused); off by default: a caller that runs the
        # module under its own stream discipline should opt in
        self.parallel_branches, self._transient = False, hoa._LaunchCache()     # streams: not module state
        self.fork_c_after = 'pools'
        self.fork_ht_prep = True         # per-forward geometry: HT preparation + pooling beside the LSS pair (see _core_segments)
        self._rank_bufs = (index_prep._RankBuffers(), index_prep._RankBuffers())
        self._packs = {}

    def _recording(self, *tensors):
        if not torch.is_grad_enabled():
            return False
        return (any(t is not None and t.requires_grad for t in tensors)
                or any(p.requires_grad for p in _tensors_of(self, 'all', (self,), buffers=False)))

    def _reference_depth_net(self, cfg):
        try:
            from mmdet3d.models.necks.view_transformer import DepthNet
        except Exception:
            return None           # forward() then raises; view_transform_core() does not need it
        return DepthNet(self.in_channels, self.in_channels, self.out_channels, self.D + 2, **cfg)

    # -------------------------------------------------------------------------------- geometry
    def pre_compute(self, input):
        """``accelerate=True``: the geometry of the first call is kept (:854-866; the reference's own
        version of this path does not survive ``get_ht_bev_feat``)."""
        if self.initial_flag or self._geo is None:
            self._geo = self._geometry(input)          # trimmed ranks: usable by both the fused and the autograd pooling
            self.initial_flag = False

    def _templates(self, dev):
        """Calibration-independent templates on the device (frustum :77-106, normalised pillar grid
        :651-673), built once."""
        if self._tmpl is None or self._tmpl[0].device != dev:
            ref = index_prep.get_reference_points_3d(self.bev_h, self.bev_w, bs=1, num_points_in_pillar=self.num_height,
                                                     device='cpu')[0]
            self._tmpl = (self.frustum.to(dev).contiguous(), ref.to(dev).contiguous())
        return self._tmpl

    def _geometry(self, input, sync=True, lazy_ranks=False, fresh=False):
        """``sync=False`` (eval mode): the rank vectors stay at their capacity with their lengths on the
        device — nothing between the calibration and the pooled BEV reads the device.  Only that forward-only
        path writes into the module's persistent rank buffers: with ``sync=True`` the ranks reach autograd
        Functions that save them for the backward (``QuickCumsumCuda`` / ``_FusedPool``; ``.int().contiguous()``
        is a no-op on them), and a later forward of the same module before that backward (adjacent frame,
        second view, eval hook) must not overwrite what was saved — those calls get fresh tensors."""
        x = input[0]
        dev = x.device
        B, N, _, Hf, Wf = x.shape
        geo = _Geometry()
        geo.cam_rows, geo.plans, geo.cam_rows_dev = {}, {}, None
        geo.raster_plan = None
        on_dev = all(torch.is_tensor(t) and t.is_cuda for t in list(input[1:7]) + [input[11]])
        if getattr(self, 'device_geometry', None) is not False and not sync and on_dev:
            # forward-only path with the calibration already on the GPU: the tiny per-camera algebra runs there
            # too (ocrf_geometry_blocks) — nothing in the forward reads the device or waits for it.  ~1 ulp from
            # the host formulation below, which stays the default (and the one the rank fixtures pin bit for bit)
            lss_block, ht_block, geo.cam_rows_dev = index_prep.geometry_blocks_hip(
                *input[1:7], input[11], self.input_size, self.znear, self.zfar)
            geo.calib = geo.c2w = None
        else:
            # the per-camera 3x3 algebra on the host, as the same torch calls the reference makes
            # (a handful of (B,N,3,3) tensors; the reference moves them to the host itself, :1086-1088)
            host = self._to_host(list(input[1:7]) + [input[11]])
            calib = host[:6]
            geo.calib, geo.c2w = calib, host[6]
            lss_block = index_prep.lss_camera_block(*calib).to(dev)
            lidar2img, img_aug, _, _ = index_prep.get_projection(*calib)
            ht_block = index_prep.ht_camera_block(lidar2img, img_aug).to(dev)
        gx, gy, gz = (int(v) for v in self.grid_size.tolist())
        frustum, tmpl = self._templates(dev)

        def rank_vectors(which=None):
            lss = ht = None
            if which in (None, 'lss'):
                lss = index_prep.voxel_pooling_prepare_v2_hip(frustum, lss_block, B, N, self.grid_lower_bound,
                                                              self.grid_interval, self.grid_size,
                                                              buffers=None if (self.accelerate or sync or fresh) else self._rank_bufs[0],
                                                              sync=sync)
            if which in (None, 'ht'):
                ht = index_prep.fast_sample_prepare_hip(tmpl, ht_block, B, N, list(self.pc_range), self.input_size,
                                                        self.grid_config['depth'], Wf, Hf, self.D,
                                                        buffers=None if (self.accelerate or sync or fresh) else self._rank_bufs[1],
                                                        sync=sync)
            return lss, ht
        geo.pix, geo.mask, geo.voxel = index_prep.ht_project_hip(tmpl, ht_block, B, N, list(self.pc_range),
                                                                 self.input_size, self.grid_config['depth'])
        if lazy_ranks:
            # per-forward geometry on the fused path: the ~20 launches of the two index preparations are issued by
            # _core_fused on the poolings' own strand, AFTER the colour / NeRF strand (which needs only the projections
            # above) has been forked — beside it instead of in front of everything
            geo.lss = geo.ht = None
            geo.rank_vectors = rank_vectors
        else:
            geo.lss, geo.ht = rank_vectors()
        return geo

    @staticmethod
    def _to_host(tensors):
        """float32 host copies of the calibration tensors.  Device tensors leave in ONE packed read-back (the
        reference reads them one by one, :1086-1088; each read is a device synchronisation); host tensors pass
        through untouched — a caller that still has the dataloader's host copies can hand those in and the
        forward does not synchronise at all."""
        ts = [t.detach().float() for t in tensors]
        on_dev = [t for t in ts if t.is_cuda]
        if len(on_dev) > 1:
            flat = torch.cat([t.reshape(-1) for t in on_dev]).cpu()
            parts = iter(flat.split([t.numel() for t in on_dev]))
            return [next(parts).view(t.shape) if t.is_cuda else t for t in ts]
        return [t.cpu() for t in ts]

    def _camera(self, geo, bs, cam_idx):
        """``data`` dict of the render call (:1135-1152), quirks included: unscaled intrinsics with the
        network-input viewport, ``c2w`` fed where a world->view transform is expected.  Host tensors."""
        return gaussian_renderer.camera_from_calibration(geo.calib[2][bs, cam_idx].numpy(), geo.c2w[bs, cam_idx].numpy(),
                                                         self.input_size[0], self.input_size[1], znear=self.znear,
                                                         zfar=self.zfar)

    def stage_cameras(self, geo, cam_idx_list, device, out=None):
        """Everything the device needs to know about the random camera choice (:1081): ``cam_sel``
        (B) int32 and the rasteriser's packed camera rows (B,36).  The rows are cached per (sample,
        camera) for the lifetime of ``geo``; with ``out`` (a dict from an earlier call) the values are
        copied into its tensors in place — the static inputs of a captured graph."""
        if getattr(geo, 'cam_rows_dev', None) is not None:
            # device geometry: the rows of every camera-frame are already on the GPU; pick the chosen ones there
            sel_d = self._small_h2d(torch.tensor(list(cam_idx_list), dtype=torch.int32), device)
            packed = geo.cam_rows_dev[torch.arange(len(cam_idx_list), device=device), sel_d.long()].contiguous()
            if out is None:
                return dict(cam_sel=sel_d, packed=packed, cam_idx_list=list(cam_idx_list))
            out['cam_sel'].copy_(sel_d)
            out['packed'].copy_(packed)
            out['cam_idx_list'] = list(cam_idx_list)
            return out
        rows = []
        for bs, c in enumerate(cam_idx_list):
            if (bs, c) not in geo.cam_rows:
                cam = self._camera(geo, bs, c)
                geo.cam_rows[(bs, c)] = pack_cameras(cam['world_view_transform'][None], cam['full_proj_transform'][None],
                                                     math.tan(float(cam['FovX']) * 0.5), math.tan(float(cam['FovY']) * 0.5),
                                                     self.input_size[0], self.input_size[1], 'cpu')
            rows.append(geo.cam_rows[(bs, c)])
        packed = torch.cat(rows, 0)
        sel = torch.tensor(list(cam_idx_list), dtype=torch.int32)
        n = len(cam_idx_list)
        if out is None:
            # ONE device buffer behind both tensors (the B x 36 rows, then B int32 words): a refresh is one copy
            buf = torch.empty(n * 37, dtype=torch.float32, device=device)
            host = torch.empty(n * 37, dtype=torch.float32)
            host[:36 * n].copy_(packed.reshape(-1))
            host[36 * n:].view(torch.int32).copy_(sel)
            buf.copy_(host)
            return dict(cam_sel=buf[36 * n:].view(torch.int32), packed=buf[:36 * n].view(n, 36),
                        cam_idx_list=list(cam_idx_list), _buf=buf)
        # static device tensors of a captured graph: refresh through PINNED host slots (an asynchronous
        # copy out of a pageable temporary may still be reading it after the temporary is gone), each
        # slot guarded by an event so that it is not rewritten while its copy is in flight
        ring = out.setdefault('_ring', [])
        if not ring:
            for _ in range(4):
                ring.append([torch.empty(n * 37, dtype=torch.float32).pin_memory(), None])
            out['_slot'] = 0
        slot = ring[out['_slot']]
        out['_slot'] = (out['_slot'] + 1) % len(ring)
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:36 * n].copy_(packed.reshape(-1))
        slot[0][36 * n:].view(torch.int32).copy_(sel)
        out['_buf'].copy_(slot[0], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(device))
        out['cam_idx_list'] = list(cam_idx_list)
        return out

    def _small_h2d(self, host_tensor, device):
        """A few host integers to the device without a hidden wait: through a ring of PINNED slots (an asynchronous
        copy out of pageable memory blocks the host until the stream reaches it), each guarded by an event."""
        # (kept in ``_transient``: pinned slots and events are not module state — ``copy.deepcopy(module)`` after a forward
        # on the device-geometry path, now the default, used to fail on the events)
        ring = self._transient.get('pin_ring')
        if ring is None:
            ring = self._transient['pin_ring'] = dict(
                slots=[[torch.empty(64, dtype=torch.int32).pin_memory(), None] for _ in range(8)], i=0)
        slot = ring['slots'][ring['i']]
        ring['i'] = (ring['i'] + 1) % len(ring['slots'])
        if slot[1] is not None:
            slot[1].synchronize()
        n = host_tensor.numel()
        slot[0][:n].copy_(host_tensor.reshape(-1))
        out = torch.empty(n, dtype=torch.int32, device=device)
        out.copy_(slot[0][:n], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(device))
        return out

    # -------------------------------------------------------------------------------- packs
    def invalidate_packs(self):
        """Forget the cached tensor lists and weight packs (needed only after assigning a NEW Parameter object to a
        sub-module; value updates, ``load_state_dict``, ``.to()`` and ``train()`` are tracked)."""
        _drop_tensor_lists(self)
        self._packs.clear()

    def _apply(self, fn, *args, **kwargs):
        _drop_tensor_lists(self)
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        _drop_tensor_lists(self)
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        _drop_tensor_lists(self)
        return super().train(mode)

    def _pack(self, name, params, build):
        key = tuple((p._version, p.data_ptr()) for p in params)
        hit = self._packs.get(name)
        if hit is None or hit[0] != key:
            self._packs[name] = (key, build())
        return self._packs[name][1]

    def _head_params(self):
        mods = (self.ObtainVoxelFeature, self.S_MLP, self.R_MLP, self.A_MLP, self.C_MLP)
        return self._pack('heads', _tensors_of(self, 'heads', mods), lambda: neck_ops.pack_gauss_head_params(*mods))

    def _nerf_params(self):
        mods = (self.image_feat_resize, self.sigma, self.C_MLP_nerf, self.img_feat_resize1, self.img_feat_resize2)
        return self._pack('nerf', _tensors_of(self, 'nerf', mods, buffers=False),
                          lambda: neck_ops.compose_nerf_maps(*mods))

    def _pos(self, name, batch, like):
        """Eval-mode positional-encoding maps are constants of the weights: built once per (weights, batch)
        instead of arange + 2 embeddings + cat + an 80-channel strided ``repeat`` per forward."""
        enc = getattr(self, name)
        return self._pack((name, batch, like.dtype), _tensors_of(self, name, (enc,), buffers=False),
                          lambda: enc(_zeros((batch, self.bev_h, self.bev_w), like)).to(like.dtype).contiguous())

    # -------------------------------------------------------------------------------- pooling
    def _pool(self, ranks, depth, feat_cl, bev_shape, scratch_tag='bev_pool_nchw'):
        if len(ranks) == 2:                       # ((five capacity vectors), device counts)
            (rb, rd, rf, st, ln), counts = ranks
            if torch.is_grad_enabled() and (depth.requires_grad or feat_cl.requires_grad):
                return bevpool.bev_pool_v2_device_counts_autograd(depth, feat_cl, rd, rf, rb, bev_shape, st, ln, counts,
                                                                  scratch_tag=scratch_tag)
            return bevpool.bev_pool_v2_device_counts(depth, feat_cl, rd, rf, rb, bev_shape, st, ln, counts,
                                                     scratch_tag=scratch_tag)
        if ranks[0] is None:
            B, Z, Y, X, C = bev_shape
            return torch.zeros(B, Z * C, Y, X, device=depth.device)
        rb, rd, rf, st, ln = ranks
        return bevpool.bev_pool_v2_collapsed(depth, feat_cl, rd, rf, rb, bev_shape, st, ln)

    def _pool_cached(self, geo, which, ranks, depth, feat_cl, bev_shape):
        """Cached geometry + forward-only call: the rank-only half of the pooling is cached as well."""
        if (geo is self._geo and len(ranks) == 5 and ranks[0] is not None and not torch.is_grad_enabled()
                and bevpool._fusable(bev_shape[-1])):
            plan = geo.plans.get(which)
            panel = bev_shape[-1] in (64, 80, 96, 128)
            if plan is None or plan.shape != tuple(int(v) for v in bev_shape):
                rb, rd, rf, st, ln = ranks
                if panel:
                    # the same back end as HotPath's default: the panel latency kernel (csrc/bev_pool_panel.hip), 13-30 %
                    # faster than the tile kernel; heavy LSS tiles cut by estimated cost
                    plan = bevpool.MfmaPoolPlan(rd, rf, rb, bev_shape, group=8, unit_cost=8.0 if which == 'lss' else None)
                else:
                    plan = bevpool.DevicePoolPlan(rd, rf, rb, bev_shape, st, ln)
                geo.plans[which] = plan
            if panel:
                return bevpool.bev_pool_v2_panel(depth, feat_cl, plan)
            return bevpool.bev_pool_v2_planned(depth, feat_cl, plan)
        return self._pool(ranks, depth, feat_cl, bev_shape)

    def get_lss_bev_feat(self, geo, depth, feat_cl):
        gx, gy, gz = (int(v) for v in self.grid_size.tolist())
        return self._pool_cached(geo, 'lss', geo.lss, depth, feat_cl, (depth.shape[0], gz, gy, gx, feat_cl.shape[-1]))

    def get_ht_bev_feat(self, geo, depth, feat_cl):
        return self._pool_cached(geo, 'ht', geo.ht, depth, feat_cl,
                                 (depth.shape[0], 1, self.bev_h, self.bev_w, feat_cl.shape[-1]))

    # -------------------------------------------------------------------------------- core
    def view_transform(self, input, depth, tran_feat, feat_channels_last=None, cameras=None):
        if self.accelerate:
            self.pre_compute(input)
        return self.view_transform_core(input, depth, tran_feat, feat_channels_last, cameras=cameras)

    def view_transform_core(self, input, depth, tran_feat, feat_channels_last=None, cam_idx_list=None, cameras=None):
        """Same inputs / outputs as the reference (:1040-1201).  ``feat_channels_last`` (B*N,H,W,C),
        when the caller already has it (``forward`` does), skips the permute of :875/:901;
        ``cam_idx_list`` overrides the random camera choice of :1081 (tests); ``cameras`` is a dict from
        ``stage_cameras`` (choice already made and staged on the device: nothing in the call then touches
        the host, so the whole call can be captured in a hipGraph)."""
        x = input[0]
        B, N, _, Hf, Wf = x.shape
        imgs_wo_norm, dtype = input[9], x.dtype
        C, Zh, Y, X = self.out_channels, self.num_height, self.bev_h, self.bev_w
        H, W = self.input_size
        # fused HIP kernels are forward-only and fold the BatchNorm running statistics: eval mode with
        # nothing recorded by autograd; anything else takes the differentiable torch formulation
        fused = not self.training and not self._recording(depth, tran_feat, feat_channels_last)
        # Under autograd too the geometry stays on the device when the calibration is there (``device_geometry`` not False):
        # the rank vectors at their capacity with device-side lengths, in tensors of THIS forward (an autograd Function saves
        # them), the poolings differentiated by ``_FusedPoolCounts``, the cameras staged from the device rows — a training
        # forward then reads nothing back (the reference synchronises at :1086-1088, and this module did, once per forward,
        # until round 6: the GPU's tail of every iteration was exposed, ~1.5-2.4 ms at cfg2)
        on_dev = all(torch.is_tensor(t) and t.is_cuda for t in list(input[1:7]) + [input[11]])
        dev_train = not fused and self.device_geometry is not False and on_dev and not self.accelerate
        geo = self._geo if (self.accelerate and self._geo is not None) else \
            self._geometry(input, sync=not (fused or dev_train), lazy_ranks=fused and not self.accelerate, fresh=dev_train)
        depth5 = depth.reshape(B, N, self.D, Hf, Wf).float()
        if feat_channels_last is None:
            feat_channels_last = tran_feat.reshape(B, N, C, Hf, Wf).permute(0, 1, 3, 4, 2)
        feat_cl = feat_channels_last.reshape(B, N, Hf, Wf, C).float().contiguous()
        if cameras is None:
            if cam_idx_list is None:
                cam_idx_list = [random.randint(0, 5) for _ in range(B)]
            cameras = self.stage_cameras(geo, cam_idx_list, x.device)
        cam_idx_list, cam_sel = cameras['cam_idx_list'], cameras['cam_sel']
        if cameras.get('packed') is None:
            # only the choice was staged (a captured per-forward-geometry graph): the rasteriser's rows of the chosen
            # cameras are picked on the device from the rows ocrf_geometry_blocks made for every camera-frame
            cameras = dict(cameras, packed=geo.cam_rows_dev[torch.arange(B, device=x.device), cam_sel.long()].contiguous())
        voxel_coor = geo.voxel.reshape(B, Zh * Y * X, 3)
        if fused:
            return self._core_fused(input, geo, depth, depth5, feat_cl, cameras, voxel_coor)
        lss_feat = self.get_lss_bev_feat(geo, depth5, feat_cl)
        ht_feat = self.get_ht_bev_feat(geo, depth5, feat_cl)
        (opacity, scaling, rotation, color, sparse, alpha_lidar, render_N,
         render_depth_N) = self._neck_torch(input, geo, ht_feat, cam_idx_list, cam_sel)
        # one differentiable render per sample (:1135-1153) from the camera rows staged on the device above — the
        # reference's per-sample camera set-up (matrix uploads, focal arithmetic: ~0.9 ms of host time per sample here)
        # was already done once, for the whole batch, by stage_cameras
        if self._bg is None or self._bg.device != x.device:
            self._bg = torch.zeros(3, device=x.device)
        render_G, render_depth_G = [], []
        for bs in range(B):
            img, dep, _, _ = rasterize_packed_autograd(voxel_coor[bs], color[bs], opacity[bs], scaling[bs], rotation[bs],
                                                       cameras['packed'][bs:bs + 1], H, W, self._bg)
            render_G.append(img), render_depth_G.append(dep)
        render_image_G_all, render_depth_G_all = torch.cat(render_G), torch.cat(render_depth_G)
        render_image = self.LinearWeightedImage(render_image_G_all, render_N)
        render_depth = self.LinearWeightedDepth(render_depth_G_all, render_depth_N)
        gt_images = imgs_wo_norm[torch.arange(B, device=x.device), cam_sel.long()] / 255.0

        # HOA-1 for the whole batch (the reference loops samples, :1159-1161)
        opacity_alpha = hoa.hoa1(self.defor_cross_attention, opacity.reshape(-1, 1), alpha_lidar, Zh, Y, X)

        channel_feat = self.fuser(lss_feat, ht_feat)
        zeros = _zeros((B, Y, X), x)            # positional encodings only read its shape / device
        bev_mask_logit = self.prob(self.positional_encoding(zeros).to(dtype) + channel_feat)
        geom_feat = self.geom_att.gate(channel_feat, bev_mask_logit)
        opacity_alpha_view = self.OpacityVoxelToBEV(opacity_alpha, self.positional_encoding1(zeros).to(dtype))
        m = torch.sigmoid(_spatial_logits(geom_feat, self.ObatinOpacityMask.conv) + opacity_alpha_view)
        geom_feat = geom_feat * m
        return geom_feat, depth, bev_mask_logit, [render_image, gt_images, render_image_G_all, render_N,
                                                  opacity_alpha_view, cam_idx_list, render_depth, render_depth_G_all,
                                                  render_depth_N]

    def _core_segments(self, input, geo, depth, depth5, feat_cl, cameras, voxel_coor, mark=lambda i: None):
        """The eval-mode step on the HIP kernels as its SEGMENTS: closures that read and write one dict of tensors, each
        running on whatever stream is current.  Who needs whom:
            b1 colours            <- inputs                       a1 poolings            <- inputs
            b2 NeRF branch, alpha <- inputs                       c  fusion, ProbNet, GA <- a1
            a2 Gaussian heads     <- a1, b1                       a3 render              <- a2
            a4 weighted images    <- a3, b2                       a5 HOA-1/2             <- a2, b2
            a6 HOA-3 gate (the module's main output)              <- c, a5
        ``_core_fused`` runs them on up to three streams in one pass (what the captured graph replays).  (Every segment
        as a linear graph of its own, replayed on four explicit streams with events between them, was measured too:
        0.77 - 0.84 ms against the one graph's 0.68 — every graph launch on the critical chain adds ~ 25 us of
        latency; profiles/r5_neck_schedule_ab.txt.)
        -> (segments, result) with result() the module's return value."""
        x, imgs_wo_norm = input[0], input[9]
        dev = x.device
        B, N, _, Hf, Wf = x.shape
        Zh, Y, X = self.num_height, self.bev_h, self.bev_w
        H, W = self.input_size
        cam_idx_list, cam_sel = cameras['cam_idx_list'], cameras['cam_sel']
        T = {}

        def b1():
            T['avg_rgb'] = neck_ops.pillar_sample_mean(imgs_wo_norm, geo.pix, geo.mask)             # (B,Zh,YX,3)
            mark(1)

        def b2():
            sparse = neck_ops.retain_valid_pixels(imgs_wo_norm, geo.pix, geo.mask, cam_sel)        # (B,3,H,W)
            w_s, c_s, nerf_block = self._nerf_params()
            z = self.image_feat_resize.stem(x.reshape(B * N, -1, Hf, Wf).float())
            alpha = neck_ops.nerf_alpha(z, w_s, c_s)                                                # (B*N,H,W)
            T['render_N'], T['render_depth_N'] = neck_ops.nerf_render(z, cam_sel, alpha, sparse, nerf_block, N)
            # the reference views the (6,H,W,1) stack as (1,6,1,W,H) before sampling it (:1123)
            alpha_lidar = neck_ops.pillar_sample_mean(alpha.view(B, N, 1, H, W), geo.pix, geo.mask, view_hw=(W, H))
            T['alpha_lidar'] = alpha_lidar.view(B, Zh, Y, X)
            T['gt_images'] = imgs_wo_norm[torch.arange(B, device=dev), cam_sel.long()] / 255.0
            mark(2)

        def a1():
            if geo.lss is None and self.parallel_branches and self.fork_ht_prep:
                # per-forward geometry: each pooling right behind its own index preparation, the HT pair on a stream of
                # its own beside the LSS pair (the preparations' scratch is per stream, the poolings' per tag): in a row
                # the four are 89 + 62 + 54 + 50 us in front of everything that reads a pooled BEV
                cur = torch.cuda.current_stream(dev)
                s_ht = self._transient.get('stream_ht')
                if s_ht is None or s_ht.device != dev:
                    s_ht = self._transient['stream_ht'] = torch.cuda.Stream(dev)
                s_ht.wait_stream(cur)
                with torch.cuda.stream(s_ht):
                    geo.ht = geo.rank_vectors('ht')[1]
                    T['ht_feat'] = self._pool(geo.ht, depth5, feat_cl, (depth5.shape[0], 1, self.bev_h, self.bev_w,
                                                                        feat_cl.shape[-1]), scratch_tag='bev_pool_nchw_b')
                geo.lss = geo.rank_vectors('lss')[0]
                T['lss_feat'] = self.get_lss_bev_feat(geo, depth5, feat_cl)
                cur.wait_stream(s_ht)
                mark(3)
                return
            if geo.lss is None:
                # per-forward geometry: beside strand B (see _geometry), both preparations on this stream
                geo.lss, geo.ht = geo.rank_vectors()
            T['lss_feat'] = self.get_lss_bev_feat(geo, depth5, feat_cl)
            T['ht_feat'] = self.get_ht_bev_feat(geo, depth5, feat_cl)
            mark(3)

        def c():
            # the fused map and the map ProbNet reads (fused map + positional encoding) leave the fusion kernel together
            channel_feat, with_pos = self.fuser(T['lss_feat'], T['ht_feat'], addend=self._pos('positional_encoding', B, x))
            mark(4)
            T['bev_mask_logit'] = self.prob(with_pos)
            mark(5)
            T['geom_feat'] = self.geom_att.gate(channel_feat, T['bev_mask_logit'])
            mark(6)

        def a2():
            T['opacity'], T['scaling'], T['rotation'], T['color'] = neck_ops.gauss_heads(
                T['ht_feat'], T['avg_rgb'], self._head_params(), Zh)
            mark(7)

        def a3():
            # every sample is a Gaussian set with one camera: ONE rasteriser call for the batch, fed with the
            # staged camera rows (the reference loops samples, :1090-1153)
            if self._bg is None or self._bg.device != dev:
                self._bg = torch.zeros(3, device=dev)
            o = self._render_sets(geo, voxel_coor, T['color'], T['opacity'], T['scaling'], T['rotation'], cameras, H, W)
            T['render_image_G_all'], T['render_depth_G_all'] = o['color'], o['depth']
            mark(8)

        def a4():
            T['render_image'] = self.LinearWeightedImage(T['render_image_G_all'], T['render_N'])
            T['render_depth'] = self.LinearWeightedDepth(T['render_depth_G_all'], T['render_depth_N'])
            mark(9)

        def a5():
            # HOA-1 for the whole batch (the reference loops samples, :1159-1161), HOA-2
            oa = hoa.hoa1(self.defor_cross_attention, T['opacity'].reshape(-1, 1), T['alpha_lidar'], Zh, Y, X)
            mark(10)
            T['opacity_alpha_view'] = self.OpacityVoxelToBEV(oa, self._pos('positional_encoding1', B, x))
            mark(11)

        def a6():
            T['bev_feat'] = self.ObatinOpacityMask.gate(T['geom_feat'], T['opacity_alpha_view'])[1]     # HOA-3
            mark(12)

        def result():
            return T['bev_feat'], depth, T['bev_mask_logit'], [
                T['render_image'], T['gt_images'], T['render_image_G_all'], T['render_N'], T['opacity_alpha_view'],
                cam_idx_list, T['render_depth'], T['render_depth_G_all'], T['render_depth_N']]

        return dict(b1=b1, b2=b2, a1=a1, c=c, a2=a2, a3=a3, a4=a4, a5=a5, a6=a6), result

    def _core_fused(self, input, geo, depth, depth5, feat_cl, cameras, voxel_coor):
        """Eval-mode ``view_transform_core`` on the HIP kernels.  The step has three independent strands
        until HOA joins them — (A) poolings -> Gaussian heads -> render, (B) colour sampling + NeRF branch
        (needs only the images and the calibration), (C) BEV fusion -> ProbNet -> geometry attention (needs
        only the pooled BEVs).  With ``parallel_branches`` B and C run on two side HIP streams (captured as
        parallel branches of the graph): the many small kernels of C and the MIOpen convolutions of B fill
        the gaps of A instead of queueing behind it."""
        dev = input[0].device
        par = self.parallel_branches
        cur = torch.cuda.current_stream(dev)
        if par:
            streams = self._transient.get('streams')
            if streams is None or streams[0].device != dev:
                streams = self._transient['streams'] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
            sB, sC = streams
            sB.wait_stream(cur)                    # inputs ready; last call's consumers of our buffers done
        on = (lambda s: torch.cuda.stream(s)) if par else (lambda s: contextlib.nullcontext())
        stamps = self._transient.get('stamps')     # diagnostic timeline (tools/timeline_neck.py)
        mark = (lambda i: _lib.diag_stamp(stamps, i)) if stamps is not None else (lambda i: None)
        seg, result = self._core_segments(input, geo, depth, depth5, feat_cl, cameras, voxel_coor, mark)
        mark(0)
        # ---- strand B
        with on(sB if par else None):
            seg['b1']()
            if par:
                rgb_ready = torch.cuda.Event()
                rgb_ready.record(sB)
            seg['b2']()
        # ---- strand A, first half.  The wait for B's colours sits in FRONT of the poolings although only the
        # heads need them: ROCm 7.2's graph executor serialises the two branches forked after a node (C, A)
        # when one of them starts with a second, cross-branch dependency (tools/diag_graph_parallel.py:
        # 8.2 instead of 5 kernel times); with the dependency ahead of the fork they overlap.
        if par:
            cur.wait_event(rgb_ready)
        seg['a1']()
        def strand_c():
            if par:
                sC.wait_stream(cur)
            with on(sC if par else None):
                seg['c']()
        # Where strand C is forked: C needs only the poolings.  Round 3 forked it behind the heads (beside MIOpen's
        # convolutions of C the heads took 355 us instead of 64); with the panel poolings and the head-of-list render
        # front end of round 5 the early fork is the faster one again (tools/ab_neck_r5.py, one box session, two
        # captures each: 'pools' 0.7005 / 0.7008 ms, 'heads' 0.738 / 0.743, 'render' 0.757).  Everything else tried on
        # the captured graph replays slower (profiles/r5_neck_schedule_ab.txt): HOA-1/2 behind the heads or on B's
        # stream (0.71 - 0.86 ms, uneven from capture to capture), the wait for B's colours in front of the heads
        # instead of the poolings (0.79), the step cut along its outputs into a main chain, a Gaussian strand and a
        # render strand (0.72 - 0.75: the executor put the render strand on the main chain's queue, behind it).
        fork_at = self.fork_c_after
        if fork_at == 'pools':
            strand_c()
        seg['a2']()
        if fork_at == 'heads':
            strand_c()
        # ---- strand A, second half.  (Forking again behind the heads — render + weighted images beside
        # HOA-1/2 — needs B joined in front of the heads to keep single-parent branches, and that wait costs
        # more than the overlap returns: 0.84 vs 0.80 ms, tools/ab_neck_graph.py.)
        seg['a3']()
        if fork_at == 'render':
            strand_c()
        if par:
            cur.wait_stream(sB)
        seg['a4']()
        seg['a5']()
        if par:
            cur.wait_stream(sC)
        seg['a6']()
        return result()

    def _render_sets(self, geo, voxel_coor, color, opacity, scaling, rotation, cameras, H, W):
        """One rendered view per sample.  With CACHED geometry (``accelerate=True``) the rasteriser's calibration-only
        front end is cached too: the Gaussian means are the voxel grid and every (sample, camera) pair a fixed view, so a
        static render plan over all of them (``raster_plan.RasterPlan``) leaves two launches per forward; the sample's
        random camera is a device-side index into the plan, and the plan's extent bound is guarded ON THE DEVICE (the
        per-call pipeline is armed behind it), so the result is exact whatever the scale head emits and the forward
        stays free of host reads (graph-capturable).  Otherwise: the per-call pipeline (``rasterize_sets``)."""
        B, N = voxel_coor.shape[0], 6
        state = getattr(geo, 'raster_plan', None)
        if (state is None and self.render_plan and self.accelerate and geo is self._geo and B * N <= 32
                and not torch.cuda.is_current_stream_capturing()):
            state = geo.raster_plan = self._build_raster_plan(geo, voxel_coor, scaling, rotation, B, N, H, W)
        if not state:            # not eligible (per-forward geometry, > 32 views, samples with different means)
            return rasterize_sets(voxel_coor, color, opacity, scaling, rotation, cameras['packed'], H, W, self._bg)
        plan, base = geo.raster_plan
        item_view = (base + cameras['cam_sel'].to(torch.int32)).contiguous()
        # (sample b renders one of ITS OWN N plan views: no view is named by two sets)
        return plan.render(color, opacity, scaling, rotation, self._bg, item_view=item_view,
                           guard=getattr(self, 'render_guard', 'device'), views_disjoint=True, want_radii=False)

    def check_render(self):
        """``render_guard='host'``: synchronising check of the cached render plan's status word — raises if a render since
        the last check had a Gaussian beyond the plan's extent bound (those images are not valid)."""
        state = getattr(self._geo, 'raster_plan', None) if self._geo is not None else None
        if state:
            state[0].check()
        return True

    def _build_raster_plan(self, geo, voxel_coor, scaling, rotation, B, N, H, W):
        """-> (RasterPlan over the B*N (sample, camera) views, int32 view offsets b*N) or False when the samples do not
        share one set of means.  Runs once per cached geometry (host work + one synchronisation)."""
        from .raster_plan import RasterPlan
        dev = voxel_coor.device
        if any(not torch.equal(voxel_coor[b], voxel_coor[0]) for b in range(1, B)):
            return False
        if geo.cam_rows_dev is not None:
            rows = geo.cam_rows_dev.reshape(B * N, 36).contiguous()
        else:
            host = []
            for bs in range(B):
                for c in range(N):
                    if (bs, c) not in geo.cam_rows:
                        cam = self._camera(geo, bs, c)
                        geo.cam_rows[(bs, c)] = pack_cameras(
                            cam['world_view_transform'][None], cam['full_proj_transform'][None],
                            math.tan(float(cam['FovX']) * 0.5), math.tan(float(cam['FovY']) * 0.5), H, W, 'cpu')
                    host.append(geo.cam_rows[(bs, c)])
            rows = torch.cat(host, 0).to(dev)
        plan = RasterPlan(voxel_coor[0], rows, H, W, scales=scaling, rotations=rotation, margin=self.render_plan_margin)
        return plan, torch.arange(B, device=dev, dtype=torch.int32) * N

    def _neck_torch(self, input, geo, ht_feat, cam_idx_list, cam_sel=None):
        """Training mode: the reference's op sequence (:1051-1133) as differentiable torch ops."""
        x, imgs_wo_norm = input[0], input[9]
        B, N, _, Hf, Wf = x.shape
        Zh, Y, X = self.num_height, self.bev_h, self.bev_w
        H, W = self.input_size
        mask5 = geo.mask.unsqueeze(-1)

        def sample(imgs, h, w):
            c = imgs.shape[2]
            g = geo.pix.clone()
            g[..., 0] = (g[..., 0] / (w - 1)) * 2 - 1
            g[..., 1] = (g[..., 1] / (h - 1)) * 2 - 1
            v = F.grid_sample(imgs.reshape(B * N, c, h, w).float(), g.view(B * N, 1, Zh * Y * X, 2), align_corners=True)
            v = v.view(B, N, c, Zh, Y * X).permute(0, 1, 3, 4, 2) * mask5.float()
            cnt = mask5.sum(1).float().clamp(min=1)
            return v.sum(1) / cnt
        avg_rgb = sample(imgs_wo_norm, H, W)
        # retain_valid_pixels for the selected cameras, vectorised (:1004-1024)
        sel = cam_sel.long() if cam_sel is not None else torch.tensor(cam_idx_list, device=x.device)
        ar = torch.arange(B, device=x.device)
        pix_s, mask_s = geo.pix[ar, sel].reshape(B, -1, 2), geo.mask[ar, sel].reshape(B, -1)
        imgs_s = imgs_wo_norm[ar, sel].float()
        keep = torch.zeros(B, H * W, dtype=torch.bool, device=x.device)
        hi = max(W, H) - 1
        xi = pix_s[..., 0].long().clamp(0, min(hi, W - 1))
        yi = pix_s[..., 1].long().clamp(0, min(hi, H - 1))
        keep.scatter_(1, (yi * W + xi) * mask_s.long(), mask_s)          # masked-out points all hit pixel 0 with False
        keep[:, 0] = ((yi * W + xi == 0) & mask_s).any(1)
        sparse = torch.where(keep.view(B, 1, H, W), imgs_s, torch.full_like(imgs_s, 255.0))
        # alpha of EVERY camera image: ResizeNetwork is linear and sigma starts with two Linears, so
        # sigma[:2](upsample3(upsample2(z))) is one composed (32 -> 8x8 sub-positions) map of conv2's output z
        # (neck_ops.compose_nerf_maps; here with autograd through the composition).  The reference's
        # formulation — a (B*N,80,H,W) feature image (692 MB at 12 x 256 x 704), two transposed convolutions
        # and an 80 -> 4 -> 1 MLP on 2.2 M rows, forward and backward — is kept only for the ONE selected
        # camera of each sample, whose features the four NeRF heads need.
        res, z = self.image_feat_resize, self.image_feat_resize.stem(x.reshape(B * N, -1, Hf, Wf).float())
        h2, w2 = z.shape[-2:]
        W2, b2, W3, b3 = (t.double() for t in (res.upsample2.weight, res.upsample2.bias, res.upsample3.weight,
                                               res.upsample3.bias))
        s0, s1 = self.sigma[0], self.sigma[1]
        ws = s1.weight.double() @ s0.weight.double()                                          # (1,80)
        cs = s1.weight.double() @ s0.bias.double() + s1.bias.double()
        t3 = torch.einsum('ofrs,kf->kors', W3, ws)                                             # (1,80,4,4)
        M = torch.einsum('iopq,kors->iprqs', W2, t3).reshape(W2.shape[0], 64)                 # y%8 = 4p + r, x%8 = 4q + s
        const = torch.einsum('o,kors->rs', b2, t3) + (ws @ b3 + cs)                           # (4,4)
        pre = (z.permute(0, 2, 3, 1) @ M.to(z.dtype)).view(B * N, h2, w2, 2, 4, 2, 4)
        pre = pre + const.to(z.dtype).view(1, 1, 1, 1, 4, 1, 4)
        pre = pre.permute(0, 1, 3, 4, 2, 5, 6).reshape(B * N, H, W, 1)
        alpha = 1. - torch.exp(-self.sigma[2](pre))                                           # (B*N,H,W,1)
        alpha_lidar = sample(alpha.reshape(B, N, 1, W, H), W, H).view(B, Zh, Y, X)
        fs = res.upsample3(res.upsample2(z.view(B, N, -1, h2, w2)[ar, sel])).permute(0, 2, 3, 1)   # (B,H,W,80)
        xin = torch.cat((fs, sparse.permute(0, 2, 3, 1) / 255.0), -1)
        radiance, radiance1 = self._nerf_heads_torch(xin)
        a_sel = alpha.view(B, N, H, W, 1)[ar, sel]
        render_N = (a_sel * radiance).permute(0, 3, 1, 2)
        render_depth_N = (a_sel * radiance1).permute(0, 3, 1, 2)
        if self.fused_heads_training and self._heads_fusable(ht_feat):
            # lift + four heads forward by ocrf_gauss_heads, backward by ocrf_gauss_heads_backward: the (B,Zh,Y,X,C) voxel
            # feature (333 MB at cfg2, read or written by ~100 torch kernels of an iteration) exists in neither; the
            # modules' parameters — and ht_feat, also through the BatchNorm's batch statistics — get their gradients
            # through the differentiable packing of the kernels' parameter block
            heads = (self.S_MLP, self.R_MLP, self.A_MLP, self.C_MLP)
            prm = neck_ops.pack_gauss_head_params_autograd(self.ObtainVoxelFeature, *heads, x=ht_feat)
            opacity, scaling, rotation, color = neck_ops.gauss_heads_train(ht_feat, avg_rgb.reshape(B, Zh, Y * X, 3), prm, Zh)
            return opacity, scaling, rotation, color, sparse, alpha_lidar, render_N, render_depth_N
        voxel_feat = self.ObtainVoxelFeature(ht_feat.permute(0, 2, 3, 1).unsqueeze(1)).reshape(B, Zh * Y * X, -1)
        rgb01 = avg_rgb.reshape(B, Zh * Y * X, 3) / 255.0
        return (self.A_MLP(voxel_feat), self.S_MLP(voxel_feat), self.R_MLP(voxel_feat),
                self.C_MLP(torch.cat((voxel_feat, rgb01), -1)), sparse, alpha_lidar, render_N, render_depth_N)

    def _nerf_heads_torch(self, xin):
        """radiance = img_feat_resize1(xin) * softmax(C_MLP_nerf(xin)), radiance1 = img_feat_resize2(xin) *
        softmax(D_MLP_nerf(xin)) (:1113-1121).  The four heads read the same 83-column rows (360 448 of them at cfg2): their
        first layers run as ONE 83 -> 16 Linear and their second layers as one block-diagonal 16 -> 8 Linear — two
        split-K weight gradients instead of eight (each ~115 us of host time per backward), a third of the launches."""
        heads = (self.img_feat_resize1, self.C_MLP_nerf, self.img_feat_resize2, self.D_MLP_nerf)
        kinds = (ImgFeatResize1, ColorFactorMLPNerf, ImgFeatResize2, DepthFactorMLPNerf)
        if not (all(type(m) is k for m, k in zip(heads, kinds))
                and len({(m.fc1.in_features, m.fc1.out_features) for m in heads}) == 1
                and all(m.fc1.bias is not None and m.fc2.bias is not None for m in heads)):
            return (self.img_feat_resize1(xin) * F.softmax(self.C_MLP_nerf(xin), dim=-1),
                    self.img_feat_resize2(xin) * F.softmax(self.D_MLP_nerf(xin), dim=-1))
        w1, b1 = torch.cat([m.fc1.weight for m in heads]), torch.cat([m.fc1.bias for m in heads])
        w2, b2 = torch.block_diag(*[m.fc2.weight for m in heads]), torch.cat([m.fc2.bias for m in heads])
        rows = xin.numel() // xin.shape[-1]
        lin = _TallLinear.apply if (xin.is_cuda and rows >= 8 * _TallLinear.CHUNK) else F.linear
        out = lin(torch.relu(lin(xin, w1, b1)), w2, b2)
        n1, n2, n3 = (heads[0].fc2.out_features, heads[1].fc2.out_features, heads[2].fc2.out_features)
        feat, col, dfeat, dep = out.split((n1, n2, n3, out.shape[-1] - n1 - n2 - n3), dim=-1)
        return (torch.relu(feat) * F.softmax(torch.sigmoid(col), dim=-1),
                torch.relu(dfeat) * F.softmax(torch.relu(dep), dim=-1))

    def _heads_fusable(self, ht_feat):
        """The shapes csrc/neck_train.hip has register tiles for, and exactly the reference's layer types (a converted
        ``SyncBatchNorm`` needs cross-rank statistics and takes the layer path)."""
        vfe = self.ObtainVoxelFeature
        conv, bn = vfe.conv[0], vfe.conv[1]
        C = ht_feat.shape[1]
        heads = ((self.S_MLP, ScaleFactorMLP, C, 3), (self.R_MLP, RotationFactorMLP, C, 4),
                 (self.A_MLP, OpacityFactorMLP, C, 1), (self.C_MLP, ColorFactorMLPGaussian, C + 3, 3))
        return (ht_feat.is_cuda and ht_feat.dtype == torch.float32 and self.num_height in (1, 2, 4, 6, 8, 13)
                and type(conv) is nn.Conv3d and conv.in_channels == 1 and conv.kernel_size == (1, 1, 1)
                and conv.out_channels == self.num_height
                and type(bn) is nn.BatchNorm3d and bn.affine and bn.track_running_stats
                and all(type(m) is cls and m.fc1.in_features == cin and m.fc1.out_features == 4
                        and m.fc2.out_features == cout and m.fc1.bias is not None and m.fc2.bias is not None
                        for m, cls, cin, cout in heads))

    # -------------------------------------------------------------------------------- forward
    def forward(self, input, stereo_metas=None):
        """:1319-1334.  ``input``: the 12-entry ``img_inputs`` list."""
        if self.depth_net is None:
            raise RuntimeError('OcRFViewTransformerFull.forward needs a depth_net (pass depth_net=... or install '
                               'the reference mmdet3d); view_transform_core() runs without it')
        x, mlp_input = input[0], input[7]
        B, N, C, H, W = x.shape
        y = self.depth_net(x.view(B * N, C, H, W), mlp_input, stereo_metas)
        if y.is_cuda and not self.training and not self._recording(y):
            depth, filter_depth, semantic, feat_cl = neck_ops.prefilter(y, self.D, self.out_channels,
                                                                        self.depth_threshold, self.semantic_threshold)
            bev_feat, _, bev_mask, extras = self.view_transform(input, filter_depth, None, feat_cl)
        else:
            depth = y[:, :self.D].softmax(dim=1)
            semantic = y[:, self.D:self.D + 2].softmax(dim=1)
            tran_feat = y[:, self.D + 2:self.D + 2 + self.out_channels]
            filter_depth = torch.where(depth < self.depth_threshold, torch.zeros_like(depth), depth)
            filter_feat = (semantic[:, 1:2] >= self.semantic_threshold) * tran_feat
            bev_feat, _, bev_mask, extras = self.view_transform(input, filter_depth, filter_feat)
        return bev_feat, depth, (bev_mask, semantic), extras

    def get_mlp_input(self, rot, tran, intrin, post_rot, post_tran, bda):
        """27 camera-aware scalars per view for the DepthNet's SE layers (view_transformer.py:696-722)."""
        B, N = rot.shape[:2]
        bda = bda.view(B, 1, 3, 3).repeat(1, N, 1, 1)
        v = torch.stack([intrin[:, :, 0, 0], intrin[:, :, 1, 1], intrin[:, :, 0, 2], intrin[:, :, 1, 2],
                         post_rot[:, :, 0, 0], post_rot[:, :, 0, 1], post_tran[:, :, 0], post_rot[:, :, 1, 0],
                         post_rot[:, :, 1, 1], post_tran[:, :, 1], bda[:, :, 0, 0], bda[:, :, 0, 1], bda[:, :, 1, 0],
                         bda[:, :, 1, 1], bda[:, :, 2, 2]], dim=-1)
        s2e = torch.cat([rot, tran.reshape(B, N, 3, 1)], dim=-1).reshape(B, N, -1)
        return torch.cat([v, s2e], dim=-1)


class GraphedNeck:
    """``OcRFViewTransformerFull`` inference (pre-filter + ``view_transform``) as ONE hipGraph launch per
    call for fixed shapes — with a static calibration (``accelerate=True``: the geometry is cached outside the
    graph) or, with ``module.device_geometry`` and the calibration tensors on the GPU, with the per-forward geometry
    INSIDE the graph (``accelerate=False``, the reference's working mode: calibration algebra, both index
    preparations and everything after them replay as one launch; the calibration values are copied into the
    graph's static inputs on every call):

        neck = GraphedNeck(module, example_input, example_depthnet_out)   # captures
        bev_feat, depth, (bev_mask, semantic), extras = neck(input, depthnet_out)

    ``input`` is the 12-entry ``img_inputs`` list, ``depthnet_out`` what ``module.depth_net`` returns
    (B*N, D+2+C, H, W); their values are copied into the static buffers the graph reads (image features,
    raw images and the DepthNet output — the calibration is frozen with the geometry).  The only per-call
    host decision, the random camera of each sample (view_transformer_ocrf.py:1081), is staged through two
    small static device tensors.  The three independent strands of the step run as parallel branches of
    the graph (``_core_fused``).  Outputs are static tensors, overwritten by the next call.  Eval mode,
    forward only."""

    def __init__(self, module, example_input, example_depthnet_out, warmup=3, parallel_branches=True,
                 capture_stream=None):
        calib_on_dev = all(torch.is_tensor(example_input[i]) and example_input[i].is_cuda for i in (1, 2, 3, 4, 5, 6, 11))
        self.per_forward_geometry = not module.accelerate
        if self.per_forward_geometry and not (getattr(module, 'device_geometry', None) is not False and calib_on_dev):
            raise RuntimeError('GraphedNeck needs accelerate=True (geometry cached across calls), or the calibration '
                               'tensors on the GPU with module.device_geometry not switched off (geometry inside '
                               'the graph): the host formulation of the calibration algebra cannot be captured')
        if module.training:
            raise RuntimeError('GraphedNeck is inference only: call module.eval() first')
        self.module = m = module
        dev = example_input[0].device
        self.device = dev
        self.inputs = [t.clone() if torch.is_tensor(t) else t for t in example_input]
        self.depthnet_out = example_depthnet_out.detach().float().contiguous().clone()
        self.batch = self.inputs[0].shape[0]
        with torch.no_grad():
            self._body(None)                       # geometry, packs, workspaces, MIOpen algorithms
            if self.per_forward_geometry:
                # only the camera choice is staged; the rows follow from the in-graph geometry
                self._cams = dict(cam_sel=torch.zeros(self.batch, dtype=torch.int32, device=dev), packed=None,
                                  cam_idx_list=[0] * self.batch)
            else:
                self._cams = m.stage_cameras(m._geo, [0] * self.batch, dev)
            m.parallel_branches = parallel_branches
            try:
                side = torch.cuda.Stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(warmup):
                        self._body(self._cams)
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                try:
                    self._graph = torch.cuda.CUDAGraph(keep_graph=True)      # the raw graph stays inspectable
                    inspect = True
                except TypeError:
                    self._graph, inspect = torch.cuda.CUDAGraph(), False
                with torch.cuda.graph(self._graph, stream=capture_stream):
                    self._static_out = self._body(self._cams)
                if inspect:
                    # ROCm 7.2: a graph with memset nodes faults on replay after any hipMemcpyAsync on the stream (and
                    # every replay() stages the camera choice with one): refuse it here, loudly, instead of there
                    self.census = _lib.graph_census(self._graph)
                    if self.census['memset']:
                        raise _lib.OcrfHipError(
                            f"the captured neck graph holds {self.census['memset']} memset node(s) (a torch.zeros / "
                            '.zero_() / hipMemsetAsync inside the captured region): replaying it would fault on ROCm 7.2; '
                            'zero-fill with a kernel (e.g. tensor.fill_(0) on a non-empty tensor) or outside the graph')
                    self._graph.instantiate()
                # the graph has the raw pointers of the library's scratch buffers baked in: keep those
                # buffers alive for the graph's lifetime, whatever later eager calls make of their tags
                self._scratch = _lib.workspace.hold(dev)
            finally:
                m.parallel_branches = False

    def _body(self, cameras):
        m = self.module
        depth, fdepth, sem, feat_cl = neck_ops.prefilter(self.depthnet_out, m.D, m.out_channels, m.depth_threshold,
                                                         m.semantic_threshold)
        bev, _, bev_mask, extras = m.view_transform(self.inputs, fdepth, None, feat_cl, cameras=cameras)
        return bev, depth, (bev_mask, sem), extras

    def replay(self, cam_idx_list=None):
        """Replay on the values already in the static buffers (``self.inputs``, ``self.depthnet_out``)."""
        if cam_idx_list is None:
            cam_idx_list = [random.randint(0, 5) for _ in range(self.batch)]
        if self.per_forward_geometry:
            self._cams['cam_sel'].copy_(self.module._small_h2d(torch.tensor(list(cam_idx_list), dtype=torch.int32),
                                                               self.device))
            self._cams['cam_idx_list'] = list(cam_idx_list)
        else:
            self.module.stage_cameras(self.module._geo, cam_idx_list, self.device, out=self._cams)
        self._graph.replay()
        bev, depth, masks, ex = self._static_out
        return bev, depth, masks, list(ex[:5]) + [list(cam_idx_list)] + list(ex[6:])

    def __call__(self, input, depthnet_out, cam_idx_list=None):
        # image features and the three raw-image variants; with the geometry inside the graph also the calibration
        for i in (0, 8, 9, 10) + ((1, 2, 3, 4, 5, 6, 11) if self.per_forward_geometry else ()):
            if torch.is_tensor(input[i]) and input[i] is not self.inputs[i]:
                self.inputs[i].copy_(input[i], non_blocking=True)
        if depthnet_out is not self.depthnet_out:
            self.depthnet_out.copy_(depthnet_out, non_blocking=True)
        return self.replay(cam_idx_list)
