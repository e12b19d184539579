"""Segmentation head blocks (reference search/seg_oprs.py: ConvBnRelu :17-39, FeatureFusion :181-225, Head :228-274).

Only the three classes the networks instantiate are provided; the BiSeNet/DFN leftovers in the reference file
(SeparableConvBnRelu, SELayer, ChannelAttention, BNRefine, RefineResidual, AttentionRefinement) are dead code there.
FeatureFusion keeps its never-executed `channel_attention` branch because its two 1x1 weights are part of the
state_dict (`ffm.channel_attention.{1,2}.conv.weight`)."""
import os.path as osp

import numpy as np
import torch.nn as nn

from . import functional as FN
from . import operations as _ops
from .latency import compute_latency_ms_hip as compute_latency
from .nn import BatchNorm2d, Conv2d, ReLU

latency_lookup_table = _ops.latency_lookup_table      # one table, shared with operations (same file on disk)


class ConvBnRelu(nn.Module):
    def __init__(self, in_planes, out_planes, ksize, stride, pad, dilation=1, groups=1, has_bn=True, norm_layer=BatchNorm2d,
                 bn_eps=1e-5, has_relu=True, inplace=True, has_bias=False):
        super(ConvBnRelu, self).__init__()
        self.conv = Conv2d(in_planes, out_planes, kernel_size=ksize, stride=stride, padding=pad, dilation=dilation,
                           groups=groups, bias=has_bias)
        self.has_bn = has_bn
        if self.has_bn:
            self.bn = norm_layer(out_planes, eps=bn_eps)
        self.has_relu = has_relu
        if self.has_relu:
            self.relu = ReLU(inplace=inplace)

    def forward(self, x):
        x = FN.as_nhwc(x)
        if self.has_bn and self.conv.bias is None:
            return _ops.conv_bn(x, self.conv, self.bn, relu=self.has_relu)
        x = self.conv(x)
        if self.has_bn:
            x = self.bn(x)
        if self.has_relu:
            x = self.relu(x)
        return x


class FeatureFusion(nn.Module):
    def __init__(self, in_planes, out_planes, reduction=1, Fch=16, scale=4, branch=2, norm_layer=BatchNorm2d):
        super(FeatureFusion, self).__init__()
        self.conv_1x1 = ConvBnRelu(in_planes, out_planes, 1, 1, 0, has_bn=True, norm_layer=norm_layer, has_relu=True,
                                   has_bias=False)
        self.channel_attention = nn.Sequential(      # disabled in the reference forward (seg_oprs.py:223-225)
            nn.AdaptiveAvgPool2d(1),
            ConvBnRelu(out_planes, out_planes // reduction, 1, 1, 0, has_bn=False, norm_layer=norm_layer, has_relu=True,
                       has_bias=False),
            ConvBnRelu(out_planes // reduction, out_planes, 1, 1, 0, has_bn=False, norm_layer=norm_layer, has_relu=False,
                       has_bias=False),
            nn.Sigmoid()
        )
        self._Fch = Fch
        self._scale = scale
        self._branch = branch

    @staticmethod
    def _latency(h, w, C_in, C_out):
        layer = FeatureFusion(C_in, C_out)
        return compute_latency(layer, (1, C_in, h, w))

    def forward_latency(self, size):
        name = "ff_H%d_W%d_C%d" % (size[1], size[2], size[0])
        c = self._scale * self._Fch * self._branch
        latency = _ops.lookup_latency(name, lambda: FeatureFusion._latency(size[1], size[2], c, c))
        return latency, size

    def forward(self, fm):
        # fm is already a concatenation of multiple scales
        return self.conv_1x1(fm)


class Head(nn.Module):
    def __init__(self, in_planes, out_planes=19, Fch=16, scale=4, branch=2, is_aux=False, norm_layer=BatchNorm2d):
        super(Head, self).__init__()
        mid_planes = in_planes if in_planes <= 256 else in_planes // 2     # seg_oprs.py:231-243
        self.conv_3x3 = ConvBnRelu(in_planes, mid_planes, 3, 1, 1, has_bn=True, norm_layer=norm_layer, has_relu=True,
                                   has_bias=False)
        self.conv_1x1 = Conv2d(mid_planes, out_planes, kernel_size=1, stride=1, padding=0)
        self._in_planes = in_planes
        self._out_planes = out_planes
        self._Fch = Fch
        self._scale = scale
        self._branch = branch

    @staticmethod
    def _latency(h, w, C_in, C_out=19):
        layer = Head(C_in, C_out)
        return compute_latency(layer, (1, C_in, h, w))

    def forward_latency(self, size):
        assert size[0] == self._in_planes, "size[0] %d, self._in_planes %d" % (size[0], self._in_planes)
        name = "head_H%d_W%d_Cin%d_Cout%d" % (size[1], size[2], size[0], self._out_planes)
        latency = _ops.lookup_latency(name, lambda: Head._latency(size[1], size[2], self._scale * self._Fch * self._branch,
                                                                  self._out_planes))
        return latency, (self._out_planes, size[1], size[2])

    def forward(self, x):
        fm = self.conv_3x3(x)
        return self.conv_1x1(fm)      # (N,19,h,w) view of a zero-padded 32-channel NHWC buffer
