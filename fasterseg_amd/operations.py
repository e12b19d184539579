"""Operator library of the FasterSeg search space on MI355X kernels.

Drop-in for the reference's search/operations.py (== train/operations.py): same public names
(`ConvNorm`, `BasicResidual1x`, `BasicResidual_downup_1x`, `BasicResidual2x`, `BasicResidual_downup_2x`,
`FactorizedReduce`, `OPS`, `OPS_name`, `OPS_Class`), constructor signatures, `set_ratio`, `forward`,
`forward_latency`, `_latency`, `_flops`, attribute names and therefore state_dict keys (`conv1/bn1/conv2/bn2`,
`conv.{0,1}`, `bn.bn.<w>`), and the same assertions.  What differs is underneath: every forward is a short chain of
fused HIP kernels (conv+BN-stats, BN-apply+ReLU, bilinear+ReLU) on NHWC tensors; see functional.py.

The four residual ops share one implementation parameterised by (number of convs, zoomed or not) — the reference
spells them out four times (operations.py:131-446).
"""
import atexit
import json
import os
import os.path as osp

import numpy as np
import torch
import torch.nn as nn

from . import functional as FN
from .latency import compute_latency_ms_hip as compute_latency
from .nn import BatchNorm2d, Conv2d, ReLU
from .slimmable_ops import USBatchNorm2d, USConv2d

__all__ = ['ConvNorm', 'BasicResidual1x', 'BasicResidual_downup_1x', 'BasicResidual2x', 'BasicResidual_downup_2x',
           'FactorizedReduce', 'OPS', 'OPS_name', 'OPS_Class']

# Per-operator latency table with the reference's key grammar and on-disk format (operations.py:33-36):
# a pickled dict str -> ms in ./latency_lookup_table.npy.  Entries produced here are hipEvent timings of these
# kernels on MI355X (fasterseg_amd.latency), not TensorRT/1080Ti numbers.
#
# Persistence: the reference rewrites the whole pickled dict on every miss (operations.py:116-122) - quadratic while a search fills the
# 667-key table.  Here a miss appends ONE line to `<table>.journal` (durable at once, O(1)); the `.npy` - the file the reference and
# later processes load - is rewritten once, by flush_latency_table() (registered with atexit, callable earlier), and a journal left
# behind by a killed process is merged at the next import.
latency_lookup_table = {}
table_file_name = "latency_lookup_table.npy"
_journal_dirty = False


def _journal_name():
    return table_file_name + ".journal"


def _load_latency_table():
    if osp.isfile(table_file_name):
        latency_lookup_table.update(np.load(table_file_name, allow_pickle=True).item())
    if osp.isfile(_journal_name()):
        global _journal_dirty
        with open(_journal_name()) as f:
            for line in f:
                line = line.strip()
                if line:
                    try:
                        key, ms = json.loads(line)
                    except ValueError:                      # torn last line of a killed writer
                        continue
                    latency_lookup_table[key] = ms
                    _journal_dirty = True


def flush_latency_table():
    """Writes the table in the reference's format (one np.save) and retires the journal; a no-op when nothing was measured."""
    global _journal_dirty
    if not _journal_dirty:
        return
    np.save(table_file_name, dict(latency_lookup_table))
    _journal_dirty = False
    if osp.isfile(_journal_name()):
        os.remove(_journal_name())


_load_latency_table()
atexit.register(flush_latency_table)


def lookup_latency(name, measure):
    """LUT hit, or measure + insert + journal the new entry (operations.py:116-122; see the persistence note above)."""
    global _journal_dirty
    if name in latency_lookup_table:
        return latency_lookup_table[name]
    print("not found in latency_lookup_table:", name)
    latency = measure()
    latency_lookup_table[name] = latency
    with open(_journal_name(), "a") as f:
        f.write(json.dumps([name, float(latency)]) + "\n")
    _journal_dirty = True
    return latency


def _bn_args(bn, bump=True):
    """(module holding the active statistics, use-batch-stats flag); bumps num_batches_tracked like nn.BatchNorm2d
    (bump=False: the caller hands the counter to a kernel that increments it on the device)."""
    b = bn.active() if isinstance(bn, USBatchNorm2d) else bn
    if bump and b.training and b.track_running_stats and b.num_batches_tracked is not None:
        b.num_batches_tracked.add_(1)
    return b, (b.training or not b.track_running_stats)


def conv_bn(x, conv, bn, relu):
    """conv -> BN -> [ReLU] as one fused unit; understands USConv2d/USBatchNorm2d width slices."""
    if isinstance(conv, USConv2d):
        cout, cin = conv.active_channels()
    else:
        cout, cin = conv.out_channels, conv.in_channels
    b, use_batch = _bn_args(bn, bump=False)
    nbt = b.num_batches_tracked if (b.training and b.track_running_stats) else None
    return FN.conv_bn_act(x, conv.weight, b.weight, b.bias, b.running_mean, b.running_var, conv.stride[0], conv.padding[0],
                          relu, use_batch, 0.1 if b.momentum is None else b.momentum, b.eps, cout, cin, nbt)


def _conv_macs(h, w, c_in, c_out, k):
    return h * w * c_out * c_in * k * k


class ConvNorm(nn.Module):
    '''
    conv => norm => activation (reference operations.py:42-128)
    '''
    def __init__(self, C_in, C_out, kernel_size=3, stride=1, padding=None, dilation=1, groups=1, bias=False, slimmable=True,
                 width_mult_list=[1.]):
        super(ConvNorm, self).__init__()
        self.C_in = C_in
        self.C_out = C_out
        self.kernel_size = kernel_size
        assert stride in [1, 2]
        self.stride = stride
        if padding is None:
            # assume h_out = h_in / s
            self.padding = int(np.ceil((dilation * (kernel_size - 1) + 1 - stride) / 2.))
        else:
            self.padding = padding
        self.dilation = dilation
        assert type(groups) == int
        self.groups = 1 if kernel_size == 1 else groups
        self.bias = bias
        self.slimmable = slimmable
        self.width_mult_list = width_mult_list
        self.ratio = (1., 1.)
        if bias:
            raise NotImplementedError("fasterseg_amd: ConvNorm(bias=True) is never used by the reference networks")
        if slimmable:
            conv = USConv2d(C_in, C_out, kernel_size, stride, padding=self.padding, dilation=dilation, groups=self.groups,
                            bias=bias, width_mult_list=width_mult_list)
            norm = USBatchNorm2d(C_out, width_mult_list)
        else:
            conv = Conv2d(C_in, C_out, kernel_size, stride, padding=self.padding, dilation=dilation, groups=self.groups,
                          bias=bias)
            norm = BatchNorm2d(C_out)
        self.conv = nn.Sequential(conv, norm, ReLU(inplace=True))

    def set_ratio(self, ratio):
        assert self.slimmable
        assert len(ratio) == 2
        self.__dict__['ratio'] = ratio      # plain attribute: skip nn.Module.__setattr__'s type dispatch
        self.conv[0].set_ratio(ratio)
        self.conv[1].set_ratio(ratio[1])

    @staticmethod
    def _flops(h, w, C_in, C_out, kernel_size=3, stride=1, padding=None, dilation=1, groups=1, bias=False):
        ho, wo = (h, w) if stride == 1 else (h // 2, w // 2)
        return _conv_macs(ho, wo, C_in, C_out, kernel_size) + 2 * ho * wo * C_out

    @staticmethod
    def _latency(h, w, C_in, C_out, kernel_size=3, stride=1, padding=None, dilation=1, groups=1, bias=False):
        layer = ConvNorm(C_in, C_out, kernel_size, stride, padding, dilation, groups, bias, slimmable=False)
        return compute_latency(layer, (1, C_in, h, w))

    def forward_latency(self, size):
        c_in, h_in, w_in = size
        if self.slimmable:
            assert c_in == int(self.C_in * self.ratio[0]), "c_in %d, self.C_in * self.ratio[0] %d" % (c_in, self.C_in * self.ratio[0])
            c_out = int(self.C_out * self.ratio[1])
        else:
            assert c_in == self.C_in, "c_in %d, self.C_in %d" % (c_in, self.C_in)
            c_out = self.C_out
        h_out, w_out = (h_in, w_in) if self.stride == 1 else (h_in // 2, w_in // 2)
        name = "ConvNorm_H%d_W%d_Cin%d_Cout%d_kernel%d_stride%d" % (h_in, w_in, c_in, c_out, self.kernel_size, self.stride)
        latency = lookup_latency(name, lambda: ConvNorm._latency(h_in, w_in, c_in, c_out, self.kernel_size, self.stride,
                                                                 self.padding, self.dilation, self.groups, self.bias))
        return latency, (c_out, h_out, w_out)

    def _is_stem(self, x):
        return (not self.slimmable and self.C_in == 3 and self.kernel_size == 3 and self.stride == 2 and self.padding == 1)

    def forward(self, x):
        assert x.size()[1] == self.C_in, "{} {}".format(x.size()[1], self.C_in)
        if self._is_stem(x):
            b, use_batch = _bn_args(self.conv[1])
            return FN.stem_conv_bn_act(x, self.conv[0].weight, b.weight, b.bias, b.running_mean, b.running_var, True,
                                       use_batch, 0.1 if b.momentum is None else b.momentum, b.eps)
        return conv_bn(FN.as_nhwc(x), self.conv[0], self.conv[1], relu=True)


class _Residual(nn.Module):
    """Shared body of the four conv primitives.

    NUM_CONVS = 1 | 2   ('conv' vs 'conv_2x');  ZOOM = True for the "zoomed conv" (bilinear /2 -> conv(s) at stride 1
    -> bilinear x2 when the op's stride is 1; with stride 2 the down-sample *is* the stride).  ReLU is always the last
    step, i.e. after the up-sample for zoomed ops (operations.py:270-277,436-446)."""
    NUM_CONVS = 1
    ZOOM = False
    LUT_NAME = None          # prefix of the latency key
    LUT_CLASS = None         # class whose _latency measures a miss (see BasicResidual_downup_2x)

    def __init__(self, C_in, C_out, kernel_size=3, stride=1, dilation=1, groups=1, slimmable=True, width_mult_list=[1.]):
        super(_Residual, self).__init__()
        self.C_in = C_in
        self.C_out = C_out
        self.kernel_size = kernel_size
        self.stride = stride
        self.dilation = dilation
        self.groups = groups
        self.slimmable = slimmable
        self.width_mult_list = width_mult_list
        assert stride in [1, 2]
        if self.stride == 2:
            self.dilation = 1
        self.ratio = (1., 1.)
        self.relu = ReLU(inplace=True)
        first_stride = 1 if self.ZOOM else stride
        chans = [(C_in, C_out, first_stride), (C_out, C_out, 1)][:self.NUM_CONVS]
        for i, (ci, co, s) in enumerate(chans, start=1):
            if slimmable:
                conv = USConv2d(ci, co, 3, s, padding=dilation, dilation=dilation, groups=groups, bias=False,
                                width_mult_list=width_mult_list)
                norm = USBatchNorm2d(co, width_mult_list)
            else:
                conv = Conv2d(ci, co, 3, s, padding=dilation, dilation=dilation, groups=groups, bias=False)
                norm = BatchNorm2d(co)
            setattr(self, "conv%d" % i, conv)
            setattr(self, "bn%d" % i, norm)

    def set_ratio(self, ratio):
        assert len(ratio) == 2
        self.__dict__['ratio'] = ratio      # plain attribute: skip nn.Module.__setattr__'s type dispatch
        self.conv1.set_ratio(ratio)
        self.bn1.set_ratio(ratio[1])
        if self.NUM_CONVS == 2:
            self.conv2.set_ratio((ratio[1], ratio[1]))
            self.bn2.set_ratio(ratio[1])

    @classmethod
    def _flops(cls, h, w, C_in, C_out, kernel_size=3, stride=1, dilation=1, groups=1):
        assert stride in [1, 2]
        hc, wc = (h // 2, w // 2) if (cls.ZOOM or stride == 2) else (h, w)
        macs = _conv_macs(hc, wc, C_in, C_out, 3) + 2 * hc * wc * C_out
        if cls.NUM_CONVS == 2:
            macs += _conv_macs(hc, wc, C_out, C_out, 3) + 2 * hc * wc * C_out
        return macs

    @classmethod
    def _latency(cls, h, w, C_in, C_out, kernel_size=3, stride=1, dilation=1, groups=1):
        assert stride in [1, 2]
        layer = cls(C_in, C_out, kernel_size, stride, dilation, groups, slimmable=False)
        return compute_latency(layer, (1, C_in, h, w))

    def forward_latency(self, size):
        c_in, h_in, w_in = size
        if self.slimmable:
            assert c_in == int(self.C_in * self.ratio[0]), "c_in %d, int(self.C_in * self.ratio[0]) %d" % (c_in, int(self.C_in * self.ratio[0]))
            c_out = int(self.C_out * self.ratio[1])
        else:
            assert c_in == self.C_in, "c_in %d, self.C_in %d" % (c_in, self.C_in)
            c_out = self.C_out
        h_out, w_out = (h_in, w_in) if self.stride == 1 else (h_in // 2, w_in // 2)
        name = "%s_H%d_W%d_Cin%d_Cout%d_stride%d_dilation%d" % (self.LUT_NAME, h_in, w_in, c_in, c_out, self.stride, self.dilation)
        lut_cls = self.LUT_CLASS or type(self)
        latency = lookup_latency(name, lambda: lut_cls._latency(h_in, w_in, c_in, c_out, self.kernel_size, self.stride,
                                                                self.dilation, self.groups))
        return latency, (c_out, h_out, w_out)

    def forward(self, x):
        x = FN.as_nhwc(x)
        H, W = int(x.size(2)), int(x.size(3))
        upsample = self.ZOOM and self.stride == 1
        out = FN.interpolate(x, size=(H // 2, W // 2)) if self.ZOOM else x
        out = conv_bn(out, self.conv1, self.bn1, relu=(self.NUM_CONVS == 2 or not upsample))
        if self.NUM_CONVS == 2:
            out = conv_bn(out, self.conv2, self.bn2, relu=not upsample)
        if upsample:
            out = FN.interpolate(out, size=(H, W), relu=True)
        return out


class BasicResidual1x(_Residual):
    """'conv': 3x3 conv(stride) -> BN -> ReLU (operations.py:131-200)."""
    NUM_CONVS, ZOOM, LUT_NAME = 1, False, "BasicResidual1x"


class BasicResidual_downup_1x(_Residual):
    """'conv_downup' zoomed conv (operations.py:203-277)."""
    NUM_CONVS, ZOOM, LUT_NAME = 1, True, "BasicResidual_downup_1x"


class BasicResidual2x(_Residual):
    """'conv_2x': two 3x3 convs (operations.py:280-359)."""
    NUM_CONVS, ZOOM, LUT_NAME = 2, False, "BasicResidual2x"


class BasicResidual_downup_2x(_Residual):
    """'conv_2x_downup' (operations.py:362-446).  Reference quirk kept on purpose: its forward_latency prices the op
    with the *BasicResidual2x_* key and BasicResidual2x._latency (operations.py:426-431), so the search sees zoomed
    2x cells at plain-2x cost.  Set FIX_LUT_KEY = True to use the op's own key instead."""
    NUM_CONVS, ZOOM = 2, True
    FIX_LUT_KEY = False

    @property
    def LUT_NAME(self):
        return "BasicResidual_downup_2x" if self.FIX_LUT_KEY else "BasicResidual2x"

    @property
    def LUT_CLASS(self):
        return BasicResidual_downup_2x if self.FIX_LUT_KEY else BasicResidual2x


class FactorizedReduce(nn.Module):
    """'skip' (operations.py:449-534): stride 2 -> two 1x1 stride-2 convs on x and x[:,:,1:,1:], channel concat, BN,
    ReLU (one fused unit here); stride 1 -> identity, or a slimmable 1x1 conv+BN+ReLU when widths may differ."""

    def __init__(self, C_in, C_out, stride=1, slimmable=True, width_mult_list=[1.]):
        super(FactorizedReduce, self).__init__()
        assert stride in [1, 2]
        assert C_out % 2 == 0
        self.C_in = C_in
        self.C_out = C_out
        self.stride = stride
        self.slimmable = slimmable
        self.width_mult_list = width_mult_list
        self.ratio = (1., 1.)
        if stride == 1 and slimmable:
            self.conv1 = USConv2d(C_in, C_out, 1, stride=1, padding=0, bias=False, width_mult_list=width_mult_list)
            self.bn = USBatchNorm2d(C_out, width_mult_list)
            self.relu = ReLU(inplace=True)
        elif stride == 2:
            self.relu = ReLU(inplace=True)
            if slimmable:
                self.conv1 = USConv2d(C_in, C_out // 2, 1, stride=2, padding=0, bias=False, width_mult_list=width_mult_list)
                self.conv2 = USConv2d(C_in, C_out // 2, 1, stride=2, padding=0, bias=False, width_mult_list=width_mult_list)
                self.bn = USBatchNorm2d(C_out, width_mult_list)
            else:
                self.conv1 = Conv2d(C_in, C_out // 2, 1, stride=2, padding=0, bias=False)
                self.conv2 = Conv2d(C_in, C_out // 2, 1, stride=2, padding=0, bias=False)
                self.bn = BatchNorm2d(C_out)

    def set_ratio(self, ratio):
        assert len(ratio) == 2
        self.__dict__['ratio'] = ratio      # plain attribute: skip nn.Module.__setattr__'s type dispatch
        if self.stride == 1:
            self.conv1.set_ratio(ratio)
            self.bn.set_ratio(ratio[1])
        elif self.stride == 2:
            self.conv1.set_ratio(ratio)
            self.conv2.set_ratio(ratio)
            self.bn.set_ratio(ratio[1])

    @staticmethod
    def _flops(h, w, C_in, C_out, stride=1):
        if stride == 1:
            return 0
        return _conv_macs(h // 2, w // 2, C_in, C_out, 1) + 2 * (h // 2) * (w // 2) * C_out

    @staticmethod
    def _latency(h, w, C_in, C_out, stride=1):
        layer = FactorizedReduce(C_in, C_out, stride, slimmable=False)
        return compute_latency(layer, (1, C_in, h, w))

    def forward_latency(self, size):
        c_in, h_in, w_in = size
        if self.slimmable:
            assert c_in == int(self.C_in * self.ratio[0])
            c_out = int(self.C_out * self.ratio[1])
        else:
            assert c_in == self.C_in
            c_out = self.C_out
        h_out, w_out = (h_in, w_in) if self.stride == 1 else (h_in // 2, w_in // 2)
        name = "FactorizedReduce_H%d_W%d_Cin%d_Cout%d_stride%d" % (h_in, w_in, c_in, c_out, self.stride)
        latency = lookup_latency(name, lambda: FactorizedReduce._latency(h_in, w_in, c_in, c_out, self.stride))
        return latency, (c_out, h_out, w_out)

    def forward(self, x):
        if self.stride == 2:
            x = FN.as_nhwc(x)
            if self.slimmable:
                half, cin = self.conv1.active_channels()
                half2, cin2 = self.conv2.active_channels()
                assert (half, cin) == (half2, cin2)
            else:
                half, cin = self.conv1.out_channels, self.conv1.in_channels
            b, use_batch = _bn_args(self.bn, bump=False)
            nbt = b.num_batches_tracked if (b.training and b.track_running_stats) else None
            assert b.num_features == 2 * half, "running_mean should contain %d elements not %d" % (2 * half, b.num_features)
            return FN.factorized_reduce(x, self.conv1.weight, self.conv2.weight, b.weight, b.bias, b.running_mean,
                                        b.running_var, use_batch, 0.1 if b.momentum is None else b.momentum, b.eps, half, cin, nbt)
        if self.slimmable:
            return conv_bn(FN.as_nhwc(x), self.conv1, self.bn, relu=True)
        return x


from collections import OrderedDict
OPS = {
    'skip': lambda C_in, C_out, stride, slimmable, width_mult_list: FactorizedReduce(C_in, C_out, stride, slimmable, width_mult_list),
    'conv': lambda C_in, C_out, stride, slimmable, width_mult_list: BasicResidual1x(C_in, C_out, kernel_size=3, stride=stride, dilation=1, slimmable=slimmable, width_mult_list=width_mult_list),
    'conv_downup': lambda C_in, C_out, stride, slimmable, width_mult_list: BasicResidual_downup_1x(C_in, C_out, kernel_size=3, stride=stride, dilation=1, slimmable=slimmable, width_mult_list=width_mult_list),
    'conv_2x': lambda C_in, C_out, stride, slimmable, width_mult_list: BasicResidual2x(C_in, C_out, kernel_size=3, stride=stride, dilation=1, slimmable=slimmable, width_mult_list=width_mult_list),
    'conv_2x_downup': lambda C_in, C_out, stride, slimmable, width_mult_list: BasicResidual_downup_2x(C_in, C_out, kernel_size=3, stride=stride, dilation=1, slimmable=slimmable, width_mult_list=width_mult_list),
}
OPS_name = ["FactorizedReduce", "BasicResidual1x", "BasicResidual_downup_1x", "BasicResidual2x", "BasicResidual_downup_2x"]
OPS_Class = OrderedDict()
OPS_Class['skip'] = FactorizedReduce
OPS_Class['conv'] = BasicResidual1x
OPS_Class['conv_downup'] = BasicResidual_downup_1x
OPS_Class['conv_2x'] = BasicResidual2x
OPS_Class['conv_2x_downup'] = BasicResidual_downup_2x
