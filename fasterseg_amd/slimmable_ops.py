"""Width-slimmable conv and BN bank (reference search/slimmable_ops.py).

USConv2d keeps the full OIHW weight (state_dict compatible); a forward at ratio (r_in, r_out) uses the leading
[:out', :in'] block (slimmable_ops.py:38-42).  The reference hands cuDNN a non-contiguous weight view; here the
slice is packed straight from the strided parameter by fs_pack_weight and the weight gradient is scattered back into
the full tensor by fs_unpack_weight_grad, so no slice copy is made on either pass."""
import torch.nn as tnn

from . import functional as FN
from .nn import BatchNorm2d, Conv2d, _check_conv


def make_divisible(v, divisor=8, min_value=1):
    """Channel rounding rule of slim/MobileNet (slimmable_ops.py:5-18)."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:       # never round down by more than 10 %
        new_v += divisor
    return new_v


class USConv2d(tnn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, depthwise=False,
                 bias=True, width_mult_list=[1.]):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                         groups=groups, bias=bias)
        if depthwise:
            raise NotImplementedError("fasterseg_amd: depthwise USConv2d is never instantiated by the reference OPS")
        _check_conv(self)
        self.depthwise = depthwise
        self.in_channels_max = in_channels
        self.out_channels_max = out_channels
        self.width_mult_list = width_mult_list
        self.ratio = (1., 1.)

    def set_ratio(self, ratio):
        self.__dict__['ratio'] = ratio      # plain attribute: skip nn.Module.__setattr__'s type dispatch

    def active_channels(self):
        """(out', in') for the current ratio; same membership asserts as slimmable_ops.py:37-40."""
        assert self.ratio[0] in self.width_mult_list, str(self.ratio[0]) + " in? " + str(self.width_mult_list)
        assert self.ratio[1] in self.width_mult_list, str(self.ratio[1]) + " in? " + str(self.width_mult_list)
        d = self.__dict__
        d['in_channels'] = make_divisible(self.in_channels_max * self.ratio[0])
        d['out_channels'] = make_divisible(self.out_channels_max * self.ratio[1])
        return self.out_channels, self.in_channels

    def forward(self, x):
        cout, cin = self.active_channels()
        if self.bias is not None:
            raise NotImplementedError("fasterseg_amd: biased USConv2d is not used by the reference")
        return FN.conv_slice(x, self.weight, cout, cin, self.stride[0], self.padding[0])


class USBatchNorm2d(tnn.BatchNorm2d):
    """One BatchNorm2d per width (slimmable_ops.py:51-70).  Like the reference, the module's own affine parameters
    exist (affine=True, track_running_stats=False) but are never used."""

    def __init__(self, num_features, width_mult_list=[1.]):
        super().__init__(num_features, affine=True, track_running_stats=False)
        self.num_features_max = num_features
        self.width_mult_list = width_mult_list
        self.bn = tnn.ModuleList([BatchNorm2d(make_divisible(self.num_features_max * w), affine=True)
                                  for w in width_mult_list])
        self.ratio = 1.

    def set_ratio(self, ratio):
        self.__dict__['ratio'] = ratio      # plain attribute: skip nn.Module.__setattr__'s type dispatch

    def active(self):
        assert self.ratio in self.width_mult_list
        return self.bn[self.width_mult_list.index(self.ratio)]

    def forward(self, x):
        return self.active()(x)
