"""Parameter-holding layers.  They subclass torch.nn.Conv2d / torch.nn.BatchNorm2d so that state_dict keys, OIHW
shapes and `isinstance` checks of the reference utilities (tools/utils/init_func.py:7-14 `init_weight`,
`group_weight`) keep working, but their forward() runs the HIP kernels; ATen/MIOpen is never called.  The fused
operators in fasterseg_amd.operations bypass these forwards and read the parameters directly."""
import torch
import torch.nn as tnn

from . import functional as FN


class Conv2d(tnn.Conv2d):
    """nn.Conv2d (1x1 / 3x3, stride 1|2, dilation 1, groups 1) on fs_conv2d_fwd (+ bias in the epilogue)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                         groups=groups, bias=bias)
        _check_conv(self)

    def forward(self, x):
        return FN.conv_bias(x, self.weight, self.bias, self.stride[0], self.padding[0])


def _check_conv(m):
    k = m.kernel_size
    if k[0] != k[1] or k[0] not in (1, 3):
        raise NotImplementedError("fasterseg_amd: only 1x1 and 3x3 filters are on the hot path (got %s)" % (k,))
    if m.dilation != (1, 1) or m.groups != 1:
        raise NotImplementedError("fasterseg_amd: dilation/groups other than 1 are never used by OPS "
                                  "(search/operations.py:539-545) and are not implemented")
    if m.stride[0] != m.stride[1] or m.stride[0] not in (1, 2):
        raise NotImplementedError("fasterseg_amd: stride must be 1 or 2")


class BatchNorm2d(tnn.BatchNorm2d):
    """nn.BatchNorm2d: train mode = fs_channel_stats + fs_bn_finalize + fs_affine_act, eval = folded fs_affine_act."""

    def forward(self, x):
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        use_batch = self.training or not self.track_running_stats
        return FN.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, use_batch,
                             self.momentum if self.momentum is not None else 0.1, self.eps)


class ReLU(tnn.Module):
    """Occupies the reference's nn.ReLU(inplace=True) slots.  On the hot path the activation is fused into the
    producing kernel's epilogue and this module is not called; standalone it runs fs_affine_act(scale=1, shift=0)."""

    def __init__(self, inplace=True):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        from . import kernels as K
        x = FN.as_nhwc(x)
        one = torch.ones(x.shape[1], dtype=torch.float32, device=x.device)
        return K.affine_act(x, one, torch.zeros_like(one), True)
