"""Leaf modules: torch.nn parameter containers whose forward runs HIP kernels.

Each class subclasses the torch.nn module the reference instantiates, so
constructor signatures, default initialisation (same RNG consumption order),
parameter / buffer names and ``state_dict`` layout are identical to the
reference - but ``forward`` never reaches ATen: it calls the nasseg kernels
through ``functional``.  ``FusedSequential`` is an ``nn.Sequential`` whose
forward peephole-fuses BatchNorm+activation(+residual) and ReLU+depthwise.
"""
import torch.nn as nn

from .. import functional as F
from .._lib import NassegError


class Conv2d(nn.Conv2d):
    """nn.Conv2d: dense (groups=1) on the fp32 MFMA path, or depthwise (groups=C)."""

    def _check(self):
        k, s, p, d = self.kernel_size, self.stride, self.padding, self.dilation
        if isinstance(p, str) or k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or d[0] != d[1]:
            raise NassegError("only square kernel/stride/padding/dilation are supported")
        if self.padding_mode != "zeros":
            raise NassegError("only zero padding is supported")

    @property
    def is_depthwise(self):
        return self.groups > 1 and self.groups == self.in_channels == self.out_channels

    def forward(self, x, relu_in=False):
        self._check()
        s, p, d = self.stride[0], self.padding[0], self.dilation[0]
        if self.is_depthwise:
            if self.bias is not None:
                raise NassegError("depthwise conv with bias is not on the reference path")
            return F.depthwise_conv2d(x, self.weight, s, p, d, relu_in=relu_in)
        if self.groups != 1:
            raise NassegError("grouped conv (groups={}) is not on the reference path".format(self.groups))
        if relu_in:
            x = F.relu(x)
        return F.conv2d(x, self.weight, self.bias, s, p, d)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d with an optional fused activation / residual add."""

    def forward(self, x, act=F.ACT_NONE, residual=None):
        if self.momentum is None:
            raise NassegError("cumulative-average BatchNorm (momentum=None) is not supported")
        use_batch_stats = self.training or not self.track_running_stats
        return F.batch_norm_act(
            x, self.weight, self.bias,
            self.running_mean if self.track_running_stats else None,
            self.running_var if self.track_running_stats else None,
            self.num_batches_tracked if (self.training and self.track_running_stats) else None,
            use_batch_stats, self.momentum, self.eps, act, residual)


class ReLU(nn.ReLU):
    def forward(self, x):
        return F.relu(x)


class ReLU6(nn.ReLU6):
    def forward(self, x):
        # standalone ReLU6 only appears fused behind a BatchNorm on the reference path
        raise NassegError("stand-alone ReLU6 is not on the reference path")


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        return F.max_pool2d(x, self.kernel_size, self.stride, self.padding)


class AvgPool2d(nn.AvgPool2d):
    def forward(self, x):
        if self.count_include_pad:
            raise NassegError("count_include_pad=True is not on the reference path")
        return F.avg_pool2d(x, self.kernel_size, self.stride, self.padding)


class FusedSequential(nn.Sequential):
    """nn.Sequential with BN+act(+residual) and ReLU+depthwise peephole fusion."""

    def forward(self, x, residual=None):
        mods = list(self._modules.values())
        n = len(mods)
        i = 0
        res_used = residual is None
        while i < n:
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < n else None
            if (isinstance(m, Conv2d) and m.groups == 1 and m.bias is None and isinstance(nxt, BatchNorm2d)
                    and nxt.momentum is not None and nxt.track_running_stats
                    and nxt.weight is not None and m.out_channels % 4 == 0):
                # dense conv -> BatchNorm [-> ReLU/ReLU6] [+ residual]: one fused node
                m._check()
                bn = nxt
                after = mods[i + 2] if i + 2 < n else None
                act, step = F.ACT_NONE, 2
                if isinstance(after, nn.ReLU6):
                    act, step = F.ACT_RELU6, 3
                elif isinstance(after, nn.ReLU):
                    act, step = F.ACT_RELU, 3
                res = None
                if not res_used and i + step == n:
                    res, res_used = residual, True
                x = F.conv_bn_act(x, m.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  bn.num_batches_tracked if bn.training else None, bn.training,
                                  bn.momentum, bn.eps, act, res, m.stride[0], m.padding[0],
                                  m.dilation[0])
                i += step
            elif isinstance(m, BatchNorm2d):
                act, step = F.ACT_NONE, 1
                if isinstance(nxt, nn.ReLU6):
                    act, step = F.ACT_RELU6, 2
                elif isinstance(nxt, nn.ReLU):
                    act, step = F.ACT_RELU, 2
                res = None
                if not res_used and i + step == n:
                    res, res_used = residual, True
                x = m(x, act=act, residual=res)
                i += step
            elif (isinstance(m, nn.ReLU) and not isinstance(m, nn.ReLU6)
                  and isinstance(nxt, Conv2d) and nxt.is_depthwise):
                x = nxt(x, relu_in=True)
                i += 2
            else:
                x = m(x)
                i += 1
        if not res_used:
            x = F.add(x, residual)
        return x
