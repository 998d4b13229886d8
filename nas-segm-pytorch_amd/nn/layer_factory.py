"""Op registry of the search space, backed by the nasseg HIP kernels.

Plugin boundary of the reference (src/nn/layer_factory.py:27-91):

    OPS[name](C_in, C_out, stride, affine, repeats=1)               -> nn.Module
    AGG_OPS[name](C_in0, C_in1, C_out, affine, repeats=1, larger=True) -> nn.Module

with ``forward(x)`` / ``forward(x, y)`` over NCHW-shaped fp32 tensors.  Keys,
signatures, module tree (hence ``state_dict`` names and parameter shapes),
default initialisation order and the documented quirks of the reference
(Skip ignores ``stride``; a strided SepConv strides in *every* repeat; Pool is
1x1 conv + BN *then* pooling; resize() compares sizes lexicographically) are
kept.  Every leaf runs on gfx950 through ``functional``; nothing falls back to
ATen.
"""
import os

import torch
import torch.nn as nn

from .. import functional as F
from .modules import AvgPool2d, BatchNorm2d, Conv2d, FusedSequential, MaxPool2d, ReLU, ReLU6


# ---------------------------------------------------------------------------
# small builders (reference: layer_factory.py:7-24,94-122)
# ---------------------------------------------------------------------------
def conv3x3(in_planes, out_planes, stride=1, bias=False, dilation=1):
    """3x3 convolution, 'same' padding for the given dilation."""
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation,
                  dilation=dilation, bias=bias)


def conv1x1(in_planes, out_planes, stride=1, bias=False):
    return Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=0, bias=bias)


def conv_bn(C_in, C_out, kernel_size, stride, padding, affine=True):
    return FusedSequential(
        Conv2d(C_in, C_out, kernel_size, stride=stride, padding=padding, bias=False),
        BatchNorm2d(C_out, affine=affine),
    )


def conv_bn_relu(C_in, C_out, kernel_size, stride, padding, affine=True):
    return FusedSequential(
        Conv2d(C_in, C_out, kernel_size, stride=stride, padding=padding, bias=False),
        BatchNorm2d(C_out, affine=affine),
        ReLU(inplace=False),
    )


def conv_bn_relu6(inp, oup, stride):
    return FusedSequential(Conv2d(inp, oup, 3, stride, 1, bias=False), BatchNorm2d(oup),
                           ReLU6(inplace=True))


def conv_1x1_bn_relu6(inp, oup):
    return FusedSequential(Conv2d(inp, oup, 1, 1, 0, bias=False), BatchNorm2d(oup),
                           ReLU6(inplace=True))


def _dense_bn_relu(C_in, C_out, ksize, stride, affine, dilation=1):
    conv = conv1x1(C_in, C_out, stride=stride) if ksize == 1 else conv3x3(
        C_in, C_out, stride=stride, dilation=dilation)
    return FusedSequential(conv, BatchNorm2d(C_out, affine=affine), ReLU(inplace=False))


# ---------------------------------------------------------------------------
# encoder block (reference: layer_factory.py:125-158)
# ---------------------------------------------------------------------------
class InvertedResidual(nn.Module):
    """MobileNetV2 block: 1x1 expand (also when t == 1) -> 3x3 depthwise -> 1x1 linear."""

    def __init__(self, inp, oup, stride, expand_ratio):
        super(InvertedResidual, self).__init__()
        assert stride in [1, 2]
        self.stride = stride
        self.use_res_connect = stride == 1 and inp == oup
        hidden = inp * expand_ratio
        self.conv = FusedSequential(
            Conv2d(inp, hidden, 1, 1, 0, bias=False),
            BatchNorm2d(hidden),
            ReLU6(inplace=True),
            Conv2d(hidden, hidden, 3, stride, 1, groups=hidden, bias=False),
            BatchNorm2d(hidden),
            ReLU6(inplace=True),
            Conv2d(hidden, oup, 1, 1, 0, bias=False),
            BatchNorm2d(oup),
        )

    def forward(self, x):
        # the skip connection is fused into the last BatchNorm's epilogue
        return self.conv(x, residual=x if self.use_res_connect else None)


# ---------------------------------------------------------------------------
# cell primitives
# ---------------------------------------------------------------------------
class Pool(nn.Module):
    """1x1 conv + BN (no activation) followed by 3x3 pooling (layer_factory.py:161-178)."""

    def __init__(self, C_in, C_out, stride, repeats, ksize, mode):
        super(Pool, self).__init__()
        self.conv1x1 = conv_bn(C_in, C_out, 1, 1, 0)
        if mode == "avg":
            self.pool = AvgPool2d(ksize, stride=stride, padding=(ksize // 2),
                                  count_include_pad=False)
        elif mode == "max":
            self.pool = MaxPool2d(ksize, stride=stride, padding=(ksize // 2))
        else:
            raise ValueError("Unknown pooling method {}".format(mode))

    def forward(self, x):
        if isinstance(self.pool, MaxPool2d) and self.pool.kernel_size == 3 and self.pool.padding == 1 and \
                self.pool.stride in (1, 2):
            # conv + BatchNorm -> max pooling as ONE node: the pooling applies the BatchNorm on load
            return self.conv1x1(x, pool=(3, self.pool.stride, 1))
        return self.pool(self.conv1x1(x))


class GAPConv1x1(nn.Module):
    """Global average pool -> 1x1 conv + BN + ReLU on (B,C,1,1) -> broadcast back
    (layer_factory.py:181-195; bilinear interpolation from 1x1 is a broadcast)."""

    def __init__(self, C_in, C_out):
        super(GAPConv1x1, self).__init__()
        self.conv1x1 = conv_bn_relu(C_in, C_out, 1, stride=1, padding=0)

    def forward(self, x):
        size = x.size()[2:]
        pooled = F.global_avg_pool(x)  # fp32 also under bf16 activation storage, and so is the
        # 1x1 conv + BatchNorm over B values per channel that follows; back to x's storage below
        return F.broadcast_to(self.conv1x1(pooled), size, x.dtype)


class DilConv(nn.Module):
    """ReLU -> dilated depthwise -> 1x1 -> BN (layer_factory.py:198-222)."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, dilation, affine=True):
        super(DilConv, self).__init__()
        self.op = FusedSequential(
            ReLU(inplace=False),
            Conv2d(C_in, C_in, kernel_size=kernel_size, stride=stride, padding=padding,
                   dilation=dilation, groups=C_in, bias=False),
            Conv2d(C_in, C_out, kernel_size=1, padding=0, bias=False),
            BatchNorm2d(C_out, affine=affine),
        )

    def forward(self, x, defer_tail=False):
        return self.op(x, defer_tail=defer_tail)


class SepConv(nn.Module):
    """``repeats`` x [depthwise k x k -> 1x1 -> BN -> ReLU]; the stride applies to
    every repeat (layer_factory.py:225-265)."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, dilation=1, affine=True,
                 repeats=1):
        super(SepConv, self).__init__()
        self.op = FusedSequential()
        width = C_in
        for idx in range(repeats):
            stage = FusedSequential(
                Conv2d(width, width, kernel_size=kernel_size, stride=stride, padding=padding,
                       dilation=dilation, groups=width, bias=False),
                Conv2d(width, C_out, kernel_size=1, padding=0, bias=False),
                BatchNorm2d(C_out, affine=affine),
                ReLU(inplace=False),
            )
            self.op.add_module("sep_{}".format(idx), stage)
            width = C_out

    def forward(self, x, defer_tail=False):
        """defer_tail: the last BatchNorm + ReLU may come back pending (functional.Pending) for an aggregation
        op that applies it as it loads (ConcatReduce)."""
        return self.op(x, defer_tail=defer_tail)


class Skip(nn.Module):
    """Channel repeat; ``stride`` is accepted and ignored (layer_factory.py:268-275)."""

    def __init__(self, C_in, C_out, stride):
        super(Skip, self).__init__()
        assert (C_out % C_in) == 0, "C_out must be divisible by C_in"
        self.repeats = (1, C_out // C_in, 1, 1)

    def forward(self, x):
        return F.channel_repeat(x, self.repeats[1])


class Identity(nn.Module):
    def __init__(self):
        super(Identity, self).__init__()

    def forward(self, x):
        return x


class Zero(nn.Module):
    """Zeros of the (channel-repeated, spatially strided) input shape (layer_factory.py:286-297)."""

    def __init__(self, C_in, C_out, stride):
        super(Zero, self).__init__()
        self.stride = stride
        assert (C_out % C_in) == 0, "C_out must be divisible by C_in"
        self.repeats = (1, C_out // C_in, 1, 1)

    def forward(self, x):
        B, C, H, W = x.shape
        s = self.stride
        return F.zero_of(x, C * self.repeats[1], (H + s - 1) // s, (W + s - 1) // s)


class FactorizedReduce(nn.Module):
    """ReLU -> two stride-2 1x1 convs (the second on the map shifted by one pixel) -> cat -> BN
    (layer_factory.py:300-313; no caller in the reference - kept so that the module namespace is
    complete).  An odd spatial size makes the two halves differ by one pixel: RuntimeError, as
    torch.cat would raise."""

    def __init__(self, C_in, C_out, affine=True):
        super(FactorizedReduce, self).__init__()
        assert C_out % 2 == 0
        self.relu = ReLU(inplace=False)
        self.conv_1 = Conv2d(C_in, C_out // 2, 1, stride=2, padding=0, bias=False)
        self.conv_2 = Conv2d(C_in, C_out // 2, 1, stride=2, padding=0, bias=False)
        self.bn = BatchNorm2d(C_out, affine=affine)

    def forward(self, x):
        x = self.relu(x)
        a, b = self.conv_1(x), self.conv_2(x[:, :, 1:, 1:])
        if tuple(a.shape[2:]) != tuple(b.shape[2:]):
            raise RuntimeError("Sizes of tensors must match except in dimension 1")
        return self.bn(F.concat_resize([a, b], a.shape[2:]))


def resize(x1, x2, largest=True):
    """Bilinearly bring the two maps to a common size: the larger one if
    ``largest`` else the smaller.  Sizes are compared as (H, W) tuples, i.e.
    lexicographically, exactly like torch.Size (layer_factory.py:338-350)."""
    s1, s2 = tuple(x1.size()[2:]), tuple(x2.size()[2:])
    if s1 == s2:
        return x1, x2
    first_is_target = (s1 > s2) if largest else (s1 < s2)
    if first_is_target:
        return x1, F.bilinear_resize(x2, s1)
    return F.bilinear_resize(x1, s2), x2


class Adapt(nn.Module):
    """Optional 1x1 conv+BN+ReLU per input to ``C_out`` channels, then resize()
    (layer_factory.py:316-335)."""

    def __init__(self, C_in0, C_in1, C_out, larger):
        super(Adapt, self).__init__()
        self.C_in0, self.C_in1, self.C_out = C_in0, C_in1, C_out
        if C_in0 != C_out:
            self.conv0 = conv_bn_relu(C_in0, C_out, 1, 1, 0)
        if C_in1 != C_out:
            self.conv1 = conv_bn_relu(C_in1, C_out, 1, 1, 0)
        self.larger = larger

    def convs(self, x1, x2, defer_tail=False):
        """The channel adaptation alone (ConcatReduce resizes as it writes its slab)."""
        # (a pending input is the 1x1 conv's prologue: conv_chain takes it)
        if self.C_in0 != self.C_out:
            x1 = self.conv0(x1, defer_tail=defer_tail)
        if self.C_in1 != self.C_out:
            x2 = self.conv1(x2, defer_tail=defer_tail)
        return x1, x2

    def target_size(self, x1, x2):
        """(H, W) resize() brings both maps to."""
        s1, s2 = tuple(x1.size()[2:]), tuple(x2.size()[2:])
        if s1 == s2:
            return s1
        return (s1 if s1 > s2 else s2) if self.larger else (s1 if s1 < s2 else s2)

    def forward(self, x1, x2):
        x1, x2 = self.convs(F.materialize(x1), F.materialize(x2))
        return resize(x1, x2, self.larger)


class ParamSum(nn.Module):
    """a[c]*x + b[c]*y with learnable per-channel a, b (layer_factory.py:353-366)."""

    def __init__(self, C_in0, C_in1, C_out, larger):
        super(ParamSum, self).__init__()
        self.adapt = Adapt(C_in0, C_in1, C_out, larger)
        self.a = nn.Parameter(torch.ones(C_out))
        self.b = nn.Parameter(torch.ones(C_out))

    accepts_pending = True  # (inputs may be functional.Pending: a producer's BatchNorm + ReLU still to apply)

    def forward(self, x, y):
        # Adapt's 1x1 convs take pending inputs as their prologue and may leave their own tail pending; an operand
        # that has to be resized needs the finished map, the other one stays pending: F.param_sum applies it
        x, y = self.adapt.convs(x, y, defer_tail=True)
        size = self.adapt.target_size(x, y)
        if tuple(x.size()[2:]) != size:
            x = F.bilinear_resize(F.materialize(x), size)
        if tuple(y.size()[2:]) != size:
            y = F.bilinear_resize(F.materialize(y), size)
        return F.param_sum(x, y, self.a, self.b)


# elements per input above which ConcatReduce skips the concatenation (F.cat_bn_relu_conv: no 2C-wide slab in
# memory); below it the one-node slab path (F.cat_reduce) wins - measured on the headline step with the
# threshold at 2^24 / 2^25 / 2^26 elements: 251.2 / 252.6 / 254.1 img/s (its largest input has 2^25)
_SPLIT_CAT_MIN = int(os.environ.get("NASSEG_SPLIT_CAT_MIN", 1 << 27))


class ConcatReduce(nn.Module):
    """cat -> BN(2C) -> ReLU -> 1x1 (2C -> C) (layer_factory.py:369-382)."""

    def __init__(self, C_in0, C_in1, C_out, affine=True, repeats=1, larger=True):
        super(ConcatReduce, self).__init__()
        self.adapt = Adapt(C_in0, C_in1, C_out, larger)
        self.conv1x1 = FusedSequential(
            BatchNorm2d(2 * C_out, affine=affine),
            ReLU(inplace=False),
            Conv2d(2 * C_out, C_out, 1, stride=1, padding=0, bias=False),
        )

    accepts_pending = True  # (inputs may be functional.Pending: a producer's BatchNorm + ReLU still to apply)

    def forward(self, x, y):
        bn, conv = self.conv1x1[0], self.conv1x1[2]
        plain_bn = bn.affine and bn.momentum is not None and bn.track_running_stats
        x, y = self.adapt.convs(x, y, defer_tail=plain_bn)
        size = self.adapt.target_size(x, y)
        C = x.shape[1]
        split = (x.shape[0] * size[0] * size[1] * C >= _SPLIT_CAT_MIN and C % 4 == 0
                 and conv.weight.shape[0] % 4 == 0 and plain_bn and x.shape[1] == y.shape[1])
        if not split and plain_bn and F.cat_reduce_ok(x, y, conv.weight):
            # one node: both inputs written into the slab (resized, their producers' pending BatchNorm + ReLU
            # applied on load) with the slab's statistics, then the 1x1 conv normalising as it loads
            return F.cat_reduce(x, y, size, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked if bn.training else None, conv.weight, bn.training,
                                bn.momentum, bn.eps)
        x, y = F.materialize(x), F.materialize(y)
        if split:
            # large maps: no slab, two pointwise convs with the BatchNorm halves applied on load
            x, y = resize(x, y, self.adapt.larger)
            return F.cat_bn_relu_conv(x, y, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      bn.num_batches_tracked if bn.training else None, conv.weight,
                                      bn.training, bn.momentum, bn.eps)
        # (the input of the other size is interpolated straight into its half of the slab)
        z = F.concat_resize([x, y], size)
        if (F.FUSE_BN_RELU_CONV and (2 * C) % 4 == 0 and conv.weight.shape[0] % 4 == 0 and bn.affine
                and bn.momentum is not None and bn.track_running_stats):
            # the slab's BatchNorm + ReLU are applied by the 1x1 conv as it loads (no normalised slab)
            return F.bn_relu_conv(z, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  bn.num_batches_tracked if bn.training else None, conv.weight, bn.training,
                                  bn.momentum, bn.eps)
        return self.conv1x1(z)


def run_op(op, x, defer_tail=False):
    """op(x); with defer_tail the ops that end in conv + BatchNorm (+ ReLU) - SepConv, DilConv, the dense
    conv_bn_relu ops - may return a functional.Pending for an aggregation op that ``accepts_pending``."""
    if defer_tail and isinstance(op, (SepConv, DilConv, FusedSequential)):
        return op(x, defer_tail=True)
    return op(x)


# ---------------------------------------------------------------------------
# registries
# ---------------------------------------------------------------------------
# separable convs: name -> (kernel, padding, dilation)
_SEP_GEOMETRY = {
    "sep_conv_3x3": (3, 1, 1),
    "sep_conv_5x5": (5, 2, 1),
    "sep_conv_7x7": (7, 3, 1),
    "sep_conv_3x3_dil3": (3, 3, 3),
    "sep_conv_5x5_dil6": (5, 12, 6),
}
# DARTS-style dilated convs: name -> (kernel, padding, dilation)
_DIL_GEOMETRY = {"dil_conv_3x3": (3, 2, 2), "dil_conv_5x5": (5, 4, 2)}
# dense conv + BN + ReLU: name -> (kernel, dilation)
_DENSE_GEOMETRY = {"conv1x1": (1, 1), "conv3x3": (3, 1), "conv3x3_dil3": (3, 3),
                   "conv3x3_dil12": (3, 12)}


def _sep_factory(k, p, d):
    return lambda C_in, C_out, stride, affine, repeats=1: SepConv(
        C_in, C_out, k, stride, p, dilation=d, affine=affine, repeats=repeats)


def _dil_factory(k, p, d):
    return lambda C_in, C_out, stride, affine, repeats=1: DilConv(
        C_in, C_out, k, stride, p, d, affine=affine)


def _dense_factory(k, d):
    return lambda C_in, C_out, stride, affine, repeats=1: _dense_bn_relu(
        C_in, C_out, k, stride, affine, dilation=d)


OPS = {
    "none": lambda C_in, C_out, stride, affine, repeats=1: Zero(C_in, C_out, stride),
    "avg_pool_3x3": lambda C_in, C_out, stride, affine, repeats=1: Pool(
        C_in, C_out, stride, repeats, ksize=3, mode="avg"),
    "max_pool_3x3": lambda C_in, C_out, stride, affine, repeats=1: Pool(
        C_in, C_out, stride, repeats, ksize=3, mode="max"),
    "global_average_pool": lambda C_in, C_out, stride, affine, repeats=1: GAPConv1x1(C_in, C_out),
    "skip_connect": lambda C_in, C_out, stride, affine, repeats=1: Skip(C_in, C_out, stride),
}
OPS.update({name: _sep_factory(*geo) for name, geo in _SEP_GEOMETRY.items()})
OPS.update({name: _dil_factory(*geo) for name, geo in _DIL_GEOMETRY.items()})
OPS.update({name: _dense_factory(*geo) for name, geo in _DENSE_GEOMETRY.items()})

AGG_OPS = {
    "psum": lambda C_in0, C_in1, C_out, affine, repeats=1, larger=True: ParamSum(
        C_in0, C_in1, C_out, larger),
    "cat": lambda C_in0, C_in1, C_out, affine, repeats=1, larger=True: ConcatReduce(
        C_in0, C_in1, C_out, affine=affine, repeats=repeats, larger=larger),
}
