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

# bumped by whatever swaps sub-modules of a built network (TemplateDecoder._reset_clf): the
# engine caches a candidate's parameter lists and re-walks the tree when this changes
TREE_VERSION = [0]


def _require_width(channels, what):
    """The kernels move four channels (one float4) per lane: every feature-map width on the path
    must be a multiple of 4 (the image's 3 and the class logits are the exceptions the dense conv
    handles).  The reference accepts any width; refusing at CONSTRUCTION with a ValueError - which
    the engine's try_except does not swallow - beats every candidate silently scoring 0."""
    if channels % 4 != 0:
        raise ValueError("nasseg: {} = {} is not a multiple of 4 (choose agg_size / width_mult "
                         "accordingly; see INTEGRATION.md)".format(what, channels))


class Conv2d(nn.Conv2d):
    """nn.Conv2d: dense (groups=1) on the fp32 MFMA path, or depthwise (groups=C)."""

    def __init__(self, *args, **kwargs):
        super(Conv2d, self).__init__(*args, **kwargs)
        if self.groups > 1 and self.groups == self.in_channels == self.out_channels:
            _require_width(self.in_channels, "depthwise conv channels")

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

    def __init__(self, num_features, *args, **kwargs):
        super(BatchNorm2d, self).__init__(num_features, *args, **kwargs)
        _require_width(num_features, "BatchNorm2d features")

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


def _flatten(mods):
    out = []
    for m in mods:
        if isinstance(m, FusedSequential):
            out.extend(_flatten(m._modules.values()))
        else:
            out.append(m)
    return out


def _chainable(m):
    """a conv the fused chain node can run: bias-free, dense with N % 4 == 0 or depthwise"""
    if not isinstance(m, Conv2d) or m.bias is not None:
        return False
    if m.is_depthwise:
        return m.in_channels % 4 == 0
    return m.groups == 1 and m.out_channels % 4 == 0


def _bn_fusable(m):
    return (isinstance(m, BatchNorm2d) and m.momentum is not None and m.track_running_stats
            and m.weight is not None)


def run_fused(modules, x, residual=None, relu_in=False, pool=None, defer_tail=False):
    """FusedSequential.forward over an explicit list of modules - also used to run SEVERAL fused
    sequences as one (MobileNetV2 merges consecutive blocks whose boundary nobody else reads: the
    normalised activation between them is then never written).  relu_in: the input is to be passed
    through a ReLU first (the decoders' F.relu ahead of pre_clf); fused into the first conv's loads
    when the sequence starts with a conv.  defer_tail: when the sequence ENDS in a conv chain whose last
    BatchNorm (+ activation) is still pending, return it as a functional.Pending for a consumer that applies
    it on load (ConcatReduce) instead of writing the normalised map."""
    mods = _flatten(modules)
    n = len(mods)
    i = 0
    if isinstance(x, F.Pending) and not (n and _chainable(mods[0]) and not relu_in):
        x = x.materialize()  # (only a conv at the head of the sequence applies a pending input as its prologue)
    res_used = residual is None
    if relu_in and not (n and _chainable(mods[0])):
        x, relu_in = F.relu(x), False
    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        in_act0 = F.ACT_RELU if (relu_in and i == 0) else F.ACT_NONE
        if (isinstance(m, nn.ReLU) and not isinstance(m, nn.ReLU6) and _chainable(nxt)
                and nxt.is_depthwise):
            in_act0 = F.ACT_RELU  # DilConv: ReLU applied as the depthwise conv loads
            i += 1
            m = mods[i]
        if _chainable(m):
            ops = []
            while i < n and _chainable(mods[i]):
                conv = mods[i]
                conv._check()
                i += 1
                bn, act = None, F.ACT_NONE
                if i < n and _bn_fusable(mods[i]):
                    b = mods[i]
                    i += 1
                    if i < n and isinstance(mods[i], nn.ReLU6):
                        act = F.ACT_RELU6
                        i += 1
                    elif i < n and isinstance(mods[i], nn.ReLU):
                        act = F.ACT_RELU
                        i += 1
                    bn = (b.weight, b.bias, b.running_mean, b.running_var, b.num_batches_tracked,
                          b.training, b.momentum, b.eps)
                ops.append((conv.weight, conv.stride[0], conv.padding[0], conv.dilation[0],
                            conv.is_depthwise, bn, act))
            res = None
            if not res_used and i == n:
                res, res_used = residual, True
            tail = None
            if (pool is not None and i == n and res is None and ops[-1][5] is not None and ops[-1][6] == F.ACT_NONE
                    and F.FUSE_POOL_BN):
                tail, pool = pool, None  # (3x3 max pooling fused behind the chain's last BatchNorm)
            x = F.conv_chain(x, ops, in_act0, res, tail,
                             defer_tail=defer_tail and i == n and res is None and pool is None and res_used)
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
        else:
            x = m(x)
            i += 1
    if not res_used:
        x = F.add(x, residual)
    if pool is not None:  # (not fusable: the plain pooling op)
        x = F.max_pool2d(x, pool[0], pool[1], pool[2])
    return x


class FusedSequential(nn.Sequential):
    """nn.Sequential whose forward runs maximal runs of [conv (BN (ReLU|ReLU6)?)?]+ as one fused
    autograd node (functional.conv_chain: statistics in the conv epilogue, normalise-on-read
    between the convs, residual add in the last normalise pass).  Nested FusedSequentials
    (SepConv stages) are flattened first; anything else runs module by module."""

    def forward(self, x, residual=None, relu_in=False, pool=None, defer_tail=False):
        """relu_in: the input is to be passed through a ReLU first (the decoders' F.relu ahead of
        pre_clf); fused into the first conv's loads when the sequence starts with a conv.
        pool = (3, stride, 1): 3x3 max pooling of the result (Pool), fused behind a final BatchNorm."""
        return run_fused(self._modules.values(), x, residual, relu_in, pool, defer_tail)

