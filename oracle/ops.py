"""Functional restatement of the op registry (reference src/nn/layer_factory.py).

Every op is a pure function of ``(sd, prefix, x, ...)`` where ``sd`` maps the
reference's state_dict key names to CPU tensors (leaf tensors with
requires_grad when gradients are wanted).  BatchNorm running buffers inside
``sd`` are updated in place in training mode, like the modules do.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (kernel, padding, dilation) - layer_factory.py:41-55,76-81
SEP_GEOMETRY = {
    "sep_conv_3x3": (3, 1, 1), "sep_conv_5x5": (5, 2, 1), "sep_conv_7x7": (7, 3, 1),
    "sep_conv_3x3_dil3": (3, 3, 3), "sep_conv_5x5_dil6": (5, 12, 6),
}
DIL_GEOMETRY = {"dil_conv_3x3": (3, 2, 2), "dil_conv_5x5": (5, 4, 2)}
DENSE_GEOMETRY = {"conv1x1": (1, 0, 1), "conv3x3": (3, 1, 1), "conv3x3_dil3": (3, 3, 3),
                  "conv3x3_dil12": (3, 12, 12)}


def batch_norm(sd, p, x, training):
    """nn.BatchNorm2d(eps=1e-5, momentum=0.1) forward; updates running stats in training."""
    w, b = sd.get(p + ".weight"), sd.get(p + ".bias")
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    y = F.batch_norm(x, rm, rv, w, b, training, BN_MOMENTUM, BN_EPS)
    if training and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"] += 1
    return y


def bilinear(x, size):
    """nn.Upsample(size=size, mode='bilinear') (align_corners=False)."""
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)


def conv_bn(sd, p, x, training, stride=1, padding=0, dilation=1, relu=False):
    """Sequential(conv, bn[, relu]) with children named 0, 1[, 2]."""
    y = F.conv2d(x, sd[p + ".0.weight"], sd.get(p + ".0.bias"), stride, padding, dilation)
    y = batch_norm(sd, p + ".1", y, training)
    return F.relu(y) if relu else y


def sep_conv(sd, p, x, k, pad, dil, stride, repeats, training):
    # layer_factory.py:225-265 - the stride is applied in EVERY repeat
    for r in range(repeats):
        q = "{}.op.sep_{}".format(p, r)
        c = x.shape[1]
        x = F.conv2d(x, sd[q + ".0.weight"], None, stride, pad, dil, groups=c)
        x = F.conv2d(x, sd[q + ".1.weight"])
        x = F.relu(batch_norm(sd, q + ".2", x, training))
    return x


def dil_conv(sd, p, x, k, pad, dil, stride, training):
    # layer_factory.py:198-222 - ReLU -> depthwise -> pointwise -> BN
    c = x.shape[1]
    x = F.relu(x)
    x = F.conv2d(x, sd[p + ".op.1.weight"], None, stride, pad, dil, groups=c)
    x = F.conv2d(x, sd[p + ".op.2.weight"])
    return batch_norm(sd, p + ".op.3", x, training)


def pool(sd, p, x, mode, stride, training):
    # layer_factory.py:161-178 - 1x1 conv + BN first, then the pooling
    x = conv_bn(sd, p + ".conv1x1", x, training)
    if mode == "avg":
        return F.avg_pool2d(x, 3, stride, 1, count_include_pad=False)
    return F.max_pool2d(x, 3, stride, 1)


def gap_conv(sd, p, x, training):
    # layer_factory.py:181-195
    size = x.shape[2:]
    y = x.mean(2, keepdim=True).mean(3, keepdim=True)
    y = conv_bn(sd, p + ".conv1x1", y, training, relu=True)
    return bilinear(y, size)


def apply_op(name, sd, p, x, C_in, C_out, stride, repeats, training):
    """OPS[name](C_in, C_out, stride, True, repeats)(x) with parameters sd[p.*]."""
    if name == "none":
        y = x.repeat(1, C_out // C_in, 1, 1)
        return (y if stride == 1 else y[:, :, ::stride, ::stride]).mul(0.0)
    if name == "skip_connect":
        return x.repeat(1, C_out // C_in, 1, 1)  # stride ignored (layer_factory.py:268-275)
    if name == "avg_pool_3x3":
        return pool(sd, p, x, "avg", stride, training)
    if name == "max_pool_3x3":
        return pool(sd, p, x, "max", stride, training)
    if name == "global_average_pool":
        return gap_conv(sd, p, x, training)
    if name in SEP_GEOMETRY:
        k, pad, dil = SEP_GEOMETRY[name]
        return sep_conv(sd, p, x, k, pad, dil, stride, repeats, training)
    if name in DIL_GEOMETRY:
        k, pad, dil = DIL_GEOMETRY[name]
        return dil_conv(sd, p, x, k, pad, dil, stride, training)
    if name in DENSE_GEOMETRY:
        k, pad, dil = DENSE_GEOMETRY[name]
        return conv_bn(sd, p, x, training, stride, pad, dil, relu=True)
    raise KeyError(name)


def resize_pair(x1, x2, largest):
    # layer_factory.py:338-350 - torch.Size comparison is lexicographic (H, then W)
    s1, s2 = tuple(x1.shape[2:]), tuple(x2.shape[2:])
    if largest:
        if s1 > s2:
            x2 = bilinear(x2, s1)
        elif s1 < s2:
            x1 = bilinear(x1, s2)
    else:
        if s1 < s2:
            x2 = bilinear(x2, s1)
        elif s1 > s2:
            x1 = bilinear(x1, s2)
    return x1, x2


def adapt(sd, p, x1, x2, larger, training):
    # layer_factory.py:316-335 - a 1x1 conv+BN+ReLU exists only when channels differ
    if (p + ".conv0.0.weight") in sd:
        x1 = conv_bn(sd, p + ".conv0", x1, training, relu=True)
    if (p + ".conv1.0.weight") in sd:
        x2 = conv_bn(sd, p + ".conv1", x2, training, relu=True)
    return resize_pair(x1, x2, larger)


def apply_agg(name, sd, p, x, y, larger, training):
    """AGG_OPS[name](...)(x, y) with parameters sd[p.*]."""
    x, y = adapt(sd, p + ".adapt", x, y, larger, training)
    if name == "psum":
        a, b = sd[p + ".a"], sd[p + ".b"]
        return a[None, :, None, None] * x + b[None, :, None, None] * y
    if name == "cat":
        z = torch.cat([x, y], 1)
        z = F.relu(batch_norm(sd, p + ".conv1x1.0", z, training))
        return F.conv2d(z, sd[p + ".conv1x1.2.weight"])
    raise KeyError(name)
