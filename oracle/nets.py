"""Functional restatement of the decoders and the encoder
(reference src/nn/micro_decoders.py, src/nn/encoders.py) on top of oracle.ops.
"""
import torch
import torch.nn.functional as F

from . import ops

# src/rl/genotypes.py:8-35 - index -> registry key
OP_NAMES = ["conv1x1", "conv3x3", "sep_conv_3x3", "sep_conv_5x5", "global_average_pool",
            "conv3x3_dil3", "conv3x3_dil12", "sep_conv_3x3_dil3", "sep_conv_5x5_dil6",
            "skip_connect", "none"]
OP_NAMES_WACV = ["sep_conv_3x3", "sep_conv_5x5", "global_average_pool", "max_pool_3x3",
                 "sep_conv_5x5_dil6", "skip_connect"]
AGG_OP_NAMES = ["psum", "cat"]

# src/nn/encoders.py:19-27 - (t, c, n, s)
MBV2_CONFIG = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1],
               [6, 160, 3, 2], [6, 320, 1, 1]]


def collect_all(feats, inds):
    # micro_decoders.py:11-25 - only H is compared; the running concat is up-sampled as a whole
    out = feats[inds[0]]
    for i in inds[1:]:
        c = feats[i]
        if out.shape[2] > c.shape[2]:
            c = ops.bilinear(c, out.shape[2:])
        elif c.shape[2] > out.shape[2]:
            out = ops.bilinear(out, c.shape[2:])
        out = torch.cat([out, c], 1)
    return out


def sum_to_larger(x1, x2):
    # micro_decoders.py:46-51
    s1, s2 = tuple(x1.shape[2:]), tuple(x2.shape[2:])
    if s1 > s2:
        x2 = ops.bilinear(x2, s1)
    elif s1 < s2:
        x1 = ops.bilinear(x1, s2)
    return x1 + x2


def clf_head(sd, x, training):
    x = F.relu(x)
    x = ops.conv_bn(sd, "pre_clf", x, training, relu=True)
    return F.conv2d(x, sd["conv_clf.weight"], sd["conv_clf.bias"], 1, 1)


# ---------------------------------------------------------------------------
# TemplateDecoder (micro_decoders.py:257-398)
# ---------------------------------------------------------------------------
def template_decoder(sd, config, feats, inp_sizes, repeats=1, stride_power=1, training=False,
                     prefix=""):
    templates, structure = config
    n_scales = len(inp_sizes)
    chans = list(inp_sizes) + [0] * len(structure)
    feats = list(feats)
    collect = []
    for blk, (pos1, pos2, cell_id, n_rep, s_log2) in enumerate(structure):
        larger = blk >= len(structure) // 2
        n_rep += 1
        stride = 2 ** s_log2
        op1, op2, agg = templates[cell_id]
        for pos in (pos1, pos2):
            if pos in collect:
                collect.remove(pos)
        f1, f2 = feats[pos1], feats[pos2]
        new_c, prev_c, agg_c = [0, 0], [0, 0], None
        for r in range(n_rep):
            outs = []
            for li, (pos, op_id, f) in enumerate(((pos1, op1, f1), (pos2, op2, f2))):
                if r == 0:
                    cin = chans[pos]
                    cout = cin * int(stride ** stride_power)
                elif li == 0:
                    cin = cout = prev_c[-1]
                else:
                    cin = cout = agg_c
                new_c[li], prev_c[li] = cout, cin
                p = "{}_ops.{}.{}".format(prefix, blk, r * 3 + li)
                outs.append(ops.apply_op(OP_NAMES_WACV[op_id], sd, p, f, cin, cout, stride, repeats,
                                         training))
            agg_c = max(new_c)
            p = "{}_ops.{}.{}".format(prefix, blk, r * 3 + 2)
            out2 = ops.apply_agg(AGG_OP_NAMES[agg], sd, p, outs[0], outs[1], larger, training)
            f1, f2 = f2, out2
        chans[n_scales + blk] = agg_c
        feats.append(out2)
        collect.append(n_scales + blk)
    return clf_head(sd, collect_all(feats, collect), training)


# ---------------------------------------------------------------------------
# MicroDecoder (micro_decoders.py:54-254)
# ---------------------------------------------------------------------------
def contextual_cell(sd, p, cfg, x, C, repeats, training):
    feats = [x]
    collect = [0]
    op_i = 0
    for ind, entry in enumerate(cfg):
        if ind == 0:
            collect.remove(0)
            feats.append(ops.apply_op(OP_NAMES[entry], sd, "{}._ops.{}".format(p, op_i), x, C, C, 1,
                                      repeats, training))
            op_i += 1
            collect.append(1)
        else:
            pos1, pos2, o1, o2 = entry
            for pos, o in ((pos1, o1), (pos2, o2)):
                if pos in collect:
                    collect.remove(pos)
                feats.append(ops.apply_op(OP_NAMES[o], sd, "{}._ops.{}".format(p, op_i), feats[pos],
                                          C, C, 1, repeats, training))
                op_i += 1
            feats.append(sum_to_larger(feats[ind * 3 - 1], feats[ind * 3]))
            op_i += 1  # the parameter-free sum node also occupies a slot in _ops
            collect.append(ind * 3 + 1)
    out = 0
    for i in collect:
        out = out + feats[i]
    return out


def micro_decoder(sd, config, feats, agg_size=64, num_pools=4, aux_cell=False, repeats=1,
                  training=False):
    cell_cfg, conns = config
    x = [ops.conv_bn(sd, "adapt{}".format(i + 1), f, training, relu=True)
         for i, f in enumerate(feats)]
    collect = []
    aux = []
    for blk, (i1, i2) in enumerate(conns):
        for ind in (i1, i2):
            if ind in collect:
                collect.remove(ind)
        p = "cells.{}".format(blk)
        a = contextual_cell(sd, p + ".op_1", cell_cfg, x[i1], agg_size, repeats, training)
        b = contextual_cell(sd, p + ".op_2", cell_cfg, x[i2], agg_size, repeats, training)
        a = ops.conv_bn(sd, p + ".agg.branch_1", a, training, relu=True)
        b = ops.conv_bn(sd, p + ".agg.branch_2", b, training, relu=True)
        out = sum_to_larger(a, b)
        x.append(out)
        h = out
        q = "aux_clfs.{}".format(blk)
        if aux_cell:
            h = contextual_cell(sd, q + ".aux_cell", cell_cfg, h, agg_size, repeats, training)
        aux.append(F.conv2d(h, sd[q + ".aux_clf.weight"], sd[q + ".aux_clf.bias"], 1, 1))
        collect.append(blk + num_pools)
    return clf_head(sd, collect_all(x, collect), training), aux


# ---------------------------------------------------------------------------
# MobileNetV2 encoder (encoders.py:15-63, layer_factory.py:109-158)
# ---------------------------------------------------------------------------
def inverted_residual(sd, p, x, stride, training):
    inp = x.shape[1]
    q = p + ".conv"
    y = F.conv2d(x, sd[q + ".0.weight"])
    y = F.hardtanh(ops.batch_norm(sd, q + ".1", y, training), 0.0, 6.0)
    y = F.conv2d(y, sd[q + ".3.weight"], None, stride, 1, 1, groups=y.shape[1])
    y = F.hardtanh(ops.batch_norm(sd, q + ".4", y, training), 0.0, 6.0)
    y = F.conv2d(y, sd[q + ".6.weight"])
    y = ops.batch_norm(sd, q + ".7", y, training)
    if stride == 1 and inp == y.shape[1]:
        y = x + y
    return y


def mbv2(sd, x, return_layers=(1, 2, 4, 6), training=False, prefix=""):
    y = F.conv2d(x, sd[prefix + "layer1.0.weight"], None, 2, 1)
    y = F.hardtanh(ops.batch_norm(sd, prefix + "layer1.1", y, training), 0.0, 6.0)
    outs = []
    for li, (t, c, n, s) in enumerate(MBV2_CONFIG[: max(return_layers) + 1]):
        for i in range(n):
            y = inverted_residual(sd, "{}layer{}.{}".format(prefix, li + 2, i), y,
                                  s if i == 0 else 1, training)
        outs.append(y)
    return [outs[i] for i in return_layers]


class _Sub(object):
    """View of a state dict under a key prefix (``encoder.`` / ``decoder.``)."""

    def __init__(self, sd, prefix):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def __setitem__(self, k, v):
        self.sd[self.prefix + k] = v

    def __contains__(self, k):
        return (self.prefix + k) in self.sd

    def get(self, k, default=None):
        return self.sd.get(self.prefix + k, default)


def segmenter(sd, x, kind, config, inp_sizes, return_layers, training=False, **dec_kwargs):
    """Segmenter(encoder, decoder)(x) with keys ``encoder.*`` / ``decoder.*``."""
    feats = mbv2(_Sub(sd, "encoder."), x, return_layers, training)
    dsd = _Sub(sd, "decoder.")
    if kind == "template":
        return template_decoder(dsd, config, feats, inp_sizes, training=training, **dec_kwargs)
    return micro_decoder(dsd, config, feats, training=training, **dec_kwargs)
