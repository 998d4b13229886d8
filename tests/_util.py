"""Shared helpers of the test-suite: golden-vector access, model builders."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def sub_dict(npz, prefix, tensor=True):
    """{key-without-prefix: array} for every entry below ``prefix/``."""
    pre = prefix + "/"
    out = {}
    for k in npz.files:
        if k.startswith(pre):
            v = npz[k]
            out[k[len(pre):]] = torch.from_numpy(np.array(v)) if tensor else v
    return out


def checksums(sd):
    return {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}


def assert_checksums_close(got, want, rtol=1e-5, atol=1e-6, what=""):
    assert set(got) == set(want), "{}: key sets differ: {}".format(
        what, sorted(set(got) ^ set(want))[:6])
    for k in want:
        g, w = np.array(got[k]), np.array(want[k])
        assert np.allclose(g, w, rtol=rtol, atol=atol), "{} {}: {} vs {}".format(what, k, g, w)


def build_product_net(kind, genotype, classes, dec_kwargs, seed):
    """The product Segmenter built on the HOST with the reference's seeding
    protocol (torch.manual_seed, encoder first, decoder second)."""
    from nas_segm_amd.engine import Segmenter
    from nas_segm_amd.nn.encoders import mbv2
    from nas_segm_amd.nn.micro_decoders import MicroDecoder, TemplateDecoder

    torch.manual_seed(seed)
    if kind == "template":
        enc = mbv2(pretrained=False, return_layers=[1, 2])
        dec = TemplateDecoder(inp_sizes=enc.out_sizes, num_classes=classes, config=genotype,
                              **dec_kwargs)
    else:
        enc = mbv2(pretrained=False)
        dec = MicroDecoder(inp_sizes=list(enc.out_sizes), num_classes=classes, config=genotype,
                           **dec_kwargs)
    return Segmenter(enc, dec)


def oracle_forward(sd, x, rec, training=False):
    """oracle.nets.segmenter with the bookkeeping a golden 'nets' record carries."""
    from oracle import nets

    kind = rec["kind"]
    kw = dict(rec["dec_kwargs"])
    if kind == "template":
        return nets.segmenter(sd, x, "template", rec["genotype"], [24, 32], (1, 2), training,
                              repeats=kw.get("repeats", 1))
    return nets.segmenter(sd, x, "micro", rec["genotype"], None, (1, 2, 4, 6), training,
                          agg_size=kw.get("agg_size", 64), aux_cell=kw.get("aux_cell", False),
                          repeats=kw.get("repeats", 1))


def clone_sd(sd, grad_keys=None):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if grad_keys is not None and k in grad_keys and t.is_floating_point():
            t.requires_grad_(True)
        out[k] = t
    return out


def max_err(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    assert tuple(a.shape) == tuple(b.shape), "shape {} vs {}".format(tuple(a.shape), tuple(b.shape))
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max())


def assert_close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    assert tuple(a.shape) == tuple(b.shape), "{}: shape {} vs {}".format(what, tuple(a.shape), tuple(b.shape))
    if a.numel() == 0:
        return
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), "{}: max err {:.3e} (tol {:.1e}+{:.1e}*|ref|, |ref|max {:.3e}, {} / {} bad)".format(
        what, float(err.max()), atol, rtol, float(b.abs().max()), int(bad.sum()), a.numel())


def randomise_bn(module, seed):
    """tests/golden/make_golden.py:randomize_bn - BatchNorm parameters and buffers away from their defaults,
    in module order, from one seeded generator (what the teacher record was made with)"""
    gen = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)


def build_product_teacher(meta):
    """the product's rf_lw152 built on the HOST with the record's seeding protocol"""
    from nas_segm_amd.kd import rf_lw152

    torch.manual_seed(meta["seed"])
    net = rf_lw152(pretrained=False, num_classes=meta["num_classes"])
    randomise_bn(net, meta["bn_seed"])
    return net.eval()
