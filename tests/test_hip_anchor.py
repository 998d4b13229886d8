"""Every FUSED kernel of the conv chains anchored directly to a plain PyTorch fp32 CPU reference of the same
ops (round 2 compared each fused kernel with the kernel pair it replaced): conv -> BatchNorm(train) -> act
sequences with the headline network's channel pairs on maps scaled to 1 x 64 x 128, thresholds lowered so
that the large-map paths run (nasseg_conv_pw_bwd_bn incl. its wide and dx_stats forms, nasseg_dwconv_bwd_bn,
nasseg_conv_wgrad_bn, nasseg_conv_wgrad_bn_flat, nasseg_sepconv_fwd, nasseg_conv_bwd_data_bn,
nasseg_dwconv_bwd_data_bn) - output, input gradient, every parameter gradient and the BatchNorm buffers
against torch.nn's own forward / autograd of the SAME module tree on the CPU
(reference: src/nn/layer_factory.py:76-81,125-158,225-265 run these ops through torch.nn).
Also: BatchNorm statistics when |mean| >> std."""
import copy

import pytest
import torch
import torch.nn as nn

from _util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def F():
    from nas_segm_amd import functional

    return functional


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def dev(t):
    t = t.to(DEV)
    return t.contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t


def torch_reference(modules, x, residual=None, relu_in=False):
    """the torch.nn forward of the (flattened) module list: every product leaf subclasses the torch.nn
    module the reference instantiates, so its base class' forward IS the reference op"""
    from nas_segm_amd.nn.modules import _flatten

    y = torch.relu(x) if relu_in else x
    for m in _flatten(modules):
        base = [c for c in type(m).__mro__ if c.__module__.startswith("torch.nn")][0]
        y = base.forward(m, y)
    return y + residual if residual is not None else y


def lower_thresholds(monkeypatch):
    Fm = F()
    for name in ("_GROUP_WGRAD_BYTES", "_PW_BWD_MIN_BYTES", "_PW_BWD_WIDE_MIN_PIXELS", "_DW_BWD_MIN_BYTES",
                 "_FLAT_WGRAD_BN_MIN_BYTES"):
        monkeypatch.setattr(Fm, name, 0)
    # ... and the one-kernel pointwise backward rebuilds z = W x on these small maps as it does on the large ones
    prev = Fm.lib.query("nasseg_conv_pw_bwd_rz_min_pixels", -1)
    Fm.lib.query("nasseg_conv_pw_bwd_rz_min_pixels", 0)
    Fm.lib._memo.clear()

    class _Restore(object):  # (monkeypatch undoes attribute patches in reverse order: this one restores the knob)
        def __setattr__(self, name, value):
            if name == "armed" and value is False:
                Fm.lib.query("nasseg_conv_pw_bwd_rz_min_pixels", prev)
                Fm.lib._memo.clear()
            object.__setattr__(self, name, value)

    r = _Restore()
    object.__setattr__(r, "armed", False)
    monkeypatch.setattr(r, "armed", True)  # undone at teardown -> armed = False -> knob restored
    return Fm


def randomise(mods, seed):
    """BatchNorm affine parameters and buffers away from their (1, 0, 0, 1) defaults"""
    g = torch.Generator().manual_seed(seed)
    for m in mods.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def build(kind):
    from nas_segm_amd.nn.layer_factory import (OPS, InvertedResidual, conv_bn_relu, conv_bn_relu6)
    from nas_segm_amd.nn.modules import FusedSequential

    torch.manual_seed(11)
    if kind == "ir_16_24_s2":      # 16 -> 96 (one-kernel pointwise backward), dw 96 stride 2, 96 -> 24
        return InvertedResidual(16, 24, 2, 6).conv, 16, (64, 128), False, False
    if kind == "ir_24_24":         # 24 -> 144, dw 144, 144 -> 24 + residual
        return InvertedResidual(24, 24, 1, 6).conv, 24, (64, 128), True, False
    if kind == "ir_32_32":         # 32 -> 192, dw 192, 192 -> 32 + residual
        return InvertedResidual(32, 32, 1, 6).conv, 32, (32, 64), True, False
    if kind == "pre_clf":          # relu -> 224 -> 64 + BN + ReLU: the wave-split one-kernel backward
        return conv_bn_relu(224, 64, 1, 1, 0), 224, (64, 128), False, True
    if kind == "stem":             # 3 -> 32 3x3 stride 2 + BN + ReLU6: BatchNorm backward on the wgrad loads
        return conv_bn_relu6(3, 32, 2), 3, (128, 256), False, False
    if kind == "stem_stage1_2":    # the merged encoder chain: 32 -> 32, 16 -> 96 behind a BatchNorm (dx_stats)
        return FusedSequential(conv_bn_relu6(3, 32, 2), InvertedResidual(32, 16, 1, 1).conv,
                               InvertedResidual(16, 24, 2, 6).conv), 3, (128, 256), False, False
    if kind == "sep3x3_r2":        # SepConv: dw -> pw as one forward kernel, twice
        return OPS["sep_conv_3x3"](32, 32, 1, True, 2).op, 32, (64, 128), False, False
    if kind == "sep5x5_24_64":
        return OPS["sep_conv_5x5"](24, 64, 1, True, 1).op, 24, (64, 128), False, False
    if kind == "sep5x5_dil6":
        return OPS["sep_conv_5x5_dil6"](32, 32, 1, True, 2).op, 32, (64, 128), False, False
    if kind == "dil3x3":
        return OPS["dil_conv_3x3"](32, 32, 1, True).op, 32, (64, 128), False, False
    if kind == "pool_s2":          # Pool: 1x1 conv + BatchNorm -> 3x3 max pooling as one node (24 -> 48, stride 2)
        return OPS["max_pool_3x3"](24, 48, 2, True), 24, (64, 127), False, False
    if kind == "pool_s1":
        return OPS["max_pool_3x3"](32, 32, 1, True), 32, (33, 64), False, False
    raise KeyError(kind)


EXPECT = {  # entry points the case is there for (fp32 names)
    "ir_16_24_s2": ("nasseg_conv_pw_bwd_bn", "nasseg_dwconv_bwd_bn"),
    "ir_24_24": ("nasseg_conv_pw_bwd_bn", "nasseg_dwconv_bwd_bn"),
    "ir_32_32": ("nasseg_conv_pw_bwd_bn", "nasseg_dwconv_bwd_bn"),
    "pre_clf": ("nasseg_conv_pw_bwd_bn",),
    "stem": ("nasseg_conv_wgrad_bn_flat",),
    "stem_stage1_2": ("nasseg_conv_wgrad_bn_flat", "nasseg_conv_pw_bwd_bn", "nasseg_dwconv_bwd_bn"),
    "sep3x3_r2": ("nasseg_sepconv_fwd", "nasseg_conv_pw_bwd_bn"),
    "sep5x5_24_64": ("nasseg_sepconv_fwd",),
    "sep5x5_dil6": ("nasseg_sepconv_fwd",),
    "dil3x3": (),
    "pool_s2": ("nasseg_maxpool_bn_fwd", "nasseg_maxpool_bn_bwd"),
    "pool_s1": ("nasseg_maxpool_bn_fwd", "nasseg_maxpool_bn_bwd"),
}


# ... with InvertedResidual's expansion never stored (functional._irdw_ok, csrc/irdw.hip): statistics from the moments of
# the block's input, the expanded map rebuilt inside the depthwise forward / backward kernels and the pointwise backward
EXPECT_IRDW = {
    "ir_16_24_s2": ("nasseg_irdw_stats", "nasseg_irdw_fwd", "nasseg_irdw_bwd", "nasseg_conv_pw_bwd_bn"),
    "ir_24_24": ("nasseg_irdw_stats", "nasseg_irdw_fwd", "nasseg_irdw_bwd", "nasseg_conv_pw_bwd_bn"),
    "stem_stage1_2": ("nasseg_conv_wgrad_bn_flat", "nasseg_irdw_stats", "nasseg_irdw_fwd", "nasseg_irdw_bwd",
                      "nasseg_conv_pw_bwd_bn"),
}


@pytest.mark.parametrize("batch", [1, 3])
@pytest.mark.parametrize("irdw", [False, True], ids=["stored", "rebuilt"])
@pytest.mark.parametrize("kind", sorted(EXPECT))
def test_fused_chain_kernels_against_torch_cpu_autograd(kind, irdw, batch, monkeypatch):
    """batch 3: several images per slab of the one-kernel backwards, tiles of the persistent pointwise kernel that
    straddle image boundaries; rebuilt: the expansions of the InvertedResidual cases are never stored"""
    if batch > 1 and kind in ("stem", "stem_stage1_2"):
        pytest.skip("the 128 x 256 image cases stay at one image (the float64 reference of three takes a minute)")
    if irdw and kind not in EXPECT_IRDW:
        pytest.skip("no InvertedResidual expansion that is served in this case")
    Fm = lower_thresholds(monkeypatch)
    monkeypatch.setattr(Fm, "IRDW", irdw)
    monkeypatch.setattr(Fm, "_IRDW_MIN_PIXELS", 0)
    mods, cin, (H, W), residual, relu_in = build(kind)
    randomise(mods, 5)
    ref = copy.deepcopy(mods).train()
    ref64 = copy.deepcopy(mods).double().train()
    mods = mods.to(DEV).train()
    # (seed 2 at three images puts ONE pre-activation of ir_24_24 within rounding of the ReLU6 kink: the gradient then
    # differs in the 3 x 3 neighbourhood of that pixel and nowhere else - tools/diag_anchor_b3.py)
    x0 = rnd(batch, cin, H, W, seed=2 if batch == 1 else 7)
    xc = x0.clone().requires_grad_(cin > 3)  # (the image needs no gradient: the stem's flat weight-gradient path)
    is_pool = kind.startswith("pool")

    def reference(mod, xin):
        if is_pool:  # (src/nn/layer_factory.py:161-178: conv1x1 + BN, then nn.MaxPool2d)
            return nn.MaxPool2d.forward(mod.pool, torch_reference(mod.conv1x1._modules.values(), xin))
        return torch_reference(mod._modules.values(), xin, xin if residual else None, relu_in)

    yc = reference(ref, xc)
    cot = rnd(*yc.shape, seed=3)
    yc.backward(cot)
    # the same graph in float64: how far the fp32 REFERENCE is from the exact result bounds how close
    # anything can be to it - e.g. the bias of a BatchNorm that feeds conv -> BatchNorm has an
    # analytically zero gradient, and the reference's value for it is rounding noise of sums over
    # 32768 pixels (2e-4 here, next to gradients of order 1)
    xd = x0.double().requires_grad_(cin > 3)
    yd = reference(ref64, xd)
    yd.backward(cot.double())

    seen = []
    orig = Fm.lib.call

    def rec(fn, *a):
        seen.append(fn)
        return orig(fn, *a)

    monkeypatch.setattr(Fm.lib, "call", rec)
    xg = dev(x0.clone()).requires_grad_(cin > 3)
    yg = mods(xg) if is_pool else mods(xg, residual=xg if residual else None, relu_in=relu_in)
    yg.backward(dev(cot))
    monkeypatch.setattr(Fm.lib, "call", orig)
    for name in (EXPECT_IRDW if irdw else EXPECT)[kind]:
        assert name in seen, "{}: {} did not run ({})".format(kind, name, sorted(set(seen)))
    if irdw and kind != "stem_stage1_2":
        assert "nasseg_dwconv_bwd_bn" not in seen and "nasseg_dwconv" not in seen

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach().double()).abs().max()) / (float(b.abs().max()) + 1e-30)

    gp, cp, dp = dict(mods.named_parameters()), dict(ref.named_parameters()), dict(ref64.named_parameters())

    def floor(ref32, ref64_):  # the fp32 reference's own distance from the float64 result
        return float((ref32.detach().double() - ref64_.detach()).abs().max())

    worst = {"y": rel(yg, yc)}
    frac = 0.0
    if cin > 3:
        # an element whose pre-activation sits within rounding of a ReLU / ReLU6 kink may take the other
        # branch: at most 2e-5 of the elements may miss, everything else to 1e-4 of the tensor's max
        err = (xg.grad.cpu().double() - xc.grad.double()).abs()
        tol = 1e-4 * float(xc.grad.abs().max()) + 1e-4 * xc.grad.abs().double() + 4 * floor(xc.grad, xd.grad)
        frac = float((err > tol).double().mean())
        worst["dx"] = rel(xg.grad, xc.grad)
        worst["dx_frac_off"] = frac
    for k in cp:
        worst["d" + k] = float((gp[k].grad.cpu().double() - cp[k].grad.double()).abs().max()) / (
            float(cp[k].grad.abs().max()) + 4 * floor(cp[k].grad, dp[k].grad) + 1e-30)
    print("ANCHOR {:14s} ".format(kind) + " ".join("{}={:.1e}".format(k, v) for k, v in sorted(
        worst.items(), key=lambda kv: -kv[1])[:6]))
    assert_close(yg, yc, 1e-4 * float(yc.abs().max()), 1e-4, kind + ": output")
    assert frac <= 2e-5, "{}: dx: {:.2e} of the elements off (max rel err {:.2e})".format(kind, frac, worst["dx"])
    for k in cp:
        assert_close(gp[k].grad, cp[k].grad, 1e-4 * float(cp[k].grad.abs().max()) + 4 * floor(cp[k].grad, dp[k].grad),
                     1e-4, "{}: gradient of {}".format(kind, k))
    gb, cb = dict(mods.named_buffers()), dict(ref.named_buffers())
    for k in cb:
        if cb[k].dtype == torch.int64:
            assert int(gb[k]) == int(cb[k]), k
        else:
            assert_close(gb[k], cb[k], 1e-6, 2e-5, "{}: buffer {}".format(kind, k))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("kind", ["sep3x3_r2", "ir_16_24_s2", "stem_stage1_2"])
def test_the_rebuilt_conv_output_trains_like_the_stored_one(kind, dtype, monkeypatch):
    """nasseg_conv_pw_bwd_bn with z == NULL rebuilds z = W x from the input tile it stages (conv_pwbwd.hip, RZ); with z
    given it loads what the forward stored.  A chain trained through a fused SepConv / merged encoder forward and either
    backward ends in the same bits - output, input gradient, every parameter gradient - in fp32 and with bfloat16
    storage (where the rebuilt value is rounded like the stored one)"""
    Fm = lower_thresholds(monkeypatch)
    monkeypatch.setattr(Fm, "IRDW", False)
    mods, cin, (H, W), residual, relu_in = build(kind)
    randomise(mods, 5)
    mods = mods.to(DEV).train()
    state = copy.deepcopy(mods.state_dict())
    x0 = dev(rnd(2 if kind != "stem_stage1_2" else 1, cin, H, W, seed=2)).to(dtype)
    cot = None
    got = {}
    for mode, min_pixels in (("rebuilt", 0), ("stored", 1 << 40)):
        Fm.lib.query("nasseg_conv_pw_bwd_rz_min_pixels", min_pixels)
        Fm.lib._memo.clear()
        mods.load_state_dict(state)
        for q in mods.parameters():
            q.grad = None
        seen = []
        orig = Fm.lib.call

        def rec(fn, *a):
            if fn.endswith("conv_pw_bwd_bn"):
                seen.append(a[2] is None)  # (the z argument)
            return orig(fn, *a)

        monkeypatch.setattr(Fm.lib, "call", rec)
        xg = x0.clone().requires_grad_(cin > 3)
        yg = mods(xg, residual=xg if residual else None, relu_in=relu_in)
        if cot is None:
            cot = dev(rnd(*yg.shape, seed=3)).to(dtype)
        yg.backward(cot)
        monkeypatch.setattr(Fm.lib, "call", orig)
        assert seen and (any(seen) if mode == "rebuilt" else not any(seen)), (mode, seen)
        got[mode] = (yg.detach().clone(), xg.grad.clone() if cin > 3 else None,
                     dict((k, q.grad.clone()) for k, q in mods.named_parameters()))
    ya, xa, pa = got["rebuilt"]
    yb, xb, pb = got["stored"]
    assert torch.equal(ya, yb)
    if xa is not None:
        assert torch.equal(xa, xb)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k


@pytest.mark.parametrize("cfg", [(24, 24, 6, 33, 40), (64, 64, 6, 16, 20), (32, 32, 6, 20, 28)])
def test_residual_gradient_in_the_backward_data_epilogue(cfg, monkeypatch):
    """InvertedResidual on a small map (src/nn/layer_factory.py:276-321): the block's input is also its skip, and the
    plain backward-data call of its first conv adds the skip's gradient in its epilogue (functional.FUSE_RES_GRAD)
    instead of leaving dx + dres to an accumulation by autograd - same bits as that, and torch-CPU autograd's values"""
    from nas_segm_amd.nn.layer_factory import InvertedResidual

    Fm = F()
    cin, cout, t, H, W = cfg
    torch.manual_seed(4)
    mods = InvertedResidual(cin, cout, 1, t).conv
    randomise(mods, 6)
    ref = copy.deepcopy(mods).train()
    mods = mods.to(DEV).train()
    x0 = rnd(2, cin, H, W, seed=9)
    xc = x0.clone().requires_grad_(True)
    yc = torch_reference(ref._modules.values(), xc, xc)
    cot = rnd(*yc.shape, seed=10)
    yc.backward(cot)

    def run(fuse):
        monkeypatch.setattr(Fm, "FUSE_RES_GRAD", fuse)
        seen = []
        orig = Fm.lib.call

        def rec(fn, *a):
            seen.append((fn, a))
            return orig(fn, *a)

        monkeypatch.setattr(Fm.lib, "call", rec)
        for prm in mods.parameters():
            prm.grad = None
        xg = dev(x0.clone()).requires_grad_(True)
        yg = mods(xg, residual=xg)
        yg.backward(dev(cot))
        monkeypatch.setattr(Fm.lib, "call", orig)
        # (the plain backward-data call: nasseg_conv_fwd with transposed = 1 or the flipped-weights form; its
        #  residual pointer is argument 11)
        with_res = [a for fn, a in seen if fn == "nasseg_conv_fwd" and a[11]]
        assert "nasseg_conv_pw_bwd_bn" not in [fn for fn, _ in seen]
        return xg.grad.clone(), with_res

    g1, r1 = run(True)
    g0, r0 = run(False)
    assert len(r1) == len(r0) + 1
    assert torch.equal(g0, g1)
    assert_close(g1, xc.grad, 1e-4 * float(xc.grad.abs().max()), 1e-4, "dx")


@pytest.mark.parametrize("path", ["bn_stats", "conv_epilogue", "dw_epilogue"])
@pytest.mark.parametrize("ratio", [5.0, 50.0])
def test_batchnorm_statistics_when_the_mean_dwarfs_the_deviation(path, ratio):
    """train-mode BatchNorm of a tensor with |mean| = ratio * std (per channel, either sign): mean, biased
    variance (through invstd) and the normalised output against float64.  The statistics are the
    producer conv's epilogue sums (conv_epilogue: 1x1 conv with an identity weight; dw_epilogue: 3x3
    depthwise with a centre tap of 1) or nasseg_bn_stats (bn_stats)."""
    Fm = F()
    B, C, H, W = 2, 32, 96, 128
    g = torch.Generator().manual_seed(7)
    sign = torch.where(torch.rand(C, generator=g) > 0.5, 1.0, -1.0)
    std = torch.rand(C, generator=g) + 0.5
    x = torch.randn(B, C, H, W, generator=g) * std.view(1, C, 1, 1) + (sign * ratio * std).view(1, C, 1, 1)
    xd = x.double()
    mean64 = xd.mean(dim=(0, 2, 3))
    var64 = xd.var(dim=(0, 2, 3), unbiased=False)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    want = ((xd - mean64.view(1, C, 1, 1)) / torch.sqrt(var64 + 1e-5).view(1, C, 1, 1) * gamma.double().view(1, C, 1, 1)
            + beta.double().view(1, C, 1, 1))
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    xg = dev(x)
    if path == "bn_stats":
        y = Fm.batch_norm_act(xg, gamma.to(DEV), beta.to(DEV), rm, rv, nbt, True, 0.1, 1e-5, 0, None)
    elif path == "conv_epilogue":
        w = torch.eye(C).view(C, C, 1, 1).to(DEV)
        y = Fm.conv_bn_act(xg, w, gamma.to(DEV), beta.to(DEV), rm, rv, nbt, True, 0.1, 1e-5, 0)
    else:
        w = torch.zeros(C, 1, 3, 3)
        w[:, 0, 1, 1] = 1.0
        bn = (gamma.to(DEV), beta.to(DEV), rm, rv, nbt, True, 0.1, 1e-5)
        y = Fm.conv_chain(xg, [(w.to(DEV), 1, 1, 1, True, bn, 0)])
    unb = var64 * (B * H * W) / (B * H * W - 1)
    # running_mean = 0.1 * mean, running_var = 0.9 + 0.1 * unbiased variance: 1e-4 of the batch variance
    assert_close(rm.cpu().double() / 0.1, mean64, 1e-5 * float(std.max()) * ratio, 1e-6, "batch mean")
    got_var = (rv.cpu().double() - 0.9) / 0.1
    err = float(((got_var - unb) / unb).abs().max())
    print("BNVAR {} ratio {}: max rel err of the variance {:.2e}".format(path, ratio, err))
    assert err < 1e-4, "variance off by {:.2e} (relative) at |mean| = {} std".format(err, ratio)
    assert_close(y, want, 2e-4, 1e-4, "normalised output")


def test_a_failed_capture_leaves_batchnorm_and_gradients_untouched(monkeypatch):
    """NASSEG_GRAPH=auto captures by default: if the warm-up or the capture raises, the engine launches
    from the host - and the candidate's BatchNorm statistics must be what they were before the attempt
    (engine/graphed.py:_capture restores in a finally block), no half-written gradients left."""
    from nas_segm_amd.engine import graphed
    from nas_segm_amd.engine.trainer import _segmenter_stepper
    from nas_segm_amd.nn.encoders import mbv2
    from nas_segm_amd.nn.micro_decoders import TemplateDecoder
    from nas_segm_amd.engine import Segmenter

    torch.manual_seed(0)
    enc = mbv2(pretrained=False, return_layers=[1, 2])
    dec = TemplateDecoder(enc.out_sizes, 19, [[[3, 0, 1], [4, 1, 1], [3, 1, 1]],
                                              [[0, 1, 0, 0, 1], [2, 1, 2, 1, 0], [3, 1, 1, 1, 0]]], agg_size=32, repeats=1)
    net = Segmenter(enc, dec).to(DEV).train()
    oe = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9)
    od = torch.optim.Adam(net.decoder.parameters(), lr=3e-3)
    image = dev(rnd(2, 3, 65, 97, seed=1))
    target = torch.randint(0, 19, (2, 65, 97), generator=torch.Generator().manual_seed(2)).to(DEV)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    calls = {"n": 0}
    orig = graphed._GraphedStep._fwd_bwd

    def flaky(self, with_optimisers):
        calls["n"] += 1
        if calls["n"] == 2:  # (the second warm-up pass: the first has already updated every BatchNorm)
            raise RuntimeError("HIP out of memory (injected)")
        return orig(self, with_optimisers)

    monkeypatch.setattr(graphed._GraphedStep, "_fwd_bwd", flaky)
    stepper = _segmenter_stepper(net, image, target, oe, od, 255, 3.0, 3.0, -1)
    assert stepper is None and calls["n"] == 2
    after = net.state_dict()
    for k, v in before.items():
        assert torch.equal(v, after[k]), "{} changed by the failed capture".format(k)
    assert all(p.grad is None for p in net.parameters())
    # the failure is remembered for this shape (no second attempt), other shapes still get their try
    assert _segmenter_stepper(net, image, target, oe, od, 255, 3.0, 3.0, -1) is None and calls["n"] == 2
    monkeypatch.setattr(graphed._GraphedStep, "_fwd_bwd", orig)
    other = dev(rnd(2, 3, 49, 65, seed=3))
    t2 = target[:, :49, :65].contiguous()
    s2 = _segmenter_stepper(net, other, t2, oe, od, 255, 3.0, 3.0, -1)
    assert s2 is not None
    # ... and at most _STEPPERS_PER_CANDIDATE shapes are captured per candidate: the third runs from the host
    third = dev(rnd(2, 3, 33, 65, seed=4))
    assert _segmenter_stepper(net, third, target[:, :33, :65].contiguous(), oe, od, 255, 3.0, 3.0, -1) is None
    # freezing a parameter invalidates the captures (the graph has its gradient baked in)
    next(net.decoder.parameters()).requires_grad_(False)
    s3 = _segmenter_stepper(net, other, t2, oe, od, 255, 3.0, 3.0, -1)
    assert s3 is not None and s3 is not s2


# ---------------------------------------------------------------------------
# ConcatReduce as one node fed by its producers' pending BatchNorm + ReLU (functional._CatReduce / Pending)
# ---------------------------------------------------------------------------
def _op_reference(op, x):
    """torch.nn forward of an OPS module (src/nn/layer_factory.py:161-265)"""
    if hasattr(op, "pool"):
        return nn.MaxPool2d.forward(op.pool, torch_reference(op.conv1x1._modules.values(), x))
    return torch_reference(op.op._modules.values(), x)


def _cell_reference(cell, x1, x2):
    """op1(x1), op2(x2) -> Adapt (1x1 conv + BN + ReLU where widths differ, then the bilinear resize to the
    larger / smaller of the two sizes, src/nn/layer_factory.py:316-350) -> cat -> BN -> ReLU -> 1x1 conv
    (:369-382), all through torch.nn / torch.nn.functional on the CPU"""
    op1, op2, agg = cell
    a, b = _op_reference(op1, x1), _op_reference(op2, x2)
    ad = agg.adapt
    if ad.C_in0 != ad.C_out:
        a = torch_reference(ad.conv0._modules.values(), a)
    if ad.C_in1 != ad.C_out:
        b = torch_reference(ad.conv1._modules.values(), b)
    s1, s2 = tuple(a.shape[2:]), tuple(b.shape[2:])
    if s1 != s2:
        first = (s1 > s2) if ad.larger else (s1 < s2)
        if first:
            b = torch.nn.functional.interpolate(b, size=s1, mode="bilinear", align_corners=False)
        else:
            a = torch.nn.functional.interpolate(a, size=s2, mode="bilinear", align_corners=False)
    return torch_reference(agg.conv1x1._modules.values(), torch.cat([a, b], 1))


def _build_cell(kind):
    from nas_segm_amd.nn.layer_factory import AGG_OPS, OPS

    torch.manual_seed(13)
    if kind == "sep_sep_down":     # the headline decoder's cell: 128 x 256 and 32 x 64 maps meet at the smaller
        return nn.ModuleList([OPS["sep_conv_3x3"](32, 32, 1, True, 2), OPS["sep_conv_5x5"](32, 32, 1, True, 2),
                              AGG_OPS["cat"](32, 32, 32, True, 2, False)]), (32, (64, 128)), (32, (16, 32))
    if kind == "sep_dil_up":       # ... at the larger; a DilConv (BatchNorm without ReLU) on the small map
        return nn.ModuleList([OPS["sep_conv_3x3"](24, 24, 1, True, 1), OPS["dil_conv_3x3"](24, 24, 1, True),
                              AGG_OPS["cat"](24, 24, 24, True, 1, True)]), (24, (33, 65)), (24, (9, 17))
    if kind == "pool_sep_same":    # a finished tensor (Pool) next to a pending one, same size: no resize
        return nn.ModuleList([OPS["max_pool_3x3"](32, 32, 1, True), OPS["sep_conv_3x3"](32, 32, 1, True, 1),
                              AGG_OPS["cat"](32, 32, 32, True, 1, True)]), (32, (32, 64)), (32, (32, 64))
    if kind == "adapt_conv":       # widths differ: Adapt's 1x1 conv + BN + ReLU is the pending producer
        return nn.ModuleList([OPS["sep_conv_3x3"](24, 24, 1, True, 1), OPS["sep_conv_3x3"](48, 48, 1, True, 1),
                              AGG_OPS["cat"](24, 48, 48, True, 1, True)]), (24, (32, 64)), (48, (16, 32))
    raise KeyError(kind)


@pytest.mark.parametrize("kind,mode", [("sep_sep_down", "train"), ("sep_dil_up", "train"), ("pool_sep_same", "train"),
                                       ("adapt_conv", "train"), ("sep_sep_down", "eval"), ("pool_sep_same", "eval")])
def test_cat_reduce_cell_against_torch_cpu_autograd(kind, mode, monkeypatch):
    """mode eval: every BatchNorm on its running statistics WITH gradients (the engine's freeze_bn mode,
    src/engine/trainer.py:124-127,219-222) - nothing folds, the same pending-tail path with statistics from the buffers"""
    from nas_segm_amd.nn.layer_factory import run_op

    Fm = lower_thresholds(monkeypatch)
    cell, (c1, hw1), (c2, hw2) = _build_cell(kind)
    randomise(cell, 6)
    ref = copy.deepcopy(cell).train(mode == "train")
    ref64 = copy.deepcopy(cell).double().train(mode == "train")
    cell = cell.to(DEV).train(mode == "train")
    x1, x2 = rnd(2, c1, *hw1, seed=2), rnd(2, c2, *hw2, seed=4)
    xs = [x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)]
    yc = _cell_reference(ref, *xs)
    cot = rnd(*yc.shape, seed=3)
    yc.backward(cot)
    xd = [x1.double().requires_grad_(True), x2.double().requires_grad_(True)]
    yd = _cell_reference(ref64, *xd)
    yd.backward(cot.double())

    seen = []
    orig = Fm.lib.call

    def rec(fn, *a):
        seen.append(fn)
        return orig(fn, *a)

    monkeypatch.setattr(Fm.lib, "call", rec)
    xg = [dev(x1.clone()).requires_grad_(True), dev(x2.clone()).requires_grad_(True)]
    assert cell[2].accepts_pending
    a, b = run_op(cell[0], xg[0], True), run_op(cell[1], xg[1], True)
    n_pending = sum(isinstance(t, Fm.Pending) for t in (a, b))
    yg = cell[2](a, b)
    yg.backward(dev(cot))
    monkeypatch.setattr(Fm.lib, "call", orig)
    assert seen.count("nasseg_cat_src_fwd") == 2 and seen.count("nasseg_cat_src_bwd") == 2, sorted(set(seen))
    # a pending producer at least as large as the slab gets its BatchNorm-backward sums from nasseg_cat_src_bwd
    # (behind a down-sampling they are formed at the slab's size and nasseg_bilinear_bwd_act masks the transposed
    # gradient); what still reduces: adapt_conv's first SepConv - its consumer is Adapt's 1x1 conv, whose plain
    # backward-data hands back the gradient w.r.t. the activated input, unmasked and without sums
    # (... and producers SMALLER than the slab, which keep their own, cheap, reduction: sep_dil_up's DilConv and
    #  adapt_conv's second SepConv)
    want_reduce = {"sep_sep_down": 0, "sep_dil_up": 1, "pool_sep_same": 0, "adapt_conv": 2}[kind]
    n_reduce = seen.count("nasseg_bn_bwd_reduce") + seen.count("nasseg_bn_bwd_reduce_rows")
    assert n_reduce == want_reduce, n_reduce
    assert ("nasseg_bilinear_bwd_act" in seen) == (kind == "sep_sep_down")
    assert "nasseg_chan_copy" not in seen
    if kind != "adapt_conv":
        assert n_pending == (1 if kind == "pool_sep_same" else 2)
    # no pass of its own over the slab for the statistics, none over the producers' outputs to normalise them
    assert "nasseg_bn_stats" not in seen
    # (adapt_conv: Adapt's 1x1 conv takes the SepConv's pending output as its prologue)
    assert seen.count("nasseg_affine_act") == 0, seen.count("nasseg_affine_act")

    def floor(ref32, ref64_):
        return float((ref32.detach().double() - ref64_.detach()).abs().max())

    assert_close(yg, yc, 1e-4 * float(yc.abs().max()) + 4 * floor(yc, yd), 1e-4, kind + ": output")
    worst = {}
    for i in range(2):
        err = (xg[i].grad.cpu().double() - xs[i].grad.double()).abs()
        tol = (1e-4 * float(xs[i].grad.abs().max()) + 1e-4 * xs[i].grad.abs().double()
               + 4 * floor(xs[i].grad, xd[i].grad))
        frac = float((err > tol).double().mean())
        worst["dx{}_frac_off".format(i)] = frac
        assert frac <= 2e-5, "{}: dx{}: {:.2e} of the elements off".format(kind, i, frac)
    gp, cp, dp = dict(cell.named_parameters()), dict(ref.named_parameters()), dict(ref64.named_parameters())
    for k in cp:
        fl = floor(cp[k].grad, dp[k].grad)
        worst["d" + k] = float((gp[k].grad.cpu().double() - cp[k].grad.double()).abs().max()) / (
            float(cp[k].grad.abs().max()) + 4 * fl + 1e-30)
        assert_close(gp[k].grad, cp[k].grad, 1e-4 * float(cp[k].grad.abs().max()) + 4 * fl, 1e-4,
                     "{}: gradient of {}".format(kind, k))
    print("ANCHOR cell {:14s} ".format(kind) + " ".join("{}={:.1e}".format(k, v) for k, v in sorted(
        worst.items(), key=lambda kv: -kv[1])[:6]))
    gb, cb = dict(cell.named_buffers()), dict(ref.named_buffers())
    for k in cb:
        if cb[k].dtype == torch.int64:
            assert int(gb[k]) == int(cb[k]), k
        else:
            assert_close(gb[k], cb[k], 1e-6, 2e-5, "{}: buffer {}".format(kind, k))
    # the node and the module-by-module path (NASSEG_FUSE_CAT_REDUCE=0) agree as well
    monkeypatch.setattr(Fm, "FUSE_CAT_REDUCE", False)
    with torch.no_grad():
        y_plain = cell[2](run_op(cell[0], xg[0], True), run_op(cell[1], xg[1], True))
    assert_close(y_plain, yc, 1e-4 * float(yc.abs().max()) + 4 * floor(yc, yd), 1e-4, kind + ": unfused output")


@pytest.mark.parametrize("kind", ["sep_sep_same", "sep_pool_same", "adapt_resize"])
def test_param_sum_cell_against_torch_cpu_autograd(kind, monkeypatch):
    """ParamSum (src/nn/layer_factory.py:353-366) fed by its producers' pending BatchNorm + ReLU (round 4:
    nasseg_add_act2 forward, nasseg_psum_bwd backward, the producers' sums handed over by the side): output, both
    input gradients, every parameter gradient (the coefficients a, b included) and the BatchNorm buffers against
    torch.nn on the CPU, floors from the same graph in float64."""
    from nas_segm_amd.nn.layer_factory import AGG_OPS, OPS, run_op

    Fm = lower_thresholds(monkeypatch)
    torch.manual_seed(17)
    if kind == "sep_sep_same":      # two pending operands of one size
        cell = nn.ModuleList([OPS["sep_conv_3x3"](32, 32, 1, True, 2), OPS["sep_conv_5x5"](32, 32, 1, True, 1),
                              AGG_OPS["psum"](32, 32, 32, True, 1, True)])
        (c1, hw1), (c2, hw2) = (32, (33, 64)), (32, (33, 64))
    elif kind == "sep_pool_same":   # a pending operand next to a finished one (Pool)
        cell = nn.ModuleList([OPS["sep_conv_3x3"](24, 24, 1, True, 1), OPS["max_pool_3x3"](24, 24, 1, True),
                              AGG_OPS["psum"](24, 24, 24, True, 1, True)])
        (c1, hw1), (c2, hw2) = (24, (32, 48)), (24, (32, 48))
    else:                           # widths and sizes differ: Adapt's 1x1 conv is the pending producer, the other
        cell = nn.ModuleList([OPS["sep_conv_3x3"](24, 24, 1, True, 1), OPS["sep_conv_3x3"](48, 48, 1, True, 1),
                              AGG_OPS["psum"](24, 48, 48, True, 1, True)])  # operand is resized (finished map)
        (c1, hw1), (c2, hw2) = (24, (16, 32)), (48, (32, 64))
    randomise(cell, 8)
    with torch.no_grad():
        cell[2].a.copy_(torch.rand(cell[2].a.shape) + 0.5)
        cell[2].b.copy_(torch.rand(cell[2].b.shape) + 0.5)
    ref = copy.deepcopy(cell).train()
    ref64 = copy.deepcopy(cell).double().train()
    cell = cell.to(DEV).train()

    def reference(c, x1, x2):
        a, b = _op_reference(c[0], x1), _op_reference(c[1], x2)
        ad = c[2].adapt
        if ad.C_in0 != ad.C_out:
            a = torch_reference(ad.conv0._modules.values(), a)
        if ad.C_in1 != ad.C_out:
            b = torch_reference(ad.conv1._modules.values(), b)
        s1, s2 = tuple(a.shape[2:]), tuple(b.shape[2:])
        if s1 != s2:
            if (s1 > s2) if ad.larger else (s1 < s2):
                b = torch.nn.functional.interpolate(b, size=s1, mode="bilinear", align_corners=False)
            else:
                a = torch.nn.functional.interpolate(a, size=s2, mode="bilinear", align_corners=False)
        return a * c[2].a.view(1, -1, 1, 1) + b * c[2].b.view(1, -1, 1, 1)

    x1, x2 = rnd(2, c1, *hw1, seed=2), rnd(2, c2, *hw2, seed=4)
    xs = [x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)]
    yc = reference(ref, *xs)
    cot = rnd(*yc.shape, seed=3)
    yc.backward(cot)
    xd = [x1.double().requires_grad_(True), x2.double().requires_grad_(True)]
    yd = reference(ref64, *xd)
    yd.backward(cot.double())

    seen = []
    orig = Fm.lib.call

    def rec(fn, *a):
        seen.append(fn)
        return orig(fn, *a)

    monkeypatch.setattr(Fm.lib, "call", rec)
    xg = [dev(x1.clone()).requires_grad_(True), dev(x2.clone()).requires_grad_(True)]
    assert cell[2].accepts_pending
    yg = cell[2](run_op(cell[0], xg[0], True), run_op(cell[1], xg[1], True))
    yg.backward(dev(cot))
    monkeypatch.setattr(Fm.lib, "call", orig)
    assert seen.count("nasseg_add_act2") == 1 and seen.count("nasseg_psum_bwd") == 1, sorted(set(seen))
    assert "nasseg_axpby" not in seen and "nasseg_colred" not in seen
    # producers whose pending output went straight into the sum need no reduction pass of their own
    n_reduce = seen.count("nasseg_bn_bwd_reduce") + seen.count("nasseg_bn_bwd_reduce_rows")
    assert n_reduce == {"sep_sep_same": 0, "sep_pool_same": 0, "adapt_resize": 2}[kind]
    assert not Fm._TAIL_ROWS

    def floor(ref32, ref64_):
        return float((ref32.detach().double() - ref64_.detach()).abs().max())

    assert_close(yg, yc, 1e-4 * float(yc.abs().max()) + 4 * floor(yc, yd), 1e-4, kind + ": output")
    for i in range(2):
        err = (xg[i].grad.cpu().double() - xs[i].grad.double()).abs()
        tol = (1e-4 * float(xs[i].grad.abs().max()) + 1e-4 * xs[i].grad.abs().double()
               + 4 * floor(xs[i].grad, xd[i].grad))
        frac = float((err > tol).double().mean())
        assert frac <= 2e-5, "{}: dx{}: {:.2e} of the elements off".format(kind, i, frac)
    gp, cp, dp = dict(cell.named_parameters()), dict(ref.named_parameters()), dict(ref64.named_parameters())
    for k in cp:
        fl = floor(cp[k].grad, dp[k].grad)
        assert_close(gp[k].grad, cp[k].grad, 1e-4 * float(cp[k].grad.abs().max()) + 4 * fl, 1e-4,
                     "{}: gradient of {}".format(kind, k))
    gb, cb = dict(cell.named_buffers()), dict(ref.named_buffers())
    for k in cb:
        if cb[k].dtype == torch.int64:
            assert int(gb[k]) == int(cb[k]), k
        else:
            assert_close(gb[k], cb[k], 1e-6, 2e-5, "{}: buffer {}".format(kind, k))


@pytest.mark.parametrize("nblk", [1, 37, 512, 513, 1024, 2047, 4096, 4097, 9000])
def test_partial_row_finalisation_levels(nblk):
    """nasseg_rows_sum / nasseg_bn_finalize over nblk partial rows: one 256-thread launch (<= 512 rows), one
    1024-thread launch (<= 4096), two levels above - all against float64 sums of the same rows."""
    Fm = F()
    C = 24
    g = torch.Generator().manual_seed(nblk)
    rows = torch.randn(nblk, 2 * C, generator=g)
    rows[:, C:] = rows[:, C:].abs() * 3 + rows[:, :C] ** 2  # (sum of squares >= square of sums / n)
    part = torch.zeros(nblk + 64, 2 * C)
    part[:nblk] = rows
    part = part.to(DEV)
    sums = torch.empty(2 * C, device=DEV)
    s = Fm.current_stream()
    Fm.lib.call("nasseg_rows_sum", Fm.ptr(part), nblk, 2 * C, Fm.ptr(sums), s)
    want = rows.double().sum(0)
    assert_close(sums, want, 1e-6 * float(want.abs().max()), 1e-6, "rows_sum")
    part2 = torch.zeros(nblk + 64, 2 * C)
    part2[:nblk] = rows
    part2 = part2.to(DEV)
    M = 16 * nblk
    out = [torch.empty(C, device=DEV) for _ in range(4)]
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    Fm.lib.call("nasseg_bn_finalize", Fm.ptr(part2), nblk, M, C, 1e-5, 0.1, Fm.ptr(gd), Fm.ptr(bd),
                Fm.ptr(out[0]), Fm.ptr(out[1]), Fm.ptr(out[2]), Fm.ptr(out[3]), Fm.ptr(rm), Fm.ptr(rv), Fm.ptr(nbt), s)
    mu = want[:C] / M
    var = (want[C:] / M - mu * mu).clamp_min(0)
    istd = 1.0 / torch.sqrt(var + 1e-5)
    assert_close(out[0], mu, 1e-6, 1e-5, "mean")
    assert_close(out[1], istd, 1e-6, 1e-5, "invstd")
    assert_close(out[2], gamma.double() * istd, 1e-6, 1e-5, "scale")
    assert_close(out[3], beta.double() - mu * gamma.double() * istd, 1e-5, 1e-5, "shift")
    assert_close(rm, 0.1 * mu, 1e-6, 1e-5, "running_mean")
    assert_close(rv, 0.9 + 0.1 * var * M / (M - 1), 1e-6, 1e-5, "running_var")
    assert int(nbt) == 1


@pytest.mark.parametrize("size", [(11, 11), (21, 23)], ids=["11x11", "21x23"])
@pytest.mark.parametrize("name", ["cvpr_arch0", "cvpr_arch1_search", "cvpr_arch2_depth"])
def test_contextual_cell_with_gradient_junctions_against_torch_cpu_autograd(name, size, monkeypatch):
    """A ContextualCell (src/nn/micro_decoders.py:54-121) of the three published / searched CVPR genotypes on the
    small maps of the CVPR decoder: the cell's input feeds up to five ops, op outputs with a pending BatchNorm + ReLU
    feed a sum AND another op (or a global-average-pool op that needs the finished map).  Round 5: every node with
    several consumers goes through ONE gradient junction (nasseg_grad_junction: the consumers' gradients summed,
    masked, with the producer's BatchNorm-backward rows) instead of autograd's pairwise adds, and the BatchNorm
    backwards add up their few partial rows inside the apply kernel (nasseg_bn_bwd_reduce_rows /
    nasseg_bn_bwd_apply_rows).  Output, input gradient, every parameter gradient and the BatchNorm buffers against
    torch's own autograd of the same graph on the CPU (oracle.nets.contextual_cell: plain torch.nn.functional),
    floors from the same graph in float64; then the same step with both switches off, which must agree with the
    junction path to rounding."""
    import json
    import os

    from nas_segm_amd.nn.micro_decoders import ContextualCell
    from oracle import nets as onets

    Fm = F()
    with open(os.path.join(os.path.dirname(__file__), "golden", "nets_meta.json")) as fh:
        cfg = json.load(fh)[name]["genotype"][0]
    C, repeats, B = 32, 2, 8
    torch.manual_seed(23)
    cell = ContextualCell(cfg, C, repeats=repeats)
    randomise(cell, 5)
    sd0 = {k: v.detach().clone() for k, v in cell.state_dict().items()}
    pkeys = {k for k, _ in cell.named_parameters()}
    x = rnd(B, C, *size, seed=2)
    cot = None

    def reference(dtype):
        sd = {"cell." + k: (v.to(dtype).clone().requires_grad_(True) if k in pkeys else
                            (v.to(dtype).clone() if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
        xin = x.to(dtype).clone().requires_grad_(True)
        out = onets.contextual_cell(sd, "cell", cfg, xin, C, repeats, True)
        return sd, xin, out

    sd32, x32, y32 = reference(torch.float32)
    cot = rnd(*y32.shape, seed=3)
    y32.backward(cot)
    sd64, x64, y64 = reference(torch.float64)
    y64.backward(cot.double())

    def floor(a, b):
        return float((a.detach().double() - b.detach()).abs().max())

    def run(junction, apply_rows):
        monkeypatch.setattr(Fm, "JUNCTION", junction)
        monkeypatch.setattr(Fm, "APPLY_ROWS", apply_rows)
        m = ContextualCell(cfg, C, repeats=repeats)
        m.load_state_dict(sd0)
        m = m.to(DEV).train()
        seen = []
        orig = Fm.lib.call

        def rec(fn, *a):
            seen.append(fn)
            return orig(fn, *a)

        monkeypatch.setattr(Fm.lib, "call", rec)
        xg = dev(x.clone()).requires_grad_(True)
        yg = m(xg)
        yg.backward(dev(cot))
        torch.cuda.synchronize()
        monkeypatch.setattr(Fm.lib, "call", orig)
        return m, xg, yg, seen

    m, xg, yg, seen = run(True, True)
    assert seen.count("nasseg_grad_junction") >= 2, sorted(set(seen))
    assert seen.count("nasseg_bn_bwd_apply_rows") >= 1 and seen.count("nasseg_bn_bwd_reduce_rows") >= 1
    assert not Fm._TAIL_ROWS
    assert_close(yg, y32, 1e-4 * float(y32.abs().max()) + 4 * floor(y32, y64), 1e-4, name + ": output")
    gx, rx = xg.grad.cpu(), x32.grad
    assert_close(gx, rx, 2e-4 * float(rx.abs().max()) + 4 * floor(rx, x64.grad), 1e-4, name + ": input gradient")
    gp = dict(m.named_parameters())
    for k in sorted(pkeys):
        ref, ref64 = sd32["cell." + k].grad, sd64["cell." + k].grad
        assert ref is not None and gp[k].grad is not None, k
        assert_close(gp[k].grad, ref, 2e-4 * float(ref.abs().max()) + 4 * floor(ref, ref64) + 1e-7, 1e-4,
                     "{}: gradient of {}".format(name, k))
    for k, v in m.named_buffers():
        ref = sd32["cell." + k]
        if ref.dtype == torch.int64:
            assert int(v) == int(ref), k
        else:
            assert_close(v, ref, 1e-6, 5e-5, "{}: buffer {}".format(name, k))
    # autograd's accumulation + the summing launches (round 4's path): same numbers to rounding
    m0, xg0, yg0, seen0 = run(False, False)
    assert "nasseg_grad_junction" not in seen0 and "nasseg_bn_bwd_apply_rows" not in seen0
    assert torch.equal(yg0, yg)  # (the forward is the same launch sequence)
    assert_close(xg0.grad, xg.grad, 2e-5 * float(rx.abs().max()) + 4 * floor(rx, x64.grad), 1e-4, "switches: input gradient")
    gp0 = dict(m0.named_parameters())
    for k in sorted(pkeys):
        ref, ref64 = sd32["cell." + k].grad, sd64["cell." + k].grad
        assert_close(gp0[k].grad, gp[k].grad, 2e-5 * float(ref.abs().max()) + 4 * floor(ref, ref64) + 1e-7, 1e-4,
                     "switches: gradient of {}".format(k))
