"""CPU-only checks of the host side: the C-ABI library loads and exports every
symbol of include/nasseg.h, the plugin boundary keeps the reference's names and
signatures, constructors reproduce the reference's state_dict layout, and the
product path refuses to compute on the host."""
import inspect

import numpy as np
import os

import pytest
import torch

from _util import load_json


def test_library_exports_every_declared_symbol():
    import ctypes

    from nas_segm_amd import lib
    from nas_segm_amd._lib import LIB_PATH, parse_header

    protos = parse_header()
    assert len(protos) >= 34
    dll = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), name
    assert set(lib.symbols()) == set(protos)
    assert lib.query("nasseg_abi_version") == 1
    assert isinstance(lib.last_error(), str)


def test_workspace_queries_are_pure_host_calls():
    from nas_segm_amd import lib

    assert lib.query("nasseg_ce_workspace") > 0
    assert lib.query("nasseg_colred_workspace", 1, 1 << 20, 64) >= 2 * 64
    assert lib.query("nasseg_conv_wgrad_workspace", 4, 256, 512, 64, 224, 1, 1) >= 64 * 224
    assert lib.query("nasseg_dwconv_wgrad_workspace", 4, 24, 256, 512, 5) >= 25 * 24
    # the output tile of the LDS-tiled 3x3 kernel (one statistics row per workgroup): 8 x 32 where that is as good as any
    # (the headline's 4 x 128 x 256 maps), fewer workgroups per CU where another shape of <= 256 pixels gives them (the
    # CVPR cells' 16 x 81 x 81 maps: 528 -> at most the 512 that run at once), small tiles on small maps (the depth head's
    # 8 x 30 x 40: 64 workgroups of four rounds of MFMAs -> up to 256 of one)
    assert lib.query("nasseg_conv_fwd_stats_rows", 4, 128, 256, 48, 48, 3, 3, 1, 1, 1) == 4 * 16 * 8
    assert lib.query("nasseg_conv_fwd_stats_rows", 64, 64, 64, 48, 48, 3, 3, 1, 1, 1) == 64 * 8 * 2
    assert lib.query("nasseg_conv_fwd_stats_rows", 16, 81, 81, 64, 64, 3, 3, 1, 1, 1) <= 512
    assert lib.query("nasseg_conv_fwd_stats_rows", 16, 81, 81, 64, 64, 3, 3, 1, 3, 3) <= 512
    assert 64 < lib.query("nasseg_conv_fwd_stats_rows", 8, 30, 40, 64, 64, 3, 3, 1, 1, 1) <= 256
    # slabs of the one-kernel pointwise backward: 1024 / 512 on large maps (four / two workgroups per CU), one slab per
    # tile and CU on small ones (32 -> 192 at 8 x 60 x 80 ran 120 slabs of five tiles on 120 of the 256 CUs)
    assert lib.query("nasseg_conv_pw_bwd_slabs", 4, 512, 1024, 16, 96) == 1024
    assert lib.query("nasseg_conv_pw_bwd_slabs", 4, 256, 512, 24, 144) == 512
    assert lib.query("nasseg_conv_pw_bwd_slabs", 8, 60, 80, 32, 192) == 200
    assert lib.query("nasseg_conv_pw_bwd_slabs", 16, 41, 41, 64, 64) == 211
    # the generic weight gradient's 4 x 4 form (N, K > 32): whole waves of the 1024 workgroups that run at once - slabs x
    # k-chunks x taps: 64 x 128 at 4 x 256 x 512 was 512 x 2 + a third of a wave more, the cells' 3 x 3 64 -> 64 at
    # 16 x 81 x 81 is 110 x 9 = 990
    assert lib.query("nasseg_conv_wgrad_workspace", 4, 256, 512, 64, 128, 1, 1) // (64 * 128) == 512
    assert lib.query("nasseg_conv_wgrad_workspace", 16, 81, 81, 64, 64, 3, 3) // (9 * 64 * 64) == 110


def test_error_convention_is_runtime_error():
    from nas_segm_amd import NassegError, lib

    assert issubclass(NassegError, RuntimeError)
    with pytest.raises(RuntimeError) as e:  # argument validation happens before any launch
        lib.call("nasseg_dwconv", None, None, None, None, None, 0, None, None, 0, 1, 4, 4, 6, 4, 4, 3, 1, 1, 1,
                 0, None, None)
    assert "multiple of 4" in str(e.value)


def test_host_miou_arithmetic_matches_oracle():
    from nas_segm_amd.helpers.miou_utils import compute_iu, compute_ius_accs
    from oracle import miou as omiou

    rng = np.random.RandomState(0)
    for n in (1, 3, 19, 21):
        cm = rng.randint(0, 1000, size=(n, n)).astype(np.int64)
        cm[:, n // 2] = 0
        cm[n // 2, :] = 0
        a, b = compute_ius_accs(cm), omiou.compute_ius_accs(cm)
        for u, v in zip(a, b):
            assert u.dtype == v.dtype and np.array_equal(u, v)
        assert np.array_equal(compute_iu(cm), b[0])
    big = np.zeros((2, 2), dtype=np.int64)
    big[1, 1] = 2 ** 32
    with pytest.raises(OverflowError):
        compute_ius_accs(big)
    # denominators wrap in 32-bit exactly like the C original
    wrap = np.array([[2 ** 31, 2 ** 31 - 10], [2 ** 31 - 1, 5]], dtype=np.int64)
    for u, v in zip(compute_ius_accs(wrap), omiou.compute_ius_accs(wrap)):
        assert np.array_equal(u, v)


def test_registry_keys_and_signatures():
    from nas_segm_amd.nn.layer_factory import AGG_OPS, OPS
    from nas_segm_amd.rl.genotypes import AGG_OP_NAMES, OP_NAMES, OP_NAMES_WACV

    assert sorted(OPS) == sorted([
        "none", "avg_pool_3x3", "max_pool_3x3", "global_average_pool", "skip_connect",
        "sep_conv_3x3", "sep_conv_5x5", "sep_conv_7x7", "dil_conv_3x3", "dil_conv_5x5", "conv1x1",
        "conv3x3", "conv3x3_dil3", "conv3x3_dil12", "sep_conv_3x3_dil3", "sep_conv_5x5_dil6"])
    assert sorted(AGG_OPS) == ["cat", "psum"]
    for f in OPS.values():
        assert list(inspect.signature(f).parameters) == ["C_in", "C_out", "stride", "affine", "repeats"]
    for f in AGG_OPS.values():
        assert list(inspect.signature(f).parameters) == ["C_in0", "C_in1", "C_out", "affine", "repeats", "larger"]
    assert len(OP_NAMES) == 11 and len(OP_NAMES_WACV) == 6 and AGG_OP_NAMES == ["psum", "cat"]
    assert all(n in OPS for n in OP_NAMES + OP_NAMES_WACV)
    assert OP_NAMES[8] == "sep_conv_5x5_dil6" and OP_NAMES_WACV[3] == "max_pool_3x3"
    with pytest.raises(AssertionError):
        OPS["skip_connect"](8, 12, 1, True)


def test_controller_samples_build_valid_decoders():
    """configs sampled by the reference controller (golden) drive the decoders unchanged"""
    from nas_segm_amd.nn.micro_decoders import MicroDecoder, TemplateDecoder

    ctrl = load_json("controller.json")
    assert ctrl["cvpr"]["action_size"] == 20 and ctrl["wacv"]["action_size"] == 44
    for s in ctrl["cvpr"]["samples"]:
        sizes = [24, 32, 96, 320]
        dec = MicroDecoder(sizes, 21, s["config"], agg_size=48, aux_cell=True, repeats=1)
        assert sizes == [48] * 4  # the reference's in-place mutation of inp_sizes is kept
        assert len(dec.cells) == 3 and isinstance(dec.info, str)
        assert "#Contextual" in dec.prettify(1000)
    for s in ctrl["wacv"]["samples"]:
        dec = TemplateDecoder([24, 32], 19, s["config"], agg_size=48, repeats=1)
        assert len(dec._ops) == 7 and dec.num_classes == 19


def test_state_dict_layout_and_param_counts():
    from nas_segm_amd.helpers.utils import compute_params
    from _util import build_product_net

    meta = load_json("nets_meta.json")
    for name, rec in meta.items():
        net = build_product_net(rec["kind"], rec["genotype"], rec["classes"], rec["dec_kwargs"], rec["seed"])
        assert set(net.state_dict()) == set(rec["checksums"]), name
        total, no_aux = compute_params(net)
        assert total == rec["n_params"]
        assert no_aux <= total
    w = meta["wacv_arch0"]
    assert w["n_params"] == 280147 and meta["wacv_arch1"]["n_params"] == 268235  # README.md:79


def test_widths_that_the_kernels_cannot_take_are_refused_at_construction():
    """the reference accepts agg_size=50; here that is a ValueError when the decoder is BUILT (not a
    RuntimeError inside every training step, which try_except would turn into a silent reward 0)"""
    from nas_segm_amd.nn.layer_factory import OPS
    from nas_segm_amd.nn.micro_decoders import TemplateDecoder

    genotype = load_json("nets_meta.json")["wacv_arch0"]["genotype"]
    with pytest.raises(ValueError) as e:
        TemplateDecoder([24, 32], 19, genotype, agg_size=50, repeats=1)
    assert "multiple of 4" in str(e.value)
    with pytest.raises(ValueError):
        OPS["sep_conv_3x3"](6, 6, 1, True)
    OPS["conv3x3"](8, 8, 1, True)  # fine


def test_product_has_no_cpu_fallback():
    from nas_segm_amd import NassegError
    from nas_segm_amd.nn.layer_factory import OPS

    mod = OPS["sep_conv_3x3"](8, 8, 1, True)
    with pytest.raises(NassegError):
        mod(torch.randn(1, 8, 5, 5))
    from nas_segm_amd.helpers.miou_utils import fast_cm

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            fast_cm(np.zeros(4, np.uint8), np.zeros(4, np.uint8), 2)


def test_product_never_imports_the_oracle():
    import os
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nas-segm-pytorch_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(dp, f)


def test_try_except_scores_runtime_errors_zero_only():
    from nas_segm_amd.helpers.utils import try_except

    @try_except
    def boom(kind):
        raise kind("x")

    assert boom(RuntimeError) == 0
    with pytest.raises(ValueError):
        boom(ValueError)


def test_optimiser_side_of_the_step_matches_the_reference_given_its_gradients():
    """tests/golden/engine_optim.npz: the raw gradients of two reference train_segmenter steps and
    the parameters / Polyak averages they led to.  Fed the same gradients, the product's
    post-backward half of the step (engine.trainer.finish_step: rank sync, per-sub-module norm
    clipping, SGD encoder / Adam decoder; then the Polyak update) must land on the same
    parameters to fp32 rounding - independent of how well conditioned the network's backward is."""
    from _util import build_product_net, load_npz, sub_dict
    from nas_segm_amd.engine import RankParallel
    from nas_segm_amd.engine.trainer import _polyak_update, _zero_grads, finish_step

    meta, npz = load_json("engine_optim_meta.json"), load_npz("engine_optim.npz")
    net = build_product_net(meta["kind"], meta["genotype"], meta["classes"], meta["dec_kwargs"], meta["seed"])
    init = sub_dict(npz, "init")
    for k, v in net.state_dict().items():
        assert torch.equal(v, init[k]), k  # same seeded initialisation as the reference's
    segmenter = RankParallel(net)
    e, d = meta["enc"], meta["dec"]
    optim_enc = torch.optim.SGD(net.encoder.parameters(), lr=e["lr"], momentum=e["momentum"],
                                weight_decay=e["weight_decay"])
    optim_dec = torch.optim.Adam(net.decoder.parameters(), lr=d["lr"], weight_decay=d["weight_decay"])
    avg_param = [p.data.clone() for p in segmenter.parameters()]
    groups = (list(net.encoder.parameters()), list(net.decoder.parameters()))
    for step in range(meta["steps"]):
        _zero_grads(segmenter, (optim_enc, optim_dec))
        grads = sub_dict(npz, "grad/{}".format(step))
        for k, p in net.named_parameters():
            p.grad = grads[k].clone()
        finish_step(segmenter, groups, optim_enc, optim_dec, meta["clip"], meta["clip"])
        _polyak_update(segmenter.parameters(), avg_param, meta["polyak_decay"])
    after, polyak = sub_dict(npz, "after"), sub_dict(npz, "polyak")
    worst, moved = 0.0, 0
    for (k, p), a in zip(net.named_parameters(), avg_param):
        for got, want in ((p.data, after[k]), (a, polyak[k])):
            err = float((got - want).abs().max())
            scale = float(want.abs().max()) + 1e-12
            worst = max(worst, err / scale)
            assert err <= 1e-6 * scale + 1e-9, (k, err, scale)
        moved += float((after[k] - init[k]).abs().max()) > 0
    assert worst < 1e-6
    assert moved > 0.8 * len(avg_param), moved  # (the record does move the parameters)


def test_install_dropin_registers_reference_module_names():
    import sys

    import nas_segm_amd

    saved = {k: sys.modules.get(k) for k in ("nn", "nn.layer_factory", "rl.genotypes", "helpers.miou_utils")}
    try:
        names = nas_segm_amd.install_dropin()
        assert "nn.micro_decoders" in names
        from nn.layer_factory import OPS  # noqa: F401  (the reference's import line)
        from helpers.miou_utils import compute_iu, compute_ius_accs, fast_cm  # noqa: F401
    finally:
        for k in list(sys.modules):
            if k in ("nn", "rl", "helpers", "engine") or k.split(".")[0] in ("nn", "rl", "helpers", "engine"):
                if k not in saved or saved[k] is None:
                    sys.modules.pop(k, None)


def test_install_dropin_can_also_stand_in_for_the_teacher_and_the_data_pipeline():
    """install_dropin(kd=True, data=True): the reference's import lines for the distillation teacher
    (src/main_search.py:456) and the loaders (:30) resolve here; in a subprocess - sys.modules stays clean."""
    import subprocess
    import sys

    code = ("import nas_segm_amd; names = nas_segm_amd.install_dropin(kd=True, data=True);"
            "from kd.rf_lw.model_lw_v2 import rf_lw152 as kd_model;"
            "from data.loaders import create_loaders;"
            "from data.datasets import PascalCustomDataset;"
            "import nas_segm_amd.kd.rf_lw as m; assert kd_model is m.rf_lw152;"
            "assert 'data.loaders' in names and 'kd.rf_lw.model_lw_v2' in names; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


@pytest.mark.parametrize("taps", [[1, 2], [1, 2, 4, 6]])
def test_encoder_groups_units_into_chains_without_crossing_a_skip_or_a_returned_map(taps, monkeypatch):
    """MobileNetV2.forward (host logic only: the fused runner is replaced by a recorder): a run of
    units ends at every block with a skip connection - before it (its input is added back) and after it
    (its sum is a tensor of its own) - and at every returned feature map; everything is run exactly
    once, in order."""
    import nas_segm_amd.nn.encoders as E
    from nas_segm_amd.nn.layer_factory import InvertedResidual

    enc = E.mbv2(pretrained=False, return_layers=taps)
    order = [enc.layer1.__class__.__name__ + ":stem"]
    names = {}
    for idx in range(enc.max_layer + 1):
        for b, unit in enumerate(getattr(enc, enc._stage_name(idx))):
            names[id(unit.conv)] = (idx, b, unit.use_res_connect)
    runs = []

    def fake_run_fused(mods, x, residual=None, relu_in=False):
        runs.append(["stem" if id(m) not in names else names[id(m)] for m in mods])
        return x

    monkeypatch.setattr(E, "run_fused", fake_run_fused)
    monkeypatch.setattr(InvertedResidual, "forward", lambda self, x: (runs.append([names[id(self.conv)]]), x)[1])
    outs = enc(torch.zeros(1, 3, 8, 8))
    assert len(outs) == len(taps)
    flat = [u for r in runs for u in r]
    want = ["stem"] + [names[id(u.conv)] for idx in range(enc.max_layer + 1)
                       for u in getattr(enc, enc._stage_name(idx))]
    assert flat == want  # every unit once, in order
    for r in runs:
        for pos, u in enumerate(r):
            if u != "stem" and u[2]:
                assert len(r) == 1, r  # a block with a skip connection runs alone
        stages_ended = [u[0] for u in r[:-1] if u != "stem" and
                        u[1] == len(getattr(enc, enc._stage_name(u[0]))) - 1 and u[0] in taps]
        assert not stages_ended, r  # no run continues past a returned map
    assert runs[0][0] == "stem" and len(runs[0]) == 3  # stem + stage 1 + the first block of stage 2


def _dry_run(monkeypatch, build, inputs):
    """Run forward + backward of a module tree on the CPU with every kernel launch replaced by a recorder
    (tensors stay uninitialised host memory; the workspace / plan queries are pure host functions): the
    HOST graph - which entry points run, in what order, with which shapes - without a GPU."""
    import nas_segm_amd.functional as F

    calls = []
    monkeypatch.setattr(F.lib, "call", lambda name, *a: (calls.append((name, a)), 0)[1])
    monkeypatch.setattr(F, "require_device", lambda *a: None)
    monkeypatch.setattr(F, "current_stream", lambda: 0)
    torch.manual_seed(0)
    net = build().train()
    xs = [torch.empty(*shape).contiguous(memory_format=torch.channels_last).requires_grad_(True) for shape in inputs]
    out = net(xs)
    out = out[0] if isinstance(out, tuple) else out
    n_fwd = len(calls)
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out.backward(torch.empty_like(out))
    # autograd's own accumulation of a gradient over several consumers is the only ATen add of such a backward
    _dry_run.accumulations = sum(1 for e in prof.events() if e.name in ("aten::add", "aten::add_"))
    return F, calls[:n_fwd], calls[n_fwd:], xs


def test_template_decoder_hands_pending_batchnorms_to_concat_reduce(monkeypatch):
    """TemplateDecoder asks the two ops of a block for PENDING outputs (functional.Pending: raw conv output +
    the last BatchNorm's statistics) when its aggregation op is ConcatReduce, which then runs as one node:
    no normalise pass (nasseg_affine_act) over an op's output, no statistics pass over the slab, one
    nasseg_cat_src_fwd / _bwd per input, and - backward - the producers' BatchNorm sums arrive with the
    gradient (functional._TAIL_ROWS, keyed by the gradient tensor ITSELF: this checks that autograd hands the
    very tensor object on), so no producer runs a reduction pass; nothing is left in the side table."""
    from nas_segm_amd.nn.micro_decoders import TemplateDecoder

    # two blocks, both [sep_conv_3x3, sep_conv_5x5, cat]: maps of two sizes meet at the smaller / the larger
    genotype = [[[0, 1, 1]], [[0, 1, 0, 0, 0], [1, 2, 0, 1, 0]]]
    F, fwd, bwd, xs = _dry_run(monkeypatch, lambda: TemplateDecoder([24, 32], 19, genotype, agg_size=32, repeats=1),
                               [(2, 24, 32, 64), (2, 32, 16, 32)])
    f_names, b_names = [n for n, _ in fwd], [n for n, _ in bwd]
    n_cat = f_names.count("nasseg_cat_src_fwd") // 2
    assert n_cat >= 2 and f_names.count("nasseg_cat_src_fwd") == 2 * n_cat
    assert "nasseg_bn_stats" not in f_names or f_names.count("nasseg_bn_stats") <= 1  # (pre_clf's slab only)
    # the only normalise passes are pre_clf's output and inputs Adapt's 1x1 convs needed materialised
    cells = sum(1 for n in f_names if n == "nasseg_sepconv_fwd" or n == "nasseg_dwconv")
    assert cells > 0
    assert f_names.count("nasseg_affine_act") <= 1 + f_names.count("nasseg_cat_src_fwd") // 2
    assert b_names.count("nasseg_cat_src_bwd") == 2 * n_cat
    pending_inputs = sum(1 for n, a in bwd if n == "nasseg_cat_src_bwd" and a[9] is not None)
    assert pending_inputs >= n_cat  # at least one pending producer per ConcatReduce here
    # every pending producer got its sums by the side: what still reduces is pre_clf's tail and materialised inputs
    reduces = b_names.count("nasseg_bn_bwd_reduce") + b_names.count("nasseg_bn_bwd_reduce_rows")
    assert reduces <= 1 + f_names.count("nasseg_affine_act")
    # (the rows that came by the side are summed by a launch of their own or - few rows - by the apply kernel itself)
    by_side = b_names.count("nasseg_rows_sum") + sum(
        1 for n, a in bwd if n == "nasseg_bn_bwd_apply_rows") - b_names.count("nasseg_bn_bwd_reduce_rows")
    assert by_side >= pending_inputs
    assert not F._TAIL_ROWS, "rows left behind: {}".format(len(F._TAIL_ROWS))
    assert all(x.grad is not None for x in xs)


def test_pending_tail_is_materialised_for_consumers_that_do_not_take_it(monkeypatch):
    """A Pending handed to anything but ConcatReduce's node costs exactly the pass it had deferred
    (functional.materialize -> nasseg_affine_act), and the no-concatenation form of ConcatReduce
    (NASSEG_SPLIT_CAT_MIN) and NASSEG_FUSE_CAT_REDUCE=0 both take that route."""
    import nas_segm_amd.nn.layer_factory as LF
    from nas_segm_amd.nn.layer_factory import AGG_OPS, OPS, run_op

    class Cell(torch.nn.Module):
        def __init__(self, agg):
            super(Cell, self).__init__()
            self.a, self.b = OPS["sep_conv_3x3"](32, 32, 1, True, 1), OPS["dil_conv_3x3"](32, 32, 1, True)
            self.agg = AGG_OPS[agg](32, 32, 32, True, 1, True)

        def forward(self, xs):
            return self.agg(run_op(self.a, xs[0], True), run_op(self.b, xs[1], True))

    shapes = [(2, 32, 16, 32), (2, 32, 16, 32)]
    F, fwd, bwd, _ = _dry_run(monkeypatch, lambda: Cell("cat"), shapes)
    assert [n for n, _ in fwd].count("nasseg_affine_act") == 0
    monkeypatch.setattr(F, "FUSE_CAT_REDUCE", False)
    F, fwd, bwd, _ = _dry_run(monkeypatch, lambda: Cell("cat"), shapes)
    assert [n for n, _ in fwd].count("nasseg_affine_act") == 2 and "nasseg_cat_src_fwd" not in [n for n, _ in fwd]
    monkeypatch.setattr(F, "FUSE_CAT_REDUCE", True)
    monkeypatch.setattr(LF, "_SPLIT_CAT_MIN", 0)
    F, fwd, bwd, _ = _dry_run(monkeypatch, lambda: Cell("cat"), shapes)
    assert [n for n, _ in fwd].count("nasseg_affine_act") == 2 and "nasseg_cat_src_fwd" not in [n for n, _ in fwd]
    # ParamSum takes pending operands too (round 4): one forward kernel applies both tails, one backward kernel
    # leaves both masked gradients with their producers' sums and the coefficient gradients' rows
    F, fwd, bwd, _ = _dry_run(monkeypatch, lambda: Cell("psum"), shapes)
    f_names, b_names = [n for n, _ in fwd], [n for n, _ in bwd]
    assert f_names.count("nasseg_affine_act") == 0 and f_names.count("nasseg_add_act2") == 1
    assert b_names.count("nasseg_psum_bwd") == 1 and b_names.count("nasseg_bn_bwd_reduce") == 0
    assert b_names.count("nasseg_bn_bwd_reduce_rows") == 0
    assert b_names.count("nasseg_axpby") == 0 and b_names.count("nasseg_colred") == 0
    assert not F._TAIL_ROWS
    monkeypatch.setattr(F, "FUSE_PENDING_PSUM", False)
    F, fwd, bwd, _ = _dry_run(monkeypatch, lambda: Cell("psum"), shapes)
    assert [n for n, _ in fwd].count("nasseg_affine_act") == 2 and "nasseg_psum_bwd" not in [n for n, _ in bwd]
    assert not F._TAIL_ROWS


def test_native_call_shim_agrees_with_ctypes():
    """The generated CPython shim (ffi_gen.py -> _nasseg_ffi) is the default call path once built: same symbols
    (bound by address from the ctypes handle), same results and error convention as ctypes; NASSEG_FFI=ctypes
    switches it off."""
    import subprocess
    import sys

    from nas_segm_amd._lib import NassegError, lib, parse_header

    lib.load()
    assert lib.ffi == "native", "build() did not produce the call shim"
    protos = parse_header()
    for name in protos:
        assert lib._fn[name] is not lib._ctypes_fn[name], name  # every prototype has a native entry
    for args in ((4, 128, 256, 64, 64, 1), (1, 7, 9, 20, 32, 0), (2, 512, 1024, 96, 16, 2)):
        assert lib._fn["nasseg_conv_fwd_stats_blocks"](*args) == lib._ctypes_fn["nasseg_conv_fwd_stats_blocks"](*args)
    assert lib._fn["nasseg_colred_workspace"](1, 1 << 33, 64) == lib._ctypes_fn["nasseg_colred_workspace"](1, 1 << 33, 64)
    # a refused call: negative status, message through nasseg_last_error() (bytes from either path)
    with pytest.raises(NassegError, match="rows_sum"):
        lib.call("nasseg_rows_sum", None, 0, 0, None, None)
    with pytest.raises(TypeError):
        lib._fn["nasseg_rows_sum"](None, 0, 0, None)  # arity is checked
    with pytest.raises(TypeError):
        lib._fn["nasseg_rows_sum"]("not an address", 1, 1, None, None)
    code = ("import nas_segm_amd; from nas_segm_amd._lib import lib; lib.load(); "
            "assert lib.ffi == 'ctypes'; print(lib.query('nasseg_cat_src_blocks', 4, 32, 64, 64))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, NASSEG_FFI="ctypes"),
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    assert int(out.stdout.strip()) == lib.query("nasseg_cat_src_blocks", 4, 32, 64, 64)


def test_clip_and_step_accepts_generators():
    """clip_and_step is a public helper: parameters handed as a generator - model.parameters() -
    are clipped AND stepped on torch's path too (an exhausted generator would skip the clipping silently)"""
    import torch
    from torch import nn

    from nas_segm_amd.engine.trainer_common import clip_and_step

    def run(as_generator):
        torch.manual_seed(0)
        m = nn.Linear(7, 5)
        o = torch.optim.SGD(m.parameters(), lr=0.5)
        for p in m.parameters():
            p.grad = torch.full_like(p, 3.0)
        clip_and_step([(m.parameters() if as_generator else list(m.parameters()), 0.1, o)])
        return [p.detach().clone() for p in m.parameters()], [p.grad.clone() for p in m.parameters()]

    (pg, gg), (pl, gl) = run(True), run(False)
    for a, b in zip(pg + gg, pl + gl):
        assert torch.equal(a, b)
    total = torch.sqrt(sum((g ** 2).sum() for g in gg))
    assert abs(float(total) - 0.1) < 1e-4  # (the gradients WERE clipped)


def test_skip_gradients_ride_in_the_first_convs_backward(monkeypatch):
    """InvertedResidual blocks with a skip (src/nn/layer_factory.py:276-321): the block's input receives the gradient
    through the block and the gradient of the skip.  With functional.FUSE_RES_GRAD the first conv's backward adds the
    skip's in its epilogue - the residual operand of the plain backward-data call (small maps), or dx_res of the
    one-kernel pointwise backward (large maps) - and autograd accumulates nothing; without it, one ATen add per skip."""
    from nas_segm_amd.nn.layer_factory import InvertedResidual
    from torch import nn

    class Blocks(nn.Module):
        def __init__(self):
            super(Blocks, self).__init__()
            self.a = InvertedResidual(24, 24, 1, 6)
            self.b = InvertedResidual(24, 24, 1, 6)

        def forward(self, xs):
            return self.b(self.a(xs[0]))

    for shape, one_kernel in (((2, 24, 20, 24), False), ((4, 24, 256, 512), True)):
        F, fwd, bwd, xs = _dry_run(monkeypatch, Blocks, [shape])
        fused = _dry_run.accumulations
        names = [n for n, _ in bwd]
        assert ("nasseg_conv_pw_bwd_bn" in names) == one_kernel, names
        if one_kernel:  # (dx_res: argument 26 of nasseg_conv_pw_bwd_bn, K -> 6K expansions only)
            with_skip = [a for n, a in bwd if n == "nasseg_conv_pw_bwd_bn" and a[21] == 24 and a[26]]
        else:           # (the residual operand of nasseg_conv_fwd: argument 11)
            with_skip = [a for n, a in bwd if n == "nasseg_conv_fwd" and a[11]]
        assert len(with_skip) == 2 and fused == 0, (shape, len(with_skip), fused)
        monkeypatch.setattr(F, "FUSE_RES_GRAD", False)
        F, fwd0, bwd0, xs0 = _dry_run(monkeypatch, Blocks, [shape])
        monkeypatch.setattr(F, "FUSE_RES_GRAD", True)
        assert _dry_run.accumulations == 2 and [n for n, _ in bwd0] == names


def test_gradient_junctions_replace_autograd_accumulation(monkeypatch):
    """Nodes with several consumers - a cell's input read by five ops, op outputs read by a sum and another op, the
    decoder maps read by several cells / blocks and collect_all (src/nn/micro_decoders.py:95-121,237-251,380-398) -
    are fanned out (functional.fan_out): backward adds their consumers' gradients in ONE nasseg_grad_junction launch
    per node, autograd accumulates nothing; a junction over a pending node hands the producer's BatchNorm-backward
    rows on, so that chain runs no reduction of its own.  NASSEG_JUNCTION=0 is the old graph."""
    from nas_segm_amd.nn.micro_decoders import MicroDecoder, TemplateDecoder

    rec = load_json("nets_meta.json")
    cvpr, wacv = rec["cvpr_arch0"], rec["wacv_arch0"]
    shapes4 = [(2, 24, 24, 24), (2, 32, 12, 12), (2, 96, 6, 6), (2, 320, 6, 6)]
    shapes2 = [(2, 24, 32, 64), (2, 32, 16, 32)]
    builders = {
        "cvpr": (lambda: MicroDecoder([24, 32, 96, 320], 21, cvpr["genotype"], agg_size=32, repeats=2), shapes4),
        "wacv": (lambda: TemplateDecoder([24, 32], 19, wacv["genotype"], agg_size=32, repeats=2), shapes2),
    }
    for name, (build, shapes) in builders.items():
        F, fwd, bwd, xs = _dry_run(monkeypatch, build, shapes)
        with_j = (_dry_run.accumulations, [n for n, _ in bwd])
        assert not F._TAIL_ROWS
        assert sum(x.grad is not None for x in xs) >= len(xs) - 1  # (the CVPR genotype leaves one encoder map unused)
        monkeypatch.setattr(F, "JUNCTION", False)
        F, fwd0, bwd0, xs0 = _dry_run(monkeypatch, build, shapes)
        without = (_dry_run.accumulations, [n for n, _ in bwd0])
        monkeypatch.setattr(F, "JUNCTION", True)
        assert [n for n, _ in fwd] == [n for n, _ in fwd0] or name == "cvpr"  # (cvpr: finished maps written by the junction)
        assert with_j[0] == 0, (name, with_j[0])
        assert without[0] >= 4 and "nasseg_grad_junction" not in without[1]
        n_j = with_j[1].count("nasseg_grad_junction")
        assert 1 <= n_j <= without[0], (name, n_j, without[0])
        if name == "cvpr":
            # junctions over pending nodes replaced their chains' reduction passes
            def reduces(names):
                return names.count("nasseg_bn_bwd_reduce") + names.count("nasseg_bn_bwd_reduce_rows")

            assert reduces(with_j[1]) < reduces(without[1])
            # every launch the old graph made that the new one does not is a reduction; what the new one adds are junctions
            assert len(with_j[1]) - n_j <= len(without[1])


def test_fanned_out_nodes_are_consumed_exactly_once():
    """functional.fan_out hands every consumer of a decoder node its own alias so that their gradients meet in one
    junction launch; an alias nobody reads would leave its share out of that sum silently, and a consumer the decoder
    did not count would read an alias that belongs to somebody else: both are loud now."""
    import torch

    from nas_segm_amd import functional as F
    from nas_segm_amd.nn.micro_decoders import _Handles

    t = torch.zeros(1, 4, 2, 2)
    h = _Handles([t], [[False, False, False]])
    assert h.take(0) is t and h.take(0) is t
    with pytest.raises(F.NassegError):
        h.check_all_taken()          # the third consumer never came
    assert h.take(0) is t
    h.check_all_taken()
    with pytest.raises(F.NassegError):
        h.take(0)                    # a fourth was never counted
    single = _Handles([t], [[False]])
    assert single.take(0) is t       # (a node with one consumer is not fanned out)
    single.check_all_taken()


def test_a_stale_stepper_is_recorded_again_by_the_cache():
    """engine/trainer._cached_stepper: a cached stepper whose capture no longer matches its optimisers (stale(): the
    hyper-parameters a capture with optimisers inside bakes in by value) is dropped and built again; a shape whose
    capture failed once stays on the host path"""
    from nas_segm_amd.engine.trainer import _cached_stepper

    class Stepper(object):
        def __init__(self):
            self.is_stale = False

        def stale(self):
            return self.is_stale

    class Owner(object):
        pass

    owner, built = Owner(), []

    def build():
        built.append(Stepper())
        return built[-1]

    a = _cached_stepper(owner, "_slot", ("base",), ("shape",), build)
    assert _cached_stepper(owner, "_slot", ("base",), ("shape",), build) is a and len(built) == 1
    a.is_stale = True
    b = _cached_stepper(owner, "_slot", ("base",), ("shape",), build)
    assert b is not a and len(built) == 2
    assert _cached_stepper(owner, "_slot", ("base",), ("shape",), build) is b

    def failing():
        raise RuntimeError("out of memory")

    assert _cached_stepper(owner, "_slot", ("base",), ("other",), failing) is None
    assert _cached_stepper(owner, "_slot", ("base",), ("other",), build) is None and len(built) == 2

