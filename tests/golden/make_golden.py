"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never has
it).  The reference is imported unmodified from /root/reference/src; its engine
functions hard-code ``.cuda()``, so ``Tensor.cuda`` / ``Module.cuda`` are made
identity for the duration of this script.  The Cython module
src/helpers/miou_utils.pyx does not compile under Cython 3 + NumPy 2 as is
(removed ``np.int_t`` / ``np.float_t`` aliases); a copy under /tmp gets a
type-alias-only edit (np.int_t -> np.int64_t, np.float_t -> np.float64_t,
np.int_ -> np.int64) and is built there - nothing of the reference is written
into the repository, only inputs and outputs (data) are.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np

np.int = int  # src/helpers/storage.py:20 uses the removed alias

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
TMP = "/tmp/nasseg_golden"

sys.path[:0] = [os.path.join(REF, "src"), REF]

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

torch.set_num_threads(8)


def build_cython():
    hdir = os.path.join(TMP, "helpers")
    os.makedirs(hdir, exist_ok=True)
    src = open(os.path.join(REF, "src/helpers/miou_utils.pyx")).read()
    src = src.replace("np.int_t", "np.int64_t").replace("np.float_t", "np.float64_t")
    src = src.replace("dtype=np.int_)", "dtype=np.int64)")
    open(os.path.join(hdir, "miou_utils.pyx"), "w").write(src)
    setup = (
        "from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy\n"
        "setup(ext_modules=cythonize([Extension('helpers.miou_utils', ['helpers/miou_utils.pyx'],"
        " include_dirs=[numpy.get_include()])], language_level=2))\n"
    )
    open(os.path.join(TMP, "setup.py"), "w").write(setup)
    subprocess.check_call([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=TMP)
    sys.path.insert(0, TMP)


def t2n(t):
    return t.detach().cpu().numpy()


class Store(object):
    def __init__(self):
        self.d = {}

    def put(self, key, value):
        if isinstance(value, torch.Tensor):
            value = t2n(value)
        self.d[key] = np.array(value, copy=True)  # never alias live module buffers

    def put_sd(self, prefix, sd):
        for k, v in sd.items():
            self.put(prefix + "/" + k, v)

    def save(self, name):
        path = os.path.join(OUT, name)
        np.savez_compressed(path, **self.d)
        print("wrote", path, "{:.1f} KB".format(os.path.getsize(path) / 1024.0), len(self.d), "arrays")


def randomize_bn(module, gen):
    for m in module.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)


def checksums(sd):
    """per-key (sum, abs-sum) in float64 - pins seeded initialisation without storing weights"""
    return {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}


# ---------------------------------------------------------------------------
# A. op registry
# ---------------------------------------------------------------------------
def gen_ops():
    from nn.layer_factory import AGG_OPS, OPS

    st = Store()
    cases = []
    gen = torch.Generator().manual_seed(1234)
    for name in sorted(OPS.keys()):
        for stride in (1, 2):
            C_in, C_out = 8, (8 if stride == 1 else 16)
            case = "{}__s{}".format(name, stride)
            torch.manual_seed(100 + len(cases))
            mod = OPS[name](C_in, C_out, stride, True, 2)
            randomize_bn(mod, gen)
            x = torch.randn(2, C_in, 13, 17, generator=gen)
            st.put(case + "/x", x)
            st.put_sd(case + "/sd", mod.state_dict())
            mod.eval()
            with torch.no_grad():
                st.put(case + "/y_eval", mod(x))
            mod.train()
            xg = x.clone().requires_grad_(True)
            y = mod(xg)
            g = torch.randn(y.shape, generator=gen)
            st.put(case + "/y_train", y)
            st.put(case + "/g", g)
            params = [(k, p) for k, p in mod.named_parameters()]
            if y.requires_grad:
                grads = torch.autograd.grad(y, [xg] + [p for _, p in params], g, allow_unused=True)
                st.put(case + "/dx", grads[0] if grads[0] is not None else torch.zeros_like(x))
                for (k, _), gr in zip(params, grads[1:]):
                    st.put(case + "/grad/" + k, gr)
            st.put_sd(case + "/sd_after", {k: v for k, v in mod.state_dict().items()
                                           if "running" in k or "num_batches" in k})
            cases.append({"case": case, "name": name, "stride": stride, "C_in": C_in,
                          "C_out": C_out, "repeats": 2, "kind": "op"})
    # aggregation ops: (shape of x, shape of y) exercising the tuple comparison of resize()
    shapes = [((13, 17), (7, 9)), ((7, 9), (13, 17)), ((13, 17), (13, 17)), ((10, 5), (9, 20))]
    for name in sorted(AGG_OPS.keys()):
        for larger in (True, False):
            for si, (s0, s1) in enumerate(shapes):
                C0, C1, Co = 8, 16, 16
                case = "agg_{}__l{}__{}".format(name, int(larger), si)
                torch.manual_seed(500 + len(cases))
                mod = AGG_OPS[name](C0, C1, Co, True, 2, larger)
                randomize_bn(mod, gen)
                with torch.no_grad():
                    for k, p in mod.named_parameters():
                        if k in ("a", "b"):
                            p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
                x = torch.randn(2, C0, *s0, generator=gen)
                y = torch.randn(2, C1, *s1, generator=gen)
                st.put(case + "/x", x)
                st.put(case + "/y", y)
                st.put_sd(case + "/sd", mod.state_dict())
                mod.eval()
                with torch.no_grad():
                    st.put(case + "/out_eval", mod(x, y))
                mod.train()
                xg, yg = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
                out = mod(xg, yg)
                g = torch.randn(out.shape, generator=gen)
                params = [(k, p) for k, p in mod.named_parameters()]
                grads = torch.autograd.grad(out, [xg, yg] + [p for _, p in params], g)
                st.put(case + "/out_train", out)
                st.put(case + "/g", g)
                st.put(case + "/dx", grads[0])
                st.put(case + "/dy", grads[1])
                for (k, _), gr in zip(params, grads[2:]):
                    st.put(case + "/grad/" + k, gr)
                st.put_sd(case + "/sd_after", {k: v for k, v in mod.state_dict().items()
                                               if "running" in k or "num_batches" in k})
                cases.append({"case": case, "name": name, "larger": larger, "C_in0": C0,
                              "C_in1": C1, "C_out": Co, "kind": "agg"})
    st.save("ops.npz")
    json.dump(cases, open(os.path.join(OUT, "ops_cases.json"), "w"), indent=1)


# ---------------------------------------------------------------------------
# B. whole networks
# ---------------------------------------------------------------------------
GENOTYPES = {
    # tests/test_inference.py:19-111 (published architectures)
    "cvpr_arch0": [[8, [0, 0, 5, 2], [0, 2, 8, 8], [0, 5, 1, 4]], [[3, 3], [3, 2], [3, 0]]],
    "cvpr_arch1": [[2, [1, 0, 3, 6], [0, 1, 2, 8], [2, 0, 6, 1]], [[2, 3], [3, 1], [4, 4]]],
    "cvpr_arch2": [[5, [0, 0, 4, 1], [3, 2, 0, 1], [5, 6, 5, 0]], [[1, 3], [4, 3], [2, 2]]],
    "wacv_arch0": [[[3, 0, 1], [4, 1, 1], [3, 1, 1]],
                   [[0, 1, 0, 0, 1], [2, 1, 2, 1, 0], [3, 1, 1, 1, 0], [1, 1, 2, 0, 0],
                    [3, 0, 2, 0, 0], [5, 3, 2, 1, 0], [0, 5, 0, 1, 0]]],
    "wacv_arch1": [[[1, 1, 0], [1, 3, 0], [3, 4, 0]],
                   [[1, 1, 0, 0, 0], [0, 1, 1, 1, 1], [3, 1, 2, 3, 0], [3, 0, 2, 2, 0],
                    [0, 1, 2, 0, 0], [2, 1, 1, 3, 0], [4, 0, 2, 2, 0]]],
}

NETS = [
    # name, kind, genotype, classes, decoder kwargs, input shape, seed, store full state_dict
    ("wacv_arch0", "template", "wacv_arch0", 19, dict(agg_size=64, repeats=2), (2, 3, 65, 97), 0, True),
    ("wacv_arch1", "template", "wacv_arch1", 19, dict(agg_size=64, repeats=2), (2, 3, 65, 97), 1, False),
    ("cvpr_arch0", "micro", "cvpr_arch0", 21, dict(agg_size=64, repeats=2), (2, 3, 97, 129), 2, False),
    ("cvpr_arch1_search", "micro", "cvpr_arch1", 21, dict(agg_size=48, repeats=1, aux_cell=True),
     (2, 3, 97, 129), 3, False),
    ("cvpr_arch2_depth", "micro", "cvpr_arch2", 1, dict(agg_size=64, repeats=2), (2, 3, 97, 129), 4, False),
]


def build_ref_net(kind, genotype, classes, dec_kwargs, seed):
    from functools import partial

    from nn.encoders import mbv2
    from nn.micro_decoders import MicroDecoder, TemplateDecoder

    torch.manual_seed(seed)
    config = GENOTYPES[genotype] if isinstance(genotype, str) else genotype
    if kind == "template":
        enc = mbv2(pretrained=False, return_layers=[1, 2])
        dec = TemplateDecoder(inp_sizes=enc.out_sizes, num_classes=classes, config=config,
                              **dec_kwargs)
    else:
        enc = mbv2(pretrained=False)
        dec = MicroDecoder(inp_sizes=list(enc.out_sizes), num_classes=classes,
                           config=config, **dec_kwargs)

    class EncoderDecoder(nn.Module):
        def __init__(self, encoder, decoder):
            super(EncoderDecoder, self).__init__()
            self.encoder = encoder
            self.decoder = decoder

        def forward(self, x):
            return self.decoder(self.encoder(x))

    return EncoderDecoder(enc, dec)


def make_labels(gen, B, H, W, classes):
    t = torch.randint(0, max(classes, 2), (B, H, W), generator=gen)
    t[:, H // 3: H // 3 + 5, :] = 255
    return t


def gen_nets():
    _gen_nets(NETS, "nets.npz", "nets_meta.json")


def sampled_nets():
    """BASELINE config 4: genotypes the reference controller sampled (controller.json, written by
    gen_controller under torch.manual_seed(9314)) built with the search-time defaults
    (src/utils/default_args.py:77-79: agg_size 48, sep_repeats 1, aux_cell) - plus the training
    record of the depth network (config 5; one output channel, so no segmentation loss: the
    gradients are taken under a fixed seeded cotangent instead)."""
    ctrl = json.load(open(os.path.join(OUT, "controller.json")))
    wacv = [s["config"] for s in ctrl["wacv"]["samples"]]
    cvpr = [s["config"] for s in ctrl["cvpr"]["samples"]]
    return [
        ("wacv_sampled0", "template", wacv[0], 19, dict(agg_size=48, repeats=1), (2, 3, 65, 97), 5, False),
        ("wacv_sampled1", "template", wacv[1], 19, dict(agg_size=48, repeats=1), (2, 3, 65, 97), 6, False),
        ("wacv_sampled3", "template", wacv[3], 19, dict(agg_size=48, repeats=1), (2, 3, 65, 97), 7, False),
        ("cvpr_sampled0", "micro", cvpr[0], 21, dict(agg_size=48, repeats=1, aux_cell=True),
         (2, 3, 97, 129), 8, False),
        ("cvpr_sampled3", "micro", cvpr[3], 21, dict(agg_size=48, repeats=1, aux_cell=True),
         (2, 3, 97, 129), 9, False),
        # (4 images: the cell's global-average-pool branch normalises a B x C x 1 x 1 map over B
        #  samples - with B = 2 its output is +-1 whatever the input)
        ("cvpr_arch2_depth_train", "micro", "cvpr_arch2", 1, dict(agg_size=64, repeats=2),
         (4, 3, 97, 129), 4, False),
    ]


def gen_nets_sampled():
    _gen_nets(sampled_nets(), "nets_sampled.npz", "nets_sampled_meta.json")


def _gen_nets(nets, npz_name, meta_name):
    st = Store()
    meta = {}
    for name, kind, geno, classes, kw, shape, seed, full in nets:
        net = build_ref_net(kind, geno, classes, kw, seed)
        gen = torch.Generator().manual_seed(9000 + seed)
        x = torch.randn(*shape, generator=gen)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        st.put(name + "/x", x)
        if full:
            st.put_sd(name + "/sd", sd0)
        net.eval()
        with torch.no_grad():
            out = net(x)
        aux = []
        if isinstance(out, tuple):
            out, aux = out
        st.put(name + "/logits_eval", out)
        for i, a in enumerate(aux):
            st.put(name + "/aux_eval/{}".format(i), a)
        rec = {"kind": kind, "genotype": GENOTYPES[geno] if isinstance(geno, str) else geno,
               "classes": classes, "dec_kwargs": kw,
               "shape": list(shape), "seed": seed, "full_sd": full, "checksums": checksums(sd0),
               "n_params": sum(p.numel() for p in net.parameters()), "n_aux": len(aux)}
        if classes > 1:
            # training-mode forward/backward with the loss of train_segmenter (trainer.py:233-255)
            net.train()
            target = make_labels(gen, shape[0], shape[2], shape[3], classes)
            st.put(name + "/target", target.to(torch.uint8))
            aux_weight = 0.15 if aux else -1
            output = net(x)
            aux_outs = []
            if isinstance(output, tuple):
                output, aux_outs = output
            tv = nn.functional.interpolate(target[:, None].float(), size=output.size()[2:],
                                           mode="nearest").long()[:, 0]
            crit = nn.NLLLoss(ignore_index=255)
            loss = crit(nn.LogSoftmax(dim=1)(output), tv)
            if aux_weight > 0:
                for a in aux_outs:
                    a = nn.Upsample(size=tv.size()[1:], mode="bilinear", align_corners=False)(a)
                    loss = loss + crit(nn.LogSoftmax(dim=1)(a), tv) * aux_weight
            net.zero_grad()
            loss.backward()
            st.put(name + "/logits_train", output)
            st.put(name + "/loss", loss)
            grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            rec["grad_checksums"] = checksums(grads)
            rec["aux_weight"] = aux_weight
            keys = sorted(grads.keys())
            pick = [keys[i] for i in np.linspace(0, len(keys) - 1, 14).astype(int)]
            for k in pick:
                st.put(name + "/grad/" + k, grads[k])
            bn_after = {k: v.clone() for k, v in net.state_dict().items() if "running_mean" in k}
            rec["bn_after_checksums"] = checksums(bn_after)
            # Conditioning of the whole-network gradients: tiny batches through ~100
            # train-mode BatchNorms and ReLUs make some of them change by percents when
            # the input moves by 1e-6 (fp32 rounding level).  Record, with the reference
            # itself, how far every stored gradient moves under two such perturbations;
            # parity tests use it as the tolerance floor.
            sens = {k: 0.0 for k in pick}
            mass_sens = {k: 0.0 for k in grads}
            bn_sens = {k: 0.0 for k in bn_after}
            logit_sens = 0.0
            for ps in (1, 2, 3, 4):
                net.load_state_dict(sd0)
                net.train()
                pg = torch.Generator().manual_seed(ps)
                xp = x + 1e-6 * torch.randn(x.shape, generator=pg)
                outp = net(xp)
                auxp = []
                if isinstance(outp, tuple):
                    outp, auxp = outp
                logit_sens = max(logit_sens, float((outp - output).abs().max()))
                lp = crit(nn.LogSoftmax(dim=1)(outp), tv)
                if aux_weight > 0:
                    for a in auxp:
                        a = nn.Upsample(size=tv.size()[1:], mode="bilinear", align_corners=False)(a)
                        lp = lp + crit(nn.LogSoftmax(dim=1)(a), tv) * aux_weight
                net.zero_grad()
                lp.backward()
                named = dict(net.named_parameters())
                for k in pick:
                    sens[k] = max(sens[k], float((named[k].grad - grads[k]).abs().max()))
                for k in grads:
                    mass_sens[k] = max(mass_sens[k], abs(float(named[k].grad.double().abs().sum())
                                                         - float(grads[k].double().abs().sum())))
                for k, v in net.state_dict().items():
                    if k in bn_sens:
                        bn_sens[k] = max(bn_sens[k], abs(float(v.double().abs().sum())
                                                         - float(bn_after[k].double().abs().sum())))
            rec["bn_after_mass_sensitivity"] = bn_sens
            rec["grad_sensitivity"] = sens
            # ... and how far the abs-sum ("mass") of EVERY gradient tensor moves: the tolerance
            # floor of the checksum comparison over all parameters
            rec["grad_mass_sensitivity"] = mass_sens
            rec["train_logits_sensitivity"] = logit_sens
        elif name.endswith("_train"):
            # one output channel (depth): train-mode forward and the gradients of
            # sum(output * g) for a fixed cotangent g
            net.train()
            output = net(x)
            if isinstance(output, tuple):
                output = output[0]
            g = torch.randn(output.shape, generator=gen)
            net.zero_grad()
            output.backward(g)
            st.put(name + "/logits_train", output)
            st.put(name + "/g", g)
            grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            rec["grad_checksums"] = checksums(grads)
            keys = sorted(grads.keys())
            pick = [keys[i] for i in np.linspace(0, len(keys) - 1, 14).astype(int)]
            for k in pick:
                st.put(name + "/grad/" + k, grads[k])
            sens = {k: 0.0 for k in grads}
            logit_sens = 0.0
            for ps in (1, 2, 3, 4):
                net.load_state_dict(sd0)
                net.train()
                pg = torch.Generator().manual_seed(ps)
                outp = net(x + 1e-6 * torch.randn(x.shape, generator=pg))
                if isinstance(outp, tuple):
                    outp = outp[0]
                logit_sens = max(logit_sens, float((outp - output).abs().max()))
                net.zero_grad()
                outp.backward(g)
                named = dict(net.named_parameters())
                for k in grads:
                    sens[k] = max(sens[k], float((named[k].grad - grads[k]).abs().max()))
            rec["grad_sensitivity"] = {k: sens[k] for k in pick}
            rec["grad_mass_sensitivity"] = {k: float(v) * grads[k].numel() for k, v in sens.items()}
            rec["train_logits_sensitivity"] = logit_sens
            # How the fp32 REFERENCE responds when nothing but its input image is rounded to
            # bfloat16 (8 bits of mantissa): with batch statistics this randomly initialised
            # network amplifies a relative perturbation ~200x, so a run that stores EVERY
            # activation in bf16 cannot be expected closer to this record than that response.
            net.load_state_dict(sd0)
            net.train()
            outp = net(x.to(torch.bfloat16).float())
            outp = outp[0] if isinstance(outp, tuple) else outp
            net.zero_grad()
            outp.backward(g)
            named = dict(net.named_parameters())

            def _cos(a, b):
                a, b = a.double().reshape(-1), b.double().reshape(-1)
                return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))

            rec["bf16_input_response"] = {
                "logits_rel_l2": float((outp - output).norm() / output.norm()),
                "grad_cos": {k: _cos(named[k].grad, grads[k]) for k in pick}}
            # The well-conditioned counterpart: the same gradients with every BatchNorm on its
            # running statistics (the engine's freeze_bn mode, src/engine/trainer.py:124-127) -
            # here a bf16 run must track the record closely.
            net.load_state_dict(sd0)
            net.train()
            for m in net.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
            outf = net(x)
            outf = outf[0] if isinstance(outf, tuple) else outf
            net.zero_grad()
            outf.backward(g)
            fgrads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            st.put(name + "/logits_frozen", outf)
            rec["frozen_grad_checksums"] = checksums(fgrads)
            for k in pick:
                st.put(name + "/frozen_grad/" + k, fgrads[k])
        meta[name] = rec
    st.save(npz_name)
    json.dump(meta, open(os.path.join(OUT, meta_name), "w"))


# ---------------------------------------------------------------------------
# C. mean-IoU / reward
# ---------------------------------------------------------------------------
class FakeDataset(object):
    def set_stage(self, stage):
        self.stage = stage


class FakeLoader(object):
    def __init__(self, batches):
        self.batches = batches
        self.dataset = FakeDataset()
        self.batch_sampler = type("BS", (), {"batch_size": 1})()

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


class FixedLogits(nn.Module):
    """a 'segmenter' whose output for batch i is a stored logits tensor"""

    def __init__(self, logits):
        super(FixedLogits, self).__init__()
        self.logits = logits
        self.i = 0
        self.dummy = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        out = self.logits[self.i]
        self.i += 1
        return out


def gen_miou():
    from helpers.miou_utils import compute_iu, compute_ius_accs, fast_cm

    import engine.inference as ref_inf

    st = Store()
    rng = np.random.RandomState(7)
    cases = []
    for ci, (n_cls, n_px) in enumerate([(21, 50000), (19, 20011), (5, 1), (11, 4096), (3, 0)]):
        gt = rng.randint(0, n_cls, size=n_px).astype(np.uint8)
        pr = rng.randint(0, n_cls, size=n_px).astype(np.uint8)
        if n_px > 100:
            # make two classes absent from gt and one absent from both
            gt[gt == 2] = 1
            pr[pr == 4] = 3
            gt[gt == 4] = 3
        cm = fast_cm(pr, gt, n_cls)
        iu = compute_iu(cm)
        iu2, npx, acc = compute_ius_accs(cm)
        assert np.array_equal(iu, iu2)
        case = "cm{}".format(ci)
        st.put(case + "/preds", pr)
        st.put(case + "/gt", gt)
        st.put(case + "/cm", cm)
        st.put(case + "/iu", iu)
        st.put(case + "/n_pixels", npx)
        st.put(case + "/accs", acc)
        cases.append({"case": case, "n_classes": n_cls})
    # validate(): logits at 1/4 resolution, labels at full resolution with 255 / out-of-range ids
    gen = torch.Generator().manual_seed(77)
    for vi, (n_cls, omit) in enumerate([(21, [0]), (19, [])]):
        logits = [torch.randn(2, n_cls, 17, 23, generator=gen) * 3 for _ in range(3)]
        masks = []
        for _ in range(3):
            m = torch.randint(0, n_cls - 2, (2, 65, 89), generator=gen)  # top classes absent
            m[:, 10:14, :] = 255
            m[:, :, 40] = n_cls + 3
            masks.append(m)
        loader = FakeLoader([{"image": torch.zeros(2, 3, 65, 89), "mask": mk} for mk in masks])
        reward = ref_inf.validate(FixedLogits(logits), loader, 0, 0, num_classes=n_cls,
                                  print_every=100, omit_classes=omit)
        case = "val{}".format(vi)
        for i in range(3):
            st.put("{}/logits/{}".format(case, i), logits[i])
            st.put("{}/mask/{}".format(case, i), masks[i].to(torch.uint8))
        st.put(case + "/reward", np.float64(reward))
        cases.append({"case": case, "n_classes": n_cls, "omit": omit, "n_batches": 3})
    st.save("miou.npz")
    json.dump(cases, open(os.path.join(OUT, "miou_cases.json"), "w"), indent=1)


# ---------------------------------------------------------------------------
# D. engine: reference train_segmenter / populate_task0 / train_task0 / validate
# ---------------------------------------------------------------------------
class _Seg(nn.Module):  # src/main_search.py:411-420
    def __init__(self, encoder, decoder):
        super(_Seg, self).__init__()
        self.encoder, self.decoder = encoder, decoder

    def forward(self, x):
        return self.decoder(self.encoder(x))


ENGINE_NETS = [
    ("wacv_arch0", "template", "wacv_arch0", 19, dict(agg_size=64, repeats=2), -1),
    ("cvpr_arch1_search", "micro", "cvpr_arch1", 21, dict(agg_size=48, repeats=1, aux_cell=True), 0.15),
]


def _perturbed(batches, noise_seed):
    if not noise_seed:
        return batches
    pg = torch.Generator().manual_seed(noise_seed)
    return [{"image": b["image"] + 1e-6 * torch.randn(b["image"].shape, generator=pg), "mask": b["mask"]}
            for b in batches]


def gen_engine():
    """train_segmenter / validate / populate_task0 / train_task0 of the reference on seeded
    batches; run nine times - as is (the record) and with the images perturbed by 1e-6 (eight
    seeds): how far the REFERENCE's own losses, parameters and reward move is stored next to every value
    (``*_sensitivity``) and is the floor of the parity tolerances (whole-network training through
    ~100 train-mode BatchNorms, ReLUs and max-pools is ill-conditioned: gradients move by ~1 %
    at fp32 rounding level, whatever the crop size)."""
    import engine.inference as ref_inf
    import engine.trainer as ref_tr
    from utils.solvers import create_optimisers

    st = Store()
    meta = {}
    crit = nn.NLLLoss(ignore_index=255)
    for name, kind, geno, classes, kw, aux_weight in ENGINE_NETS:
        gen = torch.Generator().manual_seed(4242)
        H, W = (65, 97) if kind == "template" else (97, 129)
        batches, vbatches = [], []
        for _ in range(2):
            img = torch.randn(2, 3, H, W, generator=gen)
            batches.append({"image": img, "mask": make_labels(gen, 2, H, W, classes).to(torch.uint8)})
        for _ in range(2):
            img = torch.randn(2, 3, H, W, generator=gen)
            vbatches.append({"image": img, "mask": make_labels(gen, 2, H, W, classes - 3).to(torch.uint8)})
        for tag, bs in (("train", batches), ("val", vbatches)):
            for i, b in enumerate(bs):
                st.put("{}/{}/image/{}".format(name, tag, i), b["image"])
                st.put("{}/{}/mask/{}".format(name, tag, i), b["mask"])

        def run(noise_seed):
            out = {}
            net = build_ref_net(kind, geno, classes, kw, seed=11)
            segmenter = nn.DataParallel(_Seg(net.encoder, net.decoder))
            out["init_checksums"] = checksums(segmenter.module.state_dict())
            sd_init = {k: v.clone() for k, v in segmenter.module.state_dict().items()}
            tb, vb = _perturbed(batches, noise_seed), _perturbed(vbatches, noise_seed)
            # --- task1: end-to-end, SGD encoder / Adam decoder (default_args.py:57-66) ---
            optim_enc, optim_dec = create_optimisers(
                "sgd", "adam", 1e-3, 3e-3, 0.9, 0.9, 1e-5, 1e-5,
                segmenter.module.encoder.parameters(), segmenter.module.decoder.parameters())
            losses = []

            def rec_crit(inp, tgt, _l=losses):
                v = crit(inp, tgt)
                _l.append(float(v))
                return v

            avg_param = [p.data.clone() for p in segmenter.parameters()]
            ret = ref_tr.train_segmenter(segmenter, FakeLoader(tb), optim_enc, optim_dec, 0, rec_crit,
                                         False, 3.0, 3.0, True, print_every=100, aux_weight=aux_weight,
                                         avg_param=avg_param, polyak_decay=0.99)
            assert ret is None, "reference train_segmenter failed"
            sd1 = {k: v.clone() for k, v in segmenter.module.state_dict().items()}
            out["task1_delta_mass"] = {k: float((sd1[k] - sd_init[k]).double().abs().sum())
                                       for k, _ in segmenter.module.named_parameters()}
            out["numel"] = {k: p.numel() for k, p in segmenter.module.named_parameters()}
            out["task1_crit_values"] = losses[:]
            out["task1_checksums"] = checksums(sd1)
            out["task1_polyak_checksums"] = checksums({str(i): a for i, a in enumerate(avg_param)})
            out["sd1"] = sd1
            # --- validation reward of the trained candidate ---
            reward = ref_inf.validate(segmenter, FakeLoader(vb), 0, 0, num_classes=classes,
                                      print_every=100, omit_classes=[0])
            out["val_reward"] = float(reward)
            # --- task0: cache encoder features, decoder-only epoch ---
            loader1 = FakeLoader([{"image": b["image"][i:i + 1], "mask": b["mask"][i:i + 1]}
                                  for b in tb for i in range(2)])
            Xy = ref_tr.populate_task0(segmenter, loader1, None, 4, do_kd=False)
            assert not isinstance(Xy, int), "reference populate_task0 failed"
            out["task0_cache_checksums"] = checksums({str(k): v for k, v in Xy.items() if k != "out_size"})
            out["task0_out_size"] = [int(s) for s in Xy["out_size"]]
            _, optim_dec0 = create_optimisers(
                "sgd", "adam", 1e-3, 3e-3, 0.9, 0.9, 1e-5, 1e-5,
                segmenter.module.encoder.parameters(), segmenter.module.decoder.parameters())
            np.random.seed(123)
            losses0 = []

            def rec_crit0(inp, tgt, _l=losses0):
                v = crit(inp, tgt)
                _l.append(float(v))
                return v

            ret = ref_tr.train_task0(Xy, segmenter, optim_dec0, 0, rec_crit0, None, 2, False, False, 0.0,
                                     3.0, False, aux_weight=max(aux_weight, 0))
            assert ret is None, "reference train_task0 failed"
            out["task0_crit_values"] = losses0[:]
            out["task0_checksums"] = checksums(segmenter.module.decoder.state_dict())
            return out

        base = run(0)
        sd1 = base.pop("sd1")
        rec = {"kind": kind, "genotype": GENOTYPES[geno], "classes": classes, "dec_kwargs": kw,
               "seed": 11, "aux_weight": aux_weight}
        rec.update(base)
        st.put(name + "/task1/conv_clf.weight", sd1["decoder.conv_clf.weight"])
        st.put(name + "/task1/layer1.0.weight", sd1["encoder.layer1.0.weight"])
        sens = {"task1_crit": [0.0] * len(base["task1_crit_values"]),
                "task0_crit": [0.0] * len(base["task0_crit_values"]), "val_reward": 0.0,
                "task1_mass": {k: 0.0 for k in base["task1_checksums"]},
                "task1_polyak_mass": {k: 0.0 for k in base["task1_polyak_checksums"]},
                "task0_cache_mass": {k: 0.0 for k in base["task0_cache_checksums"]},
                "task0_mass": {k: 0.0 for k in base["task0_checksums"]}}
        for ns in range(1, 9):
            other = run(ns)
            for key in ("task1_crit", "task0_crit"):
                sens[key] = [max(s, abs(a - b)) for s, a, b in
                             zip(sens[key], other[key + "_values"], base[key + "_values"])]
            sens["val_reward"] = max(sens["val_reward"], abs(other["val_reward"] - base["val_reward"]))
            for key, ck in (("task1_mass", "task1_checksums"), ("task1_polyak_mass", "task1_polyak_checksums"),
                            ("task0_cache_mass", "task0_cache_checksums"), ("task0_mass", "task0_checksums")):
                for k in sens[key]:
                    sens[key][k] = max(sens[key][k], abs(other[ck][k][1] - base[ck][k][1]))
        rec["sensitivity"] = sens
        meta[name] = rec
    st.save("engine.npz")
    json.dump(meta, open(os.path.join(OUT, "engine_meta.json"), "w"))


def gen_engine_kd():
    """Knowledge distillation through the task0 path (src/engine/trainer.py:53-61,147-149): the
    teacher's logits are cached next to the encoder features (bilinear to the first feature map's
    size) and ``kd_coeff * kd_crit(output, kd_y)`` joins the decoder-only loss.  The teacher here is
    a small seeded conv net - the engine only ever calls ``kd_net(image)``; the reference's own
    teacher (src/kd/rf_lw, a ResNet-152 RefineNet-LW) needs a downloaded checkpoint.  Run five
    times (as is + four 1e-6 input perturbations) like gen_engine."""
    import engine.trainer as ref_tr
    from utils.solvers import create_optimisers

    st = Store()
    name, kind, geno, classes, kw, aux_weight = ENGINE_NETS[1]
    gen = torch.Generator().manual_seed(777)
    H, W = 97, 129
    batches = []
    for i in range(4):
        img = torch.randn(1, 3, H, W, generator=gen)
        batches.append({"image": img, "mask": make_labels(gen, 1, H, W, classes).to(torch.uint8)})
        st.put("image/{}".format(i), img)
        st.put("mask/{}".format(i), batches[-1]["mask"])
    torch.manual_seed(31)
    teacher = nn.Sequential(nn.Conv2d(3, 16, 3, stride=2, padding=1), nn.ReLU(),
                            nn.Conv2d(16, classes, 3, stride=4, padding=1)).eval()
    for k, v in teacher.state_dict().items():
        st.put("teacher/" + k, v)
    crit = nn.NLLLoss(ignore_index=255)
    kd_coeff = 0.5

    def run(noise_seed):
        out = {}
        net = build_ref_net(kind, geno, classes, kw, seed=11)
        segmenter = nn.DataParallel(_Seg(net.encoder, net.decoder))
        Xy = ref_tr.populate_task0(segmenter, FakeLoader(_perturbed(batches, noise_seed)), teacher, 4, do_kd=True)
        assert not isinstance(Xy, int), "reference populate_task0 failed"
        out["kd_y_checksum"] = checksums({"kd_y": Xy["kd_y"]})["kd_y"]
        out["kd_y_shape"] = list(Xy["kd_y"].shape)
        _, optim_dec0 = create_optimisers(
            "sgd", "adam", 1e-3, 3e-3, 0.9, 0.9, 1e-5, 1e-5,
            segmenter.module.encoder.parameters(), segmenter.module.decoder.parameters())
        np.random.seed(321)
        seg_vals, kd_vals = [], []

        def rec_crit(inp, tgt):
            v = crit(inp, tgt)
            seg_vals.append(float(v))
            return v

        def rec_kd(inp, tgt):
            v = nn.functional.mse_loss(inp, tgt)
            kd_vals.append(float(v))
            return v

        ret = ref_tr.train_task0(Xy, segmenter, optim_dec0, 0, rec_crit, rec_kd, 2, False, True, kd_coeff,
                                 3.0, False, aux_weight=max(aux_weight, 0))
        assert ret is None, "reference train_task0 failed"
        out["crit_values"], out["kd_values"] = seg_vals, kd_vals
        out["checksums"] = checksums(segmenter.module.decoder.state_dict())
        return out

    base = run(0)
    sens = {"crit": [0.0] * len(base["crit_values"]), "kd": [0.0] * len(base["kd_values"]), "kd_y": 0.0,
            "mass": {k: 0.0 for k in base["checksums"]}}
    for ns in range(1, 5):
        other = run(ns)
        sens["crit"] = [max(s, abs(a - b)) for s, a, b in zip(sens["crit"], other["crit_values"], base["crit_values"])]
        sens["kd"] = [max(s, abs(a - b)) for s, a, b in zip(sens["kd"], other["kd_values"], base["kd_values"])]
        sens["kd_y"] = max(sens["kd_y"], abs(other["kd_y_checksum"][1] - base["kd_y_checksum"][1]))
        for k in sens["mass"]:
            sens["mass"][k] = max(sens["mass"][k], abs(other["checksums"][k][1] - base["checksums"][k][1]))
    rec = {"net": name, "kind": kind, "genotype": GENOTYPES[geno], "classes": classes, "dec_kwargs": kw, "seed": 11,
           "aux_weight": aux_weight, "kd_coeff": kd_coeff, "sensitivity": sens}
    rec.update(base)
    st.save("engine_kd.npz")
    json.dump(rec, open(os.path.join(OUT, "engine_kd_meta.json"), "w"))


def gen_engine_optim():
    """The optimiser side of train_segmenter in isolation: the gradients the reference's backward
    left in ``param.grad`` at every step (before clipping) are recorded together with the
    parameters / Polyak averages after the step.  Given the SAME gradients the rest of the step -
    per-sub-module clip_grad_norm_, SGD(momentum, weight decay) on the encoder, Adam(weight decay)
    on the decoder, Polyak averaging (src/engine/trainer.py:255-273) - is deterministic, so the
    product's post-step parameters are held to fp32 rounding, not to the conditioning of the
    network's backward."""
    import engine.trainer as ref_tr
    from utils.solvers import create_optimisers

    st = Store()
    meta = {}
    crit = nn.NLLLoss(ignore_index=255)
    name, kind, geno, classes, kw, aux_weight = ENGINE_NETS[0]
    kw = dict(agg_size=48, repeats=1)  # (search-time decoder size: a third of the parameters)
    gen = torch.Generator().manual_seed(555)
    batches = []
    for _ in range(2):
        img = torch.randn(2, 3, 65, 97, generator=gen)
        batches.append({"image": img, "mask": make_labels(gen, 2, 65, 97, classes).to(torch.uint8)})
    net = build_ref_net(kind, geno, classes, kw, seed=17)
    segmenter = nn.DataParallel(_Seg(net.encoder, net.decoder))
    st.put_sd("init", segmenter.module.state_dict())
    optim_enc, optim_dec = create_optimisers(
        "sgd", "adam", 1e-3, 3e-3, 0.9, 0.9, 1e-5, 1e-5,
        segmenter.module.encoder.parameters(), segmenter.module.decoder.parameters())
    avg_param = [p.data.clone() for p in segmenter.parameters()]
    names = [k for k, _ in segmenter.module.named_parameters()]
    step = [0]
    clip = nn.utils.clip_grad_norm_

    def recording_clip(parameters, max_norm, *a, **k):
        # called once per sub-module per step, encoder first (trainer.py:258-265): the
        # gradients are still the raw ones when the encoder's call arrives
        if step[0] % 2 == 0:
            for kname, p in segmenter.module.named_parameters():
                st.put("grad/{}/{}".format(step[0] // 2, kname), p.grad)
        step[0] += 1
        return clip(parameters, max_norm, *a, **k)

    nn.utils.clip_grad_norm_ = recording_clip
    try:
        for i, b in enumerate(batches):
            ret = ref_tr.train_segmenter(segmenter, FakeLoader([b]), optim_enc, optim_dec, 0, crit,
                                         False, 3.0, 3.0, True, print_every=100, aux_weight=aux_weight,
                                         avg_param=avg_param, polyak_decay=0.99)
            assert ret is None, "reference train_segmenter failed"
            if i + 1 < len(batches):
                continue  # (the last step's result depends on every earlier one)
            for kname, p in segmenter.module.named_parameters():
                st.put("after/{}".format(kname), p.data)
            for kname, a in zip(names, avg_param):
                st.put("polyak/{}".format(kname), a)
    finally:
        nn.utils.clip_grad_norm_ = clip
    assert step[0] == 2 * len(batches), step
    meta = {"kind": kind, "genotype": GENOTYPES[geno], "classes": classes, "dec_kwargs": kw, "seed": 17,
            "steps": len(batches), "enc": {"lr": 1e-3, "momentum": 0.9, "weight_decay": 1e-5},
            "dec": {"lr": 3e-3, "weight_decay": 1e-5}, "clip": 3.0, "polyak_decay": 0.99}
    st.save("engine_optim.npz")
    json.dump(meta, open(os.path.join(OUT, "engine_optim_meta.json"), "w"))


# ---------------------------------------------------------------------------
# E. controller -> decoder contract
# ---------------------------------------------------------------------------
def gen_controller():
    from rl.agent import create_agent

    out = {}
    for version, kw in [("cvpr", dict(num_ops=11, cell_num_layers=4)),
                        ("wacv", dict(num_ops=6, cell_num_layers=7))]:
        torch.manual_seed(9314)
        agent = create_agent(enc_num_layers=4 if version == "cvpr" else 2, num_agg_ops=2,
                             lstm_hidden_size=100, lstm_num_layers=2, dec_num_cells=3,
                             cell_max_repeat=4, cell_max_stride=2, ctrl_lr=1e-4,
                             ctrl_baseline_decay=0.95, ctrl_agent="ppo", ctrl_version=version, **kw)
        samples, raw = [], []
        for _ in range(6):
            config, entropy, log_prob = agent.controller.sample()
            raw.append((config, entropy, log_prob))
            action = agent.controller.config2action(config)
            samples.append({"config": config, "action": [int(a) for a in action],
                            "entropy": float(entropy), "log_prob": float(log_prob)})
        # the search loop's hand-over to the controller: the sampled records go through the
        # reference's train_agent (src/rl/agent.py:73-77 -> PPO.update ->
        # RolloutStorage.insert, helpers/storage.py:26-34) in sampling order with rewards
        # 0.01, 0.02, ...; what the rollout buffer holds afterwards is the contract a driver
        # that evaluates several candidates per iteration must reproduce
        from rl.agent import train_agent

        for k, (smp, (cfg, ent, lp)) in enumerate(zip(samples, raw)):
            train_agent(agent, (cfg, 0.01 * (k + 1), ent, lp))
        ro = agent.rollouts
        n = len(samples)
        out[version] = {"action_size": int(agent.controller.action_size()), "samples": samples,
                        "ppo_rollout": {"num_steps": int(ro.num_steps), "step": int(ro.step),
                                        "actions": ro.actions[:n].astype(int).tolist(),
                                        "rewards": ro.rewards[:n, 0].tolist(),
                                        "log_probs": ro.action_log_probs[:n, 0].tolist(),
                                        "baseline": float(agent.baseline)}}
    json.dump(out, open(os.path.join(OUT, "controller.json"), "w"), indent=1)
    print("wrote controller.json")


# ---------------------------------------------------------------------------
# H. the distillation teacher (src/kd/rf_lw/model_lw_v2.py): seeded random weights (no checkpoint can be
#    downloaded), BatchNorm parameters and buffers randomised, eval forward at 2x3x97x129
# ---------------------------------------------------------------------------
TEACHER_SEED, TEACHER_BN_SEED, TEACHER_X_SEED = 123, 5, 6


def gen_teacher():
    from kd.rf_lw.model_lw_v2 import rf_lw152

    torch.manual_seed(TEACHER_SEED)
    net = rf_lw152(pretrained=False, num_classes=21)
    randomize_bn(net, torch.Generator().manual_seed(TEACHER_BN_SEED))
    net.eval()
    x = torch.randn(2, 3, 97, 129, generator=torch.Generator().manual_seed(TEACHER_X_SEED))
    taps = {}
    hooks = [getattr(net, "layer{}".format(i)).register_forward_hook(
        lambda m, inp, out, i=i: taps.__setitem__("l{}".format(i), out.detach().clone())) for i in (1, 2, 3, 4)]
    with torch.no_grad():
        logits = net(x)
    for h in hooks:
        h.remove()
    st = Store()
    st.put("x", x)
    st.put("logits", logits)
    st.put("l1_sample", taps["l1"][:, ::32, ::5, ::7])  # (thinned: the full maps are megabytes)
    st.put("l4_sample", taps["l4"][:, ::128])
    st.save("teacher.npz")
    meta = {"seed": TEACHER_SEED, "bn_seed": TEACHER_BN_SEED, "x_seed": TEACHER_X_SEED, "num_classes": 21,
            "n_params": sum(p.numel() for p in net.parameters()),
            "tap_stats": {k: [float(v.double().mean()), float(v.double().abs().mean()), list(v.shape)]
                          for k, v in taps.items()},
            "checksums": checksums(net.state_dict())}
    json.dump(meta, open(os.path.join(OUT, "teacher_meta.json"), "w"))
    print("teacher: logits", tuple(logits.shape), "max |logit|", float(logits.abs().max()), "params", meta["n_params"])


# ---------------------------------------------------------------------------
# I. data pipeline (src/data/datasets.py): the transforms that do NOT call OpenCV, and the dataset class,
#    run by the reference itself.  cv2 is not in this image: an EMPTY module of that name lets
#    src/data/datasets.py import; no function of it exists, so nothing recorded here can depend on it
#    (ResizeScale / RandomMirror / the resizing branch of ResizeShorter are therefore not recorded).
# ---------------------------------------------------------------------------
def gen_data():
    import tempfile
    import types

    from PIL import Image

    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    from data import datasets as rds

    class Seq(object):  # (torchvision.transforms.Compose is absent too; the dataset only calls its transform)
        def __init__(self, ts):
            self.transforms = list(ts)

        def __call__(self, sample):
            for t in self.transforms:
                sample = t(sample)
            return sample

    rng = np.random.RandomState(77)
    st = Store()
    sizes = [(37, 53), (64, 48), (91, 90)]
    cases = []
    for i, (h, w) in enumerate(sizes):
        img = (rng.rand(h, w, 3) * 255).astype(np.uint8)
        msk = (rng.rand(h, w) * 21).astype(np.uint8)
        st.put("in{}/image".format(i), img)
        st.put("in{}/mask".format(i), msk)
        sample = {"image": img, "mask": msk}
        out = rds.Pad(70, [10, 20, 30], 255)(sample)
        st.put("pad{}/image".format(i), out["image"]); st.put("pad{}/mask".format(i), out["mask"])
        out = rds.CentralCrop(33)(sample)
        st.put("ccrop{}/image".format(i), out["image"]); st.put("ccrop{}/mask".format(i), out["mask"])
        np.random.seed(100 + i)
        out = rds.RandomCrop(41)(sample)
        st.put("rcrop{}/image".format(i), out["image"]); st.put("rcrop{}/mask".format(i), out["mask"])
        out = rds.ResizeShorter(min(h, w))(sample)  # (shorter side already there: the no-resize branch)
        st.put("rshort{}/image".format(i), out["image"])
        norm = rds.Normalise(1.0 / 255, np.array([0.485, 0.456, 0.406]).reshape((1, 1, 3)),
                             np.array([0.229, 0.224, 0.225]).reshape((1, 1, 3)))
        out = rds.ToTensor()(norm(sample))
        st.put("norm{}/image".format(i), out["image"]); st.put("norm{}/mask".format(i), out["mask"])
        cases.append({"h": h, "w": w})
    # the dataset class on files: two-column list, single-column list, a grey-scale image, both stages, set_config
    tmp = tempfile.mkdtemp(prefix="nasseg_data_")
    names = []
    for i, (h, w) in enumerate(sizes):
        img = (rng.rand(h, w, 3) * 255).astype(np.uint8) if i != 1 else (rng.rand(h, w) * 255).astype(np.uint8)
        msk = (rng.rand(h, w) * 21).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(tmp, "img{}.png".format(i)))
        Image.fromarray(msk).save(os.path.join(tmp, "msk{}.png".format(i)))
        st.put("file{}/image".format(i), img); st.put("file{}/mask".format(i), msk)
        names.append(("img{}.png".format(i), "msk{}.png".format(i)))
    with open(os.path.join(tmp, "two.lst"), "w") as f:
        f.write("".join("{}\t{}\n".format(a, b) for a, b in names))
    with open(os.path.join(tmp, "one.lst"), "w") as f:
        f.write("".join("{}\n".format(b) for _, b in names))
    norm = rds.Normalise(1.0 / 255, np.array([0.485, 0.456, 0.406]).reshape((1, 1, 3)),
                         np.array([0.229, 0.224, 0.225]).reshape((1, 1, 3)))
    trn = Seq([rds.ResizeShorter(16), rds.CentralCrop(30), rds.RandomCrop(24), norm, rds.ToTensor()])
    val = Seq([rds.CentralCrop(32), norm, rds.ToTensor()])
    ds = rds.PascalCustomDataset(os.path.join(tmp, "two.lst"), tmp, trn, val)
    assert len(ds) == 3
    np.random.seed(9)
    for i in range(3):
        out = ds[i]
        st.put("ds_trn{}/image".format(i), out["image"]); st.put("ds_trn{}/mask".format(i), out["mask"])
    ds.set_stage("val")
    for i in range(3):
        out = ds[i]
        st.put("ds_val{}/image".format(i), out["image"]); st.put("ds_val{}/mask".format(i), out["mask"])
    ds.set_stage("train")
    ds.set_config(20, 8)  # crop_size of transforms[2] (RandomCrop), resize_side of transforms[0]
    np.random.seed(10)
    out = ds[2]
    st.put("ds_cfg/image", out["image"]); st.put("ds_cfg/mask", out["mask"])
    try:
        one = rds.PascalCustomDataset(os.path.join(tmp, "one.lst"), tmp, None, None)
        single = [list(k) for k in one.datalist]
    except Exception as e:  # (record what the reference does with a one-column list under Python 3)
        single = "raises " + type(e).__name__
    st.save("data.npz")
    json.dump({"cases": cases, "pad": [70, [10, 20, 30], 255], "ccrop": 33, "rcrop": 41, "single_column": single,
               "names": names}, open(os.path.join(OUT, "data_meta.json"), "w"))
    shutil.rmtree(tmp, ignore_errors=True)
    print("data: single-column list ->", single)


if __name__ == "__main__":
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    import warnings

    warnings.filterwarnings("ignore")
    shutil.rmtree(TMP, ignore_errors=True)
    build_cython()
    which = sys.argv[1:] or ["ops", "nets", "miou", "engine", "controller", "nets_sampled", "engine_optim",
                             "engine_kd", "teacher", "data"]
    for w in which:
        globals()["gen_" + w]()
