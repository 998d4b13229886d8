"""How close to their tolerance the whole-network gradient-mass checks of tests/test_hip_golden.py::test_network sit,
per kernel setting (NASSEG_PWN_MODE): the ratio |mass - reference| / tolerance of every gradient tensor.
usage (GPU box): NASSEG_PWN_MODE=0|1|2 python tools/diag_mass.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_hip_golden as T  # noqa: E402
from _util import build_product_net, checksums  # noqa: E402
from nas_segm_amd import functional as F  # noqa: E402

# NASSEG_DIAG_SPLIT=fwd|bwd: only the forward (nasseg_conv_fwd) / only the backward-data (nasseg_conv_bwd_data_bn) calls
# follow NASSEG_PWN_MODE; the others run with mode 0
SPLIT = os.environ.get("NASSEG_DIAG_SPLIT")
if SPLIT:
    F.lib.load()
    want = int(os.environ.get("NASSEG_PWN_MODE", "1"))
    real_call, real_query = F.lib.call, F.lib.query
    setm = F.lib._fn["nasseg_conv_pwn_mode"]

    def is_bwd_call(name):
        return name.endswith("conv_bwd_data_bn")

    def call(name, *a):
        if name.endswith("conv_fwd") or is_bwd_call(name):
            on = (SPLIT == "bwd") == is_bwd_call(name)
            setm(want if on else 0)
        return real_call(name, *a)

    def query(name, *a):
        if name == "nasseg_conv_fwd_stats_blocks":
            on = (SPLIT == "bwd") == (a[-1] == 2)
            setm(want if on else 0)
            F.lib._memo.clear()
        return real_query(name, *a)

    F.lib.call, F.lib.query = call, query
    print("split:", SPLIT)
ONLY = os.environ.get("NASSEG_DIAG_ONLY")
names = sorted(T.NETS_META) + sorted(k for k in T.SAMPLED_META if not k.endswith("_train"))
print("NASSEG_PWN_MODE =", os.environ.get("NASSEG_PWN_MODE"))
for name in names:
    rec, npz = T._net_record(name)
    if rec["classes"] <= 1 or (ONLY and name not in ONLY.split(",")):
        continue
    net = build_product_net(rec["kind"], rec["genotype"], rec["classes"], rec["dec_kwargs"], rec["seed"]).to(T.DEV)
    x = T.cl(npz[name + "/x"])
    net.train()
    target = torch.from_numpy(npz[name + "/target"]).to(T.DEV)
    output = net(x)
    aux_outs = []
    if isinstance(output, tuple):
        output, aux_outs = output
    tv = F.nearest_label_resize(target, output.shape[2:])
    loss = F.log_softmax_nll(output, tv, 255)
    if rec["aux_weight"] > 0:
        for a in aux_outs:
            a = F.bilinear_resize(a, tv.shape[1:])
            loss = loss + F.log_softmax_nll(a, tv, 255) * rec["aux_weight"]
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    got = checksums({k: v.cpu() for k, v in grads.items()})
    ratios = []
    for k, (s, sa) in rec["grad_checksums"].items():
        floor = rec["grad_mass_sensitivity"][k]
        tol = 2e-3 * sa + 4.0 * floor + 1e-6
        ratios.append((abs(got[k][1] - sa) / tol, k, abs(got[k][1] - sa), 2e-3 * sa, floor))
    ratios.sort(reverse=True)
    r = [q[0] for q in ratios]
    print("{:22s} n={:4d} max {:.2f} | >1: {:2d} >0.5: {:3d} >0.25: {:3d} | worst: {} diff {:.2e} 2e-3*mass {:.2e} floor {:.2e}".format(
        name, len(r), r[0], sum(v > 1 for v in r), sum(v > 0.5 for v in r), sum(v > 0.25 for v in r), ratios[0][1],
        ratios[0][2], ratios[0][3], ratios[0][4]))
