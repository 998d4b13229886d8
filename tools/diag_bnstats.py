"""One train-mode forward of a golden network under NASSEG_PWN_MODE 0 and 2: per BatchNorm, how far the batch
statistics (read off the running buffers, momentum 0.1 from 0 / 1) and the logits move.  usage: python tools/diag_bnstats.py NAME"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_hip_golden as T  # noqa: E402
from _util import build_product_net  # noqa: E402
from nas_segm_amd import functional as F  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "wacv_sampled1"
rec, npz = T._net_record(name)
F.lib.load()
res = []
for mode in (0, 2):
    F.lib._fn["nasseg_conv_pwn_mode"](mode)
    F.lib._memo.clear()
    net = build_product_net(rec["kind"], rec["genotype"], rec["classes"], rec["dec_kwargs"], rec["seed"]).to(T.DEV).train()
    x = T.cl(npz[name + "/x"])
    out = net(x)
    out = out[0] if isinstance(out, tuple) else out
    bufs = {k: v.detach().double().cpu().clone() for k, v in net.state_dict().items() if "running" in k}
    res.append((out.detach().double().cpu(), bufs))
(o0, b0), (o2, b2) = res
print(name, "input", tuple(npz[name + "/x"].shape), "logits max diff", float((o0 - o2).abs().max()), "of", float(o0.abs().max()))
rows = []
for k in b0:
    if "running_mean" in k:
        mean0, mean2 = b0[k] / 0.1, b2[k] / 0.1
        kv = k.replace("running_mean", "running_var")
        var0, var2 = (b0[kv] - 0.9) / 0.1, (b2[kv] - 0.9) / 0.1
        std = var0.clamp_min(1e-12).sqrt()
        dm = ((mean0 - mean2).abs() / std).max()
        dv = ((var0 - var2).abs() / var0.abs().clamp_min(1e-12)).max()
        ratio = (mean0.abs() / std).max()
        rows.append((float(max(dm, dv)), k, float(dm), float(dv), float(ratio), float(var0.min())))
rows.sort(reverse=True)
for r in rows[:12]:
    print("{:60s} dmean/std {:.2e} dvar/var {:.2e} max|mean|/std {:.1f} min var {:.2e}".format(r[1], r[2], r[3], r[4], r[5]))
