"""segmenter_step against the same step written out with loss.backward(): where do the parameters first differ?"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import __graft_entry__ as entry

entry.build()
import torch

import test_hip_engine as T
from nas_segm_amd import functional as F
from nas_segm_amd.engine.trainer import _clip_and_step, segmenter_step

DEV = T.DEV
rec = T.load_json("nets_meta.json")["wacv_arch0"]
g = torch.Generator().manual_seed(9)
batches = [(torch.randn(2, 3, 97, 129, generator=g).to(DEV).contiguous(memory_format=torch.channels_last),
            torch.randint(0, 19, (2, 97, 129), generator=g).to(DEV)) for _ in range(2)]


def run(plain, nsteps):
    net = T.build_product_net(rec["kind"], rec["genotype"], rec["classes"], rec["dec_kwargs"], 0).to(DEV).train()
    oe = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    od = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
    grads = None
    for x, t in batches[:nsteps]:
        if plain:
            out = net(x)
            loss = F.log_softmax_nll(out, F.nearest_label_resize(t, out.shape[2:]), 255)
            oe.zero_grad()
            od.zero_grad()
            loss.backward()
            grads = dict((k, p.grad.clone()) for k, p in net.named_parameters())
            _clip_and_step([(list(net.encoder.parameters()), 3.0, oe), (list(net.decoder.parameters()), 3.0, od)])
        else:
            segmenter_step(net, x, t, oe, od, 255, 3.0, 3.0, -1)
            grads = dict((k, p.grad.clone()) for k, p in net.named_parameters())
    return T._cpu_sd(net), grads


batches.append(batches[0])
for n in (3, 3, 3):
    for env in ("1", "0"):
        os.environ["NASSEG_NATIVE_OPTIM"] = env
        import nas_segm_amd.engine.optim_native as ON
        ON.ENABLED = env != "0"
        a, ga = run(False, n)
        b, gb = run(True, n)
        bad = [k for k in a if not torch.equal(a[k], b[k])]
        badg = [k for k in ga if not torch.equal(ga[k], gb[k])]
        a2, _ = run(False, n)
        bad2 = [k for k in a if not torch.equal(a[k], a2[k])]
        print("steps", n, "native", env, "params differing:", len(bad), bad[:3], "| same path twice:", len(bad2), bad2[:3])
