import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import __graft_entry__ as entry
entry.build()
import torch
from torch import nn
from nas_segm_amd.engine.optim_native import NativeStep
DEV = "cuda:0"
def params(seed):
    g = torch.Generator().manual_seed(seed)
    return [nn.Parameter(torch.randn(5000, generator=g).to(DEV))]
for name, mk in (("sgd lr", lambda p: torch.optim.SGD(p, lr=1e-3)),
                 ("sgd wd", lambda p: torch.optim.SGD(p, lr=1e-3, weight_decay=1e-5)),
                 ("sgd mom", lambda p: torch.optim.SGD(p, lr=1e-3, momentum=0.9)),
                 ("sgd mom wd", lambda p: torch.optim.SGD(p, lr=1e-3, momentum=0.9, weight_decay=1e-5)),
                 ("adam", lambda p: torch.optim.Adam(p, lr=3e-3)),
                 ("adam wd", lambda p: torch.optim.Adam(p, lr=3e-3, weight_decay=1e-5))):
    ref, ps = params(1), params(1)
    ro, o = mk(ref), mk(ps)
    nat = NativeStep.build([(ps, 0.0, o)])
    out = []
    for s in range(4):
        g = torch.randn(5000, generator=torch.Generator().manual_seed(100 + s)).to(DEV) * 0.3
        ps[0].grad, ref[0].grad = g.clone(), g.clone()
        nat.step(); ro.step()
        nd = int((ps[0].detach() != ref[0].detach()).sum())
        st = [k for k in ro.state[ref[0]] if k != "step" and not torch.equal(ro.state[ref[0]][k], o.state[ps[0]][k])]
        out.append("{}:{}{}".format(s, nd, st))
    print(name, " ".join(out))
