"""where the batch-3 ir_24_24 anchor differs from torch: clustered (a ReLU6 kink flip) or spread (a bug)?"""
import copy, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import __graft_entry__ as entry
entry.build()
import torch
import test_hip_anchor as T

class MP:
    def setattr(self, o, n, v): setattr(o, n, v)
Fm = T.lower_thresholds(MP())
for seed in (2, 7):
    mods, cin, (H, W), residual, relu_in = T.build("ir_24_24")
    T.randomise(mods, 5)
    ref = copy.deepcopy(mods).train(); ref64 = copy.deepcopy(mods).double().train()
    mods = mods.to(T.DEV).train()
    x0 = T.rnd(3, cin, H, W, seed=seed)
    xc = x0.clone().requires_grad_(True)
    yc = T.torch_reference(ref._modules.values(), xc, xc, relu_in)
    cot = T.rnd(*yc.shape, seed=3); yc.backward(cot)
    xd = x0.double().requires_grad_(True)
    yd = T.torch_reference(ref64._modules.values(), xd, xd, relu_in); yd.backward(cot.double())
    xg = T.dev(x0.clone()).requires_grad_(True)
    yg = mods(xg, residual=xg, relu_in=relu_in); yg.backward(T.dev(cot))
    for name, a, b in (("gpu-vs-cpu32", xg.grad.cpu().double(), xc.grad.double()), ("cpu32-vs-cpu64", xc.grad.double(), xd.grad),
                       ("gpu-vs-cpu64", xg.grad.cpu().double(), xd.grad)):
        err = (a - b).abs()
        tol = 1e-4 * float(b.abs().max()) + 1e-4 * b.abs()
        bad = (err > tol).nonzero()
        px = sorted(set((int(i[0]), int(i[2]), int(i[3])) for i in bad))
        print(seed, name, "bad", len(bad), "pixels", len(px), px[:12], "max", float(err.max()))
