"""Launch list of one training step WITHOUT a GPU: every libnasseg entry point the host code would call
(name + integer arguments), in order, for a bench workload.  The kernels are not run - `lib.call` is
replaced by a recorder and tensors are uninitialised host memory - so this is a planning tool for the
host-side graph (how many launches, which shapes, what is still a pass of its own), not a measurement.

    python tools/dry_trace.py [workload] [--list] [--grep name]
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import _lib  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "headline"
want_list = "--list" in sys.argv
grep = sys.argv[sys.argv.index("--grep") + 1] if "--grep" in sys.argv else None

calls = []
phase = ["fwd"]


def record(name, *args):
    calls.append((phase[0], name, tuple(a for a in args if isinstance(a, int) and abs(a) < (1 << 24))))
    return 0


_lib.lib.call = record
for mod in list(sys.modules.values()):
    name = getattr(mod, "__name__", "")
    if not name.startswith("nas_segm_amd"):
        continue
    if hasattr(mod, "require_device"):
        mod.require_device = lambda *a: None
    if hasattr(mod, "current_stream"):
        mod.current_stream = lambda: 0

wl = bench.WORKLOADS[workload]
dev = torch.device("cpu")
from nas_segm_amd.engine import trainer  # noqa: E402

seg, net = bench.build_model(dev, workload)
net.train()
batch = int(os.environ.get("BATCH", wl[3]))
image = torch.empty(batch, 3, wl[4], wl[5]).contiguous(memory_format=torch.channels_last)
mask = torch.zeros(batch, wl[4], wl[5], dtype=torch.long)
oe = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
od = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
real_backward = torch.Tensor.backward


acc = []  # autograd's own accumulations of a tensor's gradient over several consumers (one ATen add each)


def backward(self, *a, **k):
    from torch.profiler import ProfilerActivity, profile

    phase[0] = "bwd"
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        real_backward(self, *a, **k)
    # (nothing else in backward is an ATen add: the kernels are not run)
    acc.extend(e for e in prof.events() if e.name in ("aten::add", "aten::add_"))
    phase[0] = "post"


torch.Tensor.backward = backward
trainer.segmenter_step(net, image, mask, oe, od, 255, 0.0, 0.0, -1)  # (no clipping: its norms are ATen ops)

by = collections.Counter((p, n) for p, n, _ in calls)
tot = collections.Counter(n for _, n, _ in calls)
# entry points that are two kernels (a first stage and its finaliser)
TWO = ("nasseg_bn_bwd_reduce", "nasseg_colred", "nasseg_bn_stats")
kernels = len(calls) + sum(1 for _, n, _ in calls if n.replace("nasseg_bf16_", "nasseg_") in TWO)
n_acc = len(acc)
sys.stdout.write("{} nasseg launches per step = {} kernels (ATen optimiser kernels not included), {} gradient "
                 "accumulations by autograd (one ATen add each)\n".format(len(calls), kernels, n_acc))
for n, c in tot.most_common():
    sys.stdout.write("  {:34s} {:4d}   fwd {:3d}  bwd {:3d}\n".format(n, c, by[("fwd", n)], by[("bwd", n)]))
if want_list or grep:
    for p, n, a in calls:
        if grep is None or grep in n:
            sys.stdout.write("{} {} {}\n".format(p, n, list(a)))
