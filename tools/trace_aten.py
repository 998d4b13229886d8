"""Which ATen ops (copies, fills, adds ...) still launch GPU work inside one headline training step, and
from which Python lines?  The nasseg launches are counted by bench.py's LaunchProfiler; what autograd,
the optimisers and stray tensor methods launch is invisible there (rocprofv3: ~130 copyBuffer and ~100
fill launches per step in round 2).  Run on the GPU box:  python tools/trace_aten.py [workload] [steps]"""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd.engine.trainer import segmenter_step  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "headline"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[workload]
seg, net = bench.build_model(dev, workload)
seg.train()
oe = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
od = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
image, mask = bench.synthetic_batch(wl[3], wl[4], wl[5], 0, dev, wl[2])


def step():
    return segmenter_step(seg, image, mask, oe, od, 255, 3.0, 3.0, -1)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(steps):
        step()
    torch.cuda.synchronize()

# GPU kernels by name
kern = Counter()
ktime = Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:70]] += 1
        ktime[e.name[:70]] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
print("== GPU activities that are not nasseg kernels (per step) ==")
for name, n in kern.most_common():
    if "anonymous namespace" in name or "nasseg" in name:
        continue
    print("{:8.1f} x {:9.1f} us  {}".format(n / steps, ktime[name] / max(n, 1), name))

# CPU-side aten ops that launched something, with their Python call sites
sites = Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    if not e.kernels:
        continue
    stack = [s for s in (e.stack or []) if ".py" in s and "torch/" not in s][:3]
    if not stack:
        stack = [s for s in (e.stack or []) if ".py" in s][:3]
    sites[(e.name, " <- ".join(s.strip().split("/")[-1] for s in stack))] += len(e.kernels)
print("== aten ops with GPU launches, by call site (launches per step) ==")
for (name, where), n in sites.most_common(40):
    print("{:8.1f}  {:28s} {}".format(n / steps, name, where))
