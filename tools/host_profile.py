"""Where the HOST time of a headline step goes (cProfile over eager steps on the GPU box; the device runs behind).
    python tools/host_profile.py [workload] [steps]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd.engine.trainer import segmenter_step  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "headline"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[workload]
seg, net = bench.build_model(dev, workload)
seg.train()
oe = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
od = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
image, mask = bench.synthetic_batch(wl[3], wl[4], wl[5], 0, dev, wl[2])
for _ in range(4):
    segmenter_step(seg, image, mask, oe, od, 255, 3.0, 3.0, -1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    segmenter_step(seg, image, mask, oe, od, 255, 3.0, 3.0, -1)
host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / steps
sys.stdout.write("host {:.2f} ms/step enqueue, {:.2f} ms/step with the device drained\n".format(1e3 * host, 1e3 * total))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    segmenter_step(seg, image, mask, oe, od, 255, 3.0, 3.0, -1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(32)
