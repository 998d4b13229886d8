"""Repro / root-cause harness for the stall seen in round 1 with `bench.py --graph 1` and two ranks
sharing one device: per-step host times of every rank for

    eager+ar   host launches, gradient all-reduce            (the data-parallel engine path of round 2)
    graph+ar   hipGraph replay of fwd+loss+bwd, all-reduce   (what stalled)
    graph      hipGraph replay, NO collective                (is it the graph alone, two processes on a device?)
    graph+bar  hipGraph replay, a host barrier instead of the all-reduce (is it the collective's device work?)

The 1-GPU box cannot run RCCL between two ranks (one device per rank is required), so the collective here
is gloo (host staging) - which is itself part of the question.  Run on the GPU box:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        tools/repro_dp_graph.py [workload] [steps]
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import bench  # noqa: E402
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd.engine import RankParallel, Segmenter  # noqa: E402,F401
from nas_segm_amd.engine.graphed import GraphedSegmenterStep  # noqa: E402
from nas_segm_amd.engine.trainer import segmenter_step  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cvpr321"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
wl = bench.WORKLOADS[workload]
batch = batch or max(1, wl[3] // world)


def run(variant):
    seg, net = bench.build_model(dev, workload)
    seg.train()
    oe = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    od = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
    image, mask = bench.synthetic_batch(batch, wl[4], wl[5], rank, dev, wl[2])
    if variant == "eager+ar":
        def step():
            return segmenter_step(seg, image, mask, oe, od, 255, 3.0, 3.0, -1)
    else:
        target = seg if variant == "graph+ar" else net  # (net: no world size -> no collective in the stepper)
        g = GraphedSegmenterStep(target, image, mask, oe, od, 255, 3.0, 3.0, -1)

        def step():
            loss = g.step(image, mask)
            if variant == "graph+bar" and world > 1:
                dist.barrier()
            return loss
    times = []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    for _ in range(steps):
        t0 = time.perf_counter()
        loss = step()
        float(loss)  # (the engine's per-step host sync: loss.item())
        times.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    med = sorted(times)[len(times) // 2]
    worst = max(times[3:]) if len(times) > 3 else max(times)
    sys.stdout.write("rank {} {:9s} median {:7.2f} ms  worst(after 3) {:8.2f} ms  steps: {}\n".format(
        rank, variant, med, worst, " ".join("{:.1f}".format(t) for t in times)))
    sys.stdout.flush()
    del seg, net
    import gc
    gc.collect()
    torch.cuda.empty_cache()


for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["eager+ar", "graph", "graph+bar", "graph+ar"]):
    run(v)
    if world > 1:
        dist.barrier()
if world > 1:
    dist.destroy_process_group()
