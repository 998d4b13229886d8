"""Where does the time between two replays of a recorded step go?  (a 0.55 - 0.68 ms hole
between consecutive replays of the CVPR 321x321 step.)  Times, without a profiler attached:
  full      graphed.step(image, mask) as bench.py calls it
  static    graphed.step() - inputs already in the static buffers
  replay    graph.replay() alone, and the host time inside each call
  twice     two replays per loop iteration
    python tools/replay_probe.py [workload] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cvpr321"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    import nas_segm_amd  # noqa: F401
    from nas_segm_amd.engine.graphed import GraphedSegmenterStep

    device = torch.device("cuda", 0)
    wl = bench.WORKLOADS[workload]
    segmenter, net = bench.build_model(device, workload)
    segmenter.train()
    optim_enc = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    optim_dec = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
    image, mask = bench.synthetic_batch(wl[3], wl[4], wl[5], 0, device, wl[2])
    g = GraphedSegmenterStep(segmenter, image, mask, optim_enc, optim_dec, 255, 3.0, 3.0, -1, capture_optimisers=True)
    print("layout:", getattr(g, "layout", None))

    def timed(name, fn, per=1):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for _ in range(steps):
            a = time.perf_counter()
            fn()
            host.append(time.perf_counter() - a)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps / per
        host.sort()
        print("{:8s} {:8.3f} ms per step; host per call: median {:.3f} max {:.3f} ms".format(
            name, 1e3 * t, 1e3 * host[len(host) // 2], 1e3 * host[-1]))

    timed("full", lambda: g.step(image, mask))
    timed("static", lambda: g.step())
    once = g.plan.run if getattr(g, "plan", None) is not None else g.graph.replay
    timed("replay", once)
    timed("twice", lambda: (once(), once()), per=2)
    timed("full", lambda: g.step(image, mask))
    # one replay alone on an idle GPU, by events on the step's stream: what a step costs the GPU without a neighbour
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        once()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("one replay alone (events): median {:.3f} ms, min {:.3f} ms".format(ts[len(ts) // 2], ts[0]))


if __name__ == "__main__":
    main()
