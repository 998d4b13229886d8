"""how often does the native optimiser step rebuild its tables in host-launched steps?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
import nas_segm_amd  # noqa: F401
from nas_segm_amd.engine.trainer import segmenter_step
from nas_segm_amd.engine.optim_native import cached_stepper
dev = torch.device("cuda", 0)
for wl_name in ("headline", "cvpr321"):
    wl = bench.WORKLOADS[wl_name]
    seg, net = bench.build_model(dev, wl_name)
    seg.train()
    oe = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    od = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
    image, mask = bench.synthetic_batch(wl[3], wl[4], wl[5], 0, dev, wl[2])
    for i in range(8):
        segmenter_step(seg, image, mask, oe, od, 255, 3.0, 3.0, -1)
    torch.cuda.synchronize()
    print(wl_name, "steps 8, table rebuilds", cached_stepper(oe).rebuilds, "gradient moves", cached_stepper(oe).moves)
    del seg, net, oe, od
