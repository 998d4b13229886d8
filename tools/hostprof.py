"""Host cost of one training step: the headline network on a tiny image (the GPU work vanishes, what is
left is Python + launch overhead), wall time per step and a cProfile of ten steps.  Run on the GPU box:
python tools/hostprof.py"""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nas_segm_amd
from nas_segm_amd.engine.trainer import segmenter_step
dev = torch.device("cuda", 0)
seg, net = bench.build_model(dev, "headline"); seg.train()
oe = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
od = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
image, mask = bench.synthetic_batch(1, 64, 128, 0, dev, 19)   # tiny: the host is the bottleneck
step = lambda: segmenter_step(seg, image, mask, oe, od, 255, 3.0, 3.0, -1)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("host ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
