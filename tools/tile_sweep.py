"""conv3x3_lds_kernel: microseconds per launch for a list of output tiles (NASSEG_LDS3_TILE=th,tw) over a few maps - needs a
library built with -DNASSEG_TUNE as tools/build/variants/lib_tune.so (NASSEG_EXTRA_FLAGS=-DNASSEG_TUNE python
nas-segm-pytorch_amd/build.py after touching csrc/conv_fwd.hip).  What lds3x3_tile's model was fitted to."""
import os, sys, subprocess, itertools
shapes = {"8x30x40": (8, 64, 30, 40, 64, 1, 1), "8x60x80": (8, 64, 60, 80, 64, 1, 1), "16x41x41": (16, 64, 41, 41, 64, 1, 1),
          "16x81x81": (16, 64, 81, 81, 64, 1, 1), "8x60x80d3": (8, 64, 60, 80, 64, 3, 3), "8x90x90": (8, 48, 90, 90, 48, 1, 1),
          "64x64x64": (64, 48, 64, 64, 48, 1, 1), "8x60x80n21": (8, 64, 60, 80, 21, 1, 1)}
tiles = [(8, 32), (4, 16), (2, 32), (4, 32), (8, 16), (6, 32), (6, 21), (9, 27), (7, 9), (5, 12), (8, 8), (12, 21), (3, 41), (6, 41), (4, 48), (3, 64), (16, 16), (10, 20), (5, 40), (6, 40), (4, 40), (3, 40), (2, 40), (5, 20), (10, 10), (6, 10), (15, 16), (15, 8), (8, 30), (6, 30), (4, 30), (4, 45), (5, 45), (3, 45), (6, 45)]
code = '''
import os, sys, torch
sys.path.insert(0, %r)
import nas_segm_amd
from nas_segm_amd import functional as F
lib, ptr, stream = F.lib, F.ptr, F.current_stream
lib.load()
B, K, H, W, N, pad, dil = %r
x = torch.randn(B, H, W, K, device="cuda:0"); w = torch.randn(N, K, 3, 3, device="cuda:0") * 0.05
wp = torch.empty(9 * N * K, device="cuda:0"); s = stream()
lib.call("nasseg_conv_pack_weight", ptr(w), ptr(wp), N, K, 3, 3, 0, s)
y = torch.empty(B, H, W, N, device="cuda:0")
fn = lambda: lib.call("nasseg_conv_fwd", ptr(x), K, ptr(wp), ptr(y), N, None, None, 0, None, None, 0, None, 0, B, H, W, K, H, W, N, 3, 3, 1, pad, dil, 0, None, s)
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): fn()
e1.record(); torch.cuda.synchronize()
print("US", e0.elapsed_time(e1) / 40 * 1e3)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, shp in shapes.items():
    res = []
    for t in tiles:
        if t[0] > shp[2] or t[1] > shp[3]:
            continue
        env = dict(os.environ, NASSEG_LIB=os.path.join(root, "tools/build/variants/lib_tune.so"), NASSEG_LDS3_TILE="%d,%d" % t)
        out = subprocess.run([sys.executable, "-c", code % (root, shp)], env=env, capture_output=True, text=True).stdout
        us = [float(l.split()[1]) for l in out.splitlines() if l.startswith("US")]
        if us:
            res.append((us[0], t))
    res.sort()
    print(name, " ".join("%dx%d:%.1f" % (t[0], t[1], u) for u, t in res), flush=True)
