"""one depthwise 5x5 dilation-6 forward (64 channels, 4 x 256 x 512, prologue + statistics), a few launches on
rotated buffers: the target of the counter passes of `tools/gpu.sh pmc`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
lib.load()
DEV = "cuda:0"
B, C, H, W, K, stride, pad, dil = 4, 64, 256, 512, 5, 1, 12, 6
if len(sys.argv) > 1:
    K, pad, dil = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 4
xs = [torch.randn(B, H, W, C, device=DEV) for _ in range(n)]
ys = [torch.empty(B, H, W, C, device=DEV) for _ in range(n)]
w = torch.randn(C, 1, K, K, device=DEV)
wt = torch.empty(K * K * C, device=DEV)
lib.call("nasseg_dw_pack_weight", ptr(w), ptr(wt), C, K, 0, stream())
isc, ish = torch.rand(C, device=DEV) + 0.5, torch.rand(C, device=DEV)
nb = lib.query("nasseg_dwconv_stats_blocks", B, C, H, W, K, stride, dil)
stats = torch.empty((nb + 64) * 2 * C, device=DEV)
for i in range(8):
    lib.call("nasseg_dwconv", ptr(xs[i % n]), ptr(wt), ptr(ys[i % n]), ptr(isc), ptr(ish), 2, None, None, 0, B, H, W, C,
             H, W, K, stride, pad, dil, 0, ptr(stats), stream())
torch.cuda.synchronize()
