"""one SepConv stage through nasseg_sepconv_fwd (training form: depthwise output stored, statistics rows, prologue) on
rotated buffers - the target of the counter passes of `tools/gpu.sh pmc`.
    python tools/kbench_sepconv_one.py [B C N H W k pad dil]   (default: 4 64 64 128 256 5 2 1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
lib.load()
DEV = "cuda:0"
B, C, N, H, W, k, pad, dil = [int(v) for v in sys.argv[1:9]] if len(sys.argv) > 8 else (4, 64, 64, 128, 256, 5, 2, 1)
n = 6
xs = [torch.randn(B, H, W, C, device=DEV) for _ in range(n)]
zs = [torch.empty(B, H, W, C, device=DEV) for _ in range(n)]
ys = [torch.empty(B, H, W, N, device=DEV) for _ in range(n)]
wdw, wpw = torch.randn(C, 1, k, k, device=DEV), torch.randn(N, C, 1, 1, device=DEV)
wt = torch.empty(k * k * C, device=DEV)
lib.call("nasseg_dw_pack_weight", ptr(wdw), ptr(wt), C, k, 0, stream())
isc, ish = torch.rand(C, device=DEV) + 0.5, torch.rand(C, device=DEV)
nb = lib.query("nasseg_sepconv_blocks", B, C, H, W, N, k, 1, dil)
part = torch.empty((nb + 64) * 2 * N, device=DEV)
for i in range(12):
    lib.call("nasseg_sepconv_fwd", ptr(xs[i % n]), ptr(wt), ptr(wpw), ptr(zs[i % n]), ptr(ys[i % n]), ptr(isc), ptr(ish),
             1, None, None, 0, B, H, W, C, H, W, N, k, 1, pad, dil, ptr(part), stream())
torch.cuda.synchronize()
