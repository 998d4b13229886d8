"""nasseg_dwconv forward (prologue + statistics) with and without the XCD-aware tile order (nasseg_dw_swizzle),
buffers rotated over > 512 MB so that nothing survives in the caches between launches.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
lib.load()
DEV = "cuda:0"
# B, C, H, W, K, stride, pad, dil
CASES = [(4, 64, 256, 512, 5, 1, 12, 6), (4, 64, 256, 512, 3, 1, 1, 1), (4, 24, 256, 512, 5, 1, 12, 6),
         (4, 32, 128, 256, 5, 1, 2, 1), (4, 32, 128, 256, 5, 1, 12, 6), (4, 96, 512, 1024, 3, 2, 1, 1),
         (4, 144, 256, 512, 3, 1, 1, 1), (4, 32, 512, 1024, 3, 1, 1, 1), (4, 192, 128, 256, 3, 1, 1, 1),
         (4, 64, 256, 512, 3, 1, 3, 3), (4, 48, 128, 256, 3, 1, 1, 1)]


def timeit(fn, n, reps=24):
    for i in range(4):
        fn(i % n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B, C, H, W, K, stride, pad, dil in CASES:
    Ho, Wo = F.conv_out_size(H, K, stride, pad, dil), F.conv_out_size(W, K, stride, pad, dil)
    nbytes = 4 * B * C * (H * W + Ho * Wo)
    n = max(2, (600 << 20) // nbytes)
    xs = [torch.randn(B, H, W, C, device=DEV) for _ in range(n)]
    ys = [torch.empty(B, Ho, Wo, C, device=DEV) for _ in range(n)]
    w = torch.randn(C, 1, K, K, device=DEV)
    wt = torch.empty(K * K * C, device=DEV)
    lib.call("nasseg_dw_pack_weight", ptr(w), ptr(wt), C, K, 0, stream())
    isc, ish = torch.rand(C, device=DEV) + 0.5, torch.rand(C, device=DEV)
    nb = lib.query("nasseg_dwconv_stats_blocks", B, C, Ho, Wo, K, stride, dil)
    stats = torch.empty((nb + 64) * 2 * C, device=DEV)
    s = stream()

    def run(i):
        lib.call("nasseg_dwconv", ptr(xs[i]), ptr(wt), ptr(ys[i]), ptr(isc), ptr(ish), 2, None, None, 0, B, H, W, C,
                 Ho, Wo, K, stride, pad, dil, 0, ptr(stats), s)

    out = []
    ref = None
    for swz in (0, 1, 0, 1):
        lib.call("nasseg_dw_swizzle", swz)
        us = timeit(run, n)
        out.append("swz={} {:7.1f} us {:5.2f} TB/s".format(swz, us, nbytes / us / 1e6))
        run(0)
        torch.cuda.synchronize()
        if ref is None:
            ref = ys[0].clone()
        else:
            assert torch.equal(ref, ys[0])
    print("C={:3d} {}x{} k{} s{} d{}: ".format(C, H, W, K, stride, dil) + " | ".join(out), flush=True)
