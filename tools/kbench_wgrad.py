"""Direct (no autograd) timing of nasseg_conv_wgrad at the small / mid shapes of the headline step.
usage: python tools/kbench_wgrad.py          """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

DEV = "cuda:0"
SHAPES = [  # B, K, H, W, N, k
    (4, 64, 32, 64, 64, 1), (4, 128, 32, 64, 64, 1), (4, 32, 128, 256, 32, 1), (4, 32, 128, 256, 64, 1),
    (4, 128, 128, 256, 64, 1), (4, 64, 256, 512, 32, 1), (4, 24, 256, 512, 144, 1), (4, 144, 256, 512, 24, 1),
    (4, 128, 256, 512, 64, 1), (4, 224, 256, 512, 64, 1), (4, 16, 512, 1024, 96, 1), (4, 64, 256, 512, 19, 3),
    (4, 64, 256, 512, 20, 3), (16, 64, 81, 81, 64, 3), (16, 64, 81, 81, 64, 1), (8, 48, 179, 179, 48, 3),
]


if os.environ.get("KB_ONLY_HEAD"):
    SHAPES = [s for s in SHAPES if s[5] == 3 and s[4] == 19]


def main():
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("NASSEG_")})
    s = F.current_stream()
    for (B, K, H, W, N, k) in SHAPES:
        x = torch.randn(B, K, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, N, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
        dw = torch.empty(N, K, k, k, device=DEV)
        ws = torch.empty(F.lib.query("nasseg_conv_wgrad_workspace", B, H, W, N, K, k, k), device=DEV)

        def run():
            F.lib.call("nasseg_conv_wgrad", F.ptr(x), K, F.ptr(dy), N, F.ptr(dw), F.ptr(ws), None, None, 0,
                       B, H, W, K, H, W, N, k, k, 1, k // 2, 1, s)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n
        by = 4 * (x.numel() + dy.numel())
        print("wgrad K{:4d} N{:4d} {:4d}x{:4d} k{}: {:8.1f} us {:8.1f} GB/s  ws {:6.2f} MB".format(
            K, N, H, W, k, t * 1e3, by / t / 1e6, ws.numel() * 4 / 1e6))


if __name__ == "__main__":
    main()
