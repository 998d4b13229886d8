"""The pointwise convs of the headline step (WACV arch0, 4x3x1024x2048), forward with BatchNorm statistics and
backward-data with BatchNorm-backward sums, timed per kernel choice through the C-ABI:
    pwn 0 = conv_fwd_kernel / conv_pw_kernel as dispatched today, pwn 2 = conv_pwn_kernel (csrc/conv_pwn.hip).
Buffers rotate over > 600 MB so that nothing is served by the 256 MiB Infinity Cache.
usage (GPU box): python tools/kbench_pwn.py [fwd|bwd|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

DEV = "cuda:0"
# (K, N, H, W, prologue) of forward calls / (K = channels of dy, N = channels of g and z, H, W) of backward-data calls
FWD = [(16, 96, 512, 1024, 0), (24, 144, 256, 512, 0), (32, 192, 128, 256, 0), (32, 64, 128, 256, 0),
       (224, 64, 256, 512, 1), (128, 64, 256, 512, 1), (144, 24, 256, 512, 1), (96, 16, 512, 1024, 1),
       (32, 32, 512, 1024, 1), (64, 32, 256, 512, 1), (192, 32, 128, 256, 1), (64, 64, 256, 512, 0),
       (24, 96, 256, 512, 0), (32, 16, 512, 1024, 1), (64, 128, 128, 256, 0), (64, 64, 32, 64, 0)]
BWD = [(64, 128, 256, 512), (32, 64, 256, 512), (24, 144, 256, 512), (16, 32, 512, 1024), (32, 192, 128, 256),
       (24, 96, 256, 512), (64, 128, 128, 256), (16, 96, 512, 1024), (64, 224, 256, 512), (32, 64, 128, 256)]
B = int(os.environ.get("KBENCH_B", "4"))
if os.environ.get("KBENCH_FWD"):  # "K,N,H,W,pro;..."
    FWD = [tuple(int(v) for v in c.split(",")) for c in os.environ["KBENCH_FWD"].split(";")]
if os.environ.get("KBENCH_BWD"):  # "K,N,H,W;..."
    BWD = [tuple(int(v) for v in c.split(",")) for c in os.environ["KBENCH_BWD"].split(";")]


def cl(c, h, w, dtype):
    return torch.randn(B, c, h, w, device=DEV).contiguous(memory_format=torch.channels_last).to(dtype)


def timeit(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def set_mode(pwn, pw):
    F.lib.load()
    F.lib._memo.clear()
    F.lib._fn["nasseg_conv_pwn_mode"](pwn)
    F.lib._fn["nasseg_conv_pw_min_pixels"](pw)
    F.lib._memo.clear()


def bench(which, dtype):
    pre = "nasseg_bf16_" if dtype == torch.bfloat16 else "nasseg_"
    esz = 2 if dtype == torch.bfloat16 else 4
    s = F.current_stream()
    rows = FWD if which == "fwd" else BWD
    for row in rows:
        K, N, H, W = row[:4]
        pro = row[4] if which == "fwd" else 0
        nbytes = esz * B * H * W * (K + N * (2 if which == "bwd" else 1))
        R = max(2, int(600e6 // nbytes) + 1)
        xs = [cl(K, H, W, dtype) for _ in range(R)]
        ys = [cl(N, H, W, dtype) for _ in range(R)]
        zs = [cl(N, H, W, dtype) for _ in range(R)] if which == "bwd" else None
        w = (torch.randn(N, K, device=DEV) * 0.1).contiguous() if which == "fwd" else (
            torch.randn(K, N, device=DEV) * 0.1).contiguous()  # (mode 1 packing of a 1x1 conv: [K][N] -> rows of the "forward" N x K view)
        if which == "bwd":
            w = (torch.randn(N, K, device=DEV) * 0.1).contiguous()
        vec = [torch.rand(max(K, N), device=DEV) + 0.5 for _ in range(4)]
        res = {}
        for label, pwn, pw in (("now", 0, -2), ("general", 0, 1 << 40), ("pw", 0, 0), ("pwn", 2, -2)):
            set_mode(pwn, pw)
            nb = F.lib.query("nasseg_conv_fwd_stats_blocks", B, H, W, N, K, 1 if which == "fwd" else 2)
            part = torch.empty((nb + 64) * 2 * N, device=DEV)
            if which == "fwd":
                p = (F.ptr(vec[0]), F.ptr(vec[1]), 1) if pro else (None, None, 0)

                def fn(i):
                    F.lib.call(pre + "conv_fwd", F.ptr(xs[i % R]), K, F.ptr(w), F.ptr(ys[i % R]), N, *p, None, None, 0,
                               None, 0, B, H, W, K, H, W, N, 1, 1, 1, 0, 1, 0, F.ptr(part), s)
            else:
                def fn(i):
                    F.lib.call(pre + "conv_bwd_data_bn", F.ptr(xs[i % R]), K, F.ptr(w), F.ptr(ys[i % R]), N,
                               F.ptr(zs[i % R]), N, F.ptr(vec[0]), F.ptr(vec[1]), F.ptr(vec[2]), F.ptr(vec[3]), 1, B, H,
                               W, K, H, W, N, 1, 1, 1, 0, 1, F.ptr(part), s)
            res[label] = (timeit(fn, 30), nb)
        set_mode(1, -2)
        base = res["now"][0]
        print("{} {:5s} K{:4d} N{:4d} {:4d}x{:4d}{} {:7.1f} MB | now {:7.1f} us {:6.0f} GB/s | general {:7.1f} | pw {:7.1f} | "
              "pwn {:7.1f} us {:6.0f} GB/s grid {:4d}  x{:.2f}".format(
                  "bf16" if esz == 2 else "f32 ", which, K, N, H, W, " pro" if pro else "    ", nbytes / 1e6, base,
                  nbytes / base / 1e3, res["general"][0], res["pw"][0], res["pwn"][0], nbytes / res["pwn"][0] / 1e3,
                  res["pwn"][1], base / res["pwn"][0]), flush=True)
        del xs, ys, zs


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    dts = [torch.float32] + ([torch.bfloat16] if os.environ.get("KBENCH_BF16", "1") == "1" else [])
    for dt in dts:
        for which in ("fwd", "bwd"):
            if what in (which, "all"):
                bench(which, dt)
