"""Headline benchmark: images/sec of the NAS inner-loop training step
(forward + backward + gradient all-reduce + clip + optimiser step) of the WACV
arch0 segmenter (MobileNetV2[1,2] encoder + TemplateDecoder, 19 classes) on
synthetic 2048x1024 images, 4 images per GPU, fp32 - BASELINE.json `metric`.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on
the launch stream (LaunchProfiler) in extra, untimed steps after the timed
region; `cpu_baseline` times the CPU oracle (oracle/, the restatement of the
reference's torch graph) on the host cores of the same box, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver supports dmabuf IPC only: without this RCCL's buffer exchange between the ranks of a
# node fails (hipIpcGetMemHandle: invalid argument); the image exports it, keep it if launched bare
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WACV_ARCH0 = [[[3, 0, 1], [4, 1, 1], [3, 1, 1]],
              [[0, 1, 0, 0, 1], [2, 1, 2, 1, 0], [3, 1, 1, 1, 0], [1, 1, 2, 0, 0],
               [3, 0, 2, 0, 0], [5, 3, 2, 1, 0], [0, 5, 0, 1, 0]]]
WACV_ARCH1 = [[[1, 1, 0], [1, 3, 0], [3, 4, 0]],
              [[1, 1, 0, 0, 0], [0, 1, 1, 1, 1], [3, 1, 2, 3, 0], [3, 0, 2, 2, 0],
               [0, 1, 2, 0, 0], [2, 1, 1, 3, 0], [4, 0, 2, 2, 0]]]
CVPR_ARCH0 = [[8, [0, 0, 5, 2], [0, 2, 8, 8], [0, 5, 1, 4]], [[3, 3], [3, 2], [3, 0]]]
CVPR_ARCH2_DEPTH = [[5, [0, 0, 4, 1], [3, 2, 0, 1], [5, 6, 5, 0]], [[1, 3], [4, 3], [2, 2]]]
# name -> (decoder kind, genotype, classes, default batch/GPU, H, W, description)
WORKLOADS = {
    "headline": ("template", WACV_ARCH0, 19, 4, 1024, 2048, "WACV arch0 (BASELINE metric)"),
    "arch1": ("template", WACV_ARCH1, 19, 4, 1024, 2048, "WACV arch1 (BASELINE config 3 shape)"),
    "cvpr321": ("micro", CVPR_ARCH0, 21, 16, 321, 321, "CVPR arch0 VOC 321x321 bs16 (BASELINE config 2)"),
    # one genotype sampled by the reference controller per rank (tests/golden/controller.json), search
    # defaults (agg 48, sep repeats 1); the reward of one batch is computed with the HIP mIoU kernels
    "search713": ("sampled", None, 19, 8, 713, 713, "sampled WACV cell per GPU, 713x713 bs8 (BASELINE config 4)"),
    # 1-channel depth head, berHu loss (BASELINE config 5 is bf16: run with --dtype bf16)
    "depth480": ("micro", CVPR_ARCH2_DEPTH, 1, 8, 480, 640, "CVPR depth arch, berHu, 480x640 bs8 (BASELINE config 5)"),
    # the decoder-only step on the device-resident encoder-feature cache (5 of the
    # 6 inner epochs of a candidate, default_args.py:52): search-time decoder (agg 48, sep repeats 1, aux
    # cells), 256x256 crops, batch 64 (default_args.py:5,24), cache of 1024 samples per GPU (sharded)
    "task0": ("micro_search", CVPR_ARCH0, 21, 64, 256, 256, "train_task0 step on the cached encoder features, "
              "CVPR search decoder, 256x256 crops bs64"),
    # the distillation teacher (Light-Weight RefineNet on ResNet-152, 62 M parameters), the
    # inference forward populate_task0 runs per training crop (src/engine/trainer.py:17-74); forward only
    "teacher": ("teacher", None, 21, 16, 256, 256, "KD teacher rf_lw152 inference forward, 256x256 crops bs16"),
}
NUM_CLASSES = 19
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # (same guide: 6.29 TB/s measured with a float4 copy)


def algorithmic_bytes(name, a):
    """ALGORITHMIC bytes of one launch of a C-ABI entry point (inputs read once + outputs
    written once, SURVEY.md section 8(d) per-primitive formulas).  `a` is the positional
    argument tuple of the call (include/nasseg.h order).  The bfloat16 twins move half the
    bytes (weights and per-channel vectors, which stay fp32, are negligible)."""
    if name.startswith("nasseg_bf16_"):
        return algorithmic_bytes("nasseg_" + name[len("nasseg_bf16_"):], a) // 2
    if name == "nasseg_dwconv":
        B, H, W, C, Ho, Wo, K = a[9], a[10], a[11], a[12], a[13], a[14], a[15]
        return 4 * (B * C * H * W + B * C * Ho * Wo + C * K * K)
    if name == "nasseg_sepconv_fwd":  # x read, pointwise output written (+ the depthwise output when stored)
        B, H, W, C, Ho, Wo, N, K = a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18]
        return 4 * (B * C * H * W + B * N * Ho * Wo + (B * C * Ho * Wo if a[3] else 0) + C * K * K + N * C)
    if name == "nasseg_dwconv_wgrad":
        B, H, W, C, Ho, Wo, K = a[7], a[8], a[9], a[10], a[11], a[12], a[13]
        return 4 * (B * C * H * W + B * C * Ho * Wo + C * K * K)
    if name == "nasseg_conv_fwd":
        B, Hs, Ws, K, Ho, Wo, N, kh, kw = a[13], a[14], a[15], a[16], a[17], a[18], a[19], a[20], a[21]
        res = B * Ho * Wo * N if a[11] else 0
        return 4 * (B * Hs * Ws * K + B * Ho * Wo * N + N * K * kh * kw + res)
    if name == "nasseg_conv_bwd_data_bn":  # dy read, g written, z read once (fused BN-backward sums)
        B, Hs, Ws, K, Ho, Wo, N, kh, kw = a[12], a[13], a[14], a[15], a[16], a[17], a[18], a[19], a[20]
        return 4 * (B * Hs * Ws * K + 2 * B * Ho * Wo * N + N * K * kh * kw)
    if name == "nasseg_dwconv_bwd_data_bn":
        B, H, W, C, Ho, Wo, K = a[9], a[10], a[11], a[12], a[13], a[14], a[15]
        return 4 * (B * C * H * W + 2 * B * C * Ho * Wo + C * K * K)
    if name == "nasseg_conv_wgrad":
        B, Hs, Ws, K, Ho, Wo, N, kh, kw = a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17]
        return 4 * (B * Hs * Ws * K + B * Ho * Wo * N + N * K * kh * kw)
    if name == "nasseg_conv_pw_bwd_bn":  # x, g [, z] read, dx written; dz never leaves the CU
        B, H, W, K, N = a[18], a[19], a[20], a[21], a[22]
        from nas_segm_amd import functional as NF

        # (z = W x is rebuilt from the input tile where the kernel can: not an operand it needs, not charged)
        # (... and always where z == NULL: the expansion behind nasseg_irdw_fwd was never stored)
        nz = 2 if (a[2] and NF.lib.query("nasseg_conv_pw_bwd_reads_z", B, H, W, K, N)) else 1
        skip = 1 if a[26] else 0  # (dx_res: the skip's gradient, read once)
        return 4 * (B * H * W * ((2 + skip) * K + nz * N) + 2 * N * K)
    if name == "nasseg_dwconv_bwd_bn":  # xz, g, z read, ge written
        B, H, W, C, Ho, Wo = a[20], a[21], a[22], a[23], a[24], a[25]
        return 4 * (2 * B * C * H * W + 2 * B * C * Ho * Wo + 18 * C)
    if name == "nasseg_irdw_fwd":  # x read, the depthwise output written: the expanded map exists in registers only
        B, H, W, K, C, Ho, Wo = a[10], a[11], a[12], a[13], a[14], a[15], a[16]
        return 4 * (B * H * W * K + B * Ho * Wo * C + C * K + 9 * C)
    if name == "nasseg_irdw_bwd":  # x, g, z2 read, ge written
        B, H, W, K, C, Ho, Wo = a[24], a[25], a[26], a[27], a[28], a[29], a[30]
        return 4 * (B * H * W * (K + C) + 2 * B * Ho * Wo * C + C * K + 18 * C)
    if name == "nasseg_irdw_stats":  # one pass over the block's input
        return 4 * a[5] * a[6] * a[7] * a[8]
    if name == "nasseg_conv_wgrad_bn":  # x, g, z read, dz written (BatchNorm backward applied on load)
        B, H, W, K, N = a[20], a[21], a[22], a[23], a[24]
        return 4 * (B * H * W * (K + 3 * N) + N * K)
    if name == "nasseg_dwconv_wgrad_bn":
        B, H, W, C, Ho, Wo, K = a[16], a[17], a[18], a[19], a[20], a[21], a[22]
        return 4 * (B * C * H * W + 3 * B * C * Ho * Wo + C * K * K)
    if name == "nasseg_bn_stats":
        return 4 * a[2] * a[3]
    if name == "nasseg_bn_bwd_reduce":
        return 4 * 2 * a[4] * a[5]
    if name == "nasseg_bn_bwd_apply":
        return 4 * 3 * a[7] * a[8]
    if name == "nasseg_affine_act":
        return 4 * a[5] * (3 if a[3] else 2)
    if name == "nasseg_axpby":
        return 4 * a[5] * (3 if a[1] else 2)
    if name == "nasseg_act_bwd":
        return 4 * 3 * a[3]
    if name == "nasseg_chan_copy":
        return 4 * a[9] * a[10] * (3 if a[6] else 2)
    if name in ("nasseg_pool_fwd", "nasseg_pool_bwd"):
        B, H, W, C, Ho, Wo = a[4], a[5], a[6], a[7], a[8], a[9]
        return 4 * B * C * (H * W + Ho * Wo) + B * C * Ho * Wo  # + uint8 winner index
    if name == "nasseg_cat_src_fwd":  # the input read (at its own size), its slice of the slab written
        B, Hi, Wi, C, Ho, Wo = a[8], a[9], a[10], a[11], a[12], a[13]
        return 4 * B * C * (Hi * Wi + Ho * Wo)
    if name == "nasseg_cat_src_bwd":  # du and slab slices (+ the pending producer's z) read, g written
        B, Ho, Wo, C = a[14], a[15], a[16], a[17]
        # (a pending input of the slab's size: its slab slice is rebuilt from z, not read)
        slab_read = bool(a[8]) and not (a[9] and (a[18], a[19]) == (Ho, Wo))
        return 4 * B * Ho * Wo * C * (2 + int(slab_read) + (1 if a[9] else 0))
    if name == "nasseg_maxpool_bn_fwd":  # z read, pooled map (+ uint8 winner index) written
        B, H, W, C, Ho, Wo = a[5], a[6], a[7], a[8], a[9], a[10]
        return 4 * B * C * (H * W + Ho * Wo) + (B * C * Ho * Wo if a[4] else 0)
    if name == "nasseg_maxpool_bn_bwd":  # dy + index + z read, g written
        B, H, W, C, Ho, Wo = a[7], a[8], a[9], a[10], a[11], a[12]
        return 4 * B * C * (2 * H * W + Ho * Wo) + B * C * Ho * Wo
    if name == "nasseg_bilinear_fwd":
        B, Hi, Wi, C, Ho, Wo = a[4], a[5], a[6], a[7], a[8], a[9]
        return 4 * B * C * (Hi * Wi + Ho * Wo)
    if name == "nasseg_bilinear_bwd":
        B, Hi, Wi, C, Ho, Wo = a[4], a[5], a[6], a[7], a[8], a[9]  # (ws, stream follow)
        return 4 * B * C * (Hi * Wi + Ho * Wo)
    if name == "nasseg_colred":
        n = a[9] * a[10] * a[11]
        return 4 * n * (1 + (1 if a[3] else 0) + (1 if a[5] else 0))
    if name in ("nasseg_ce_fwd", "nasseg_ce_bwd"):
        P, C = (a[3], a[4]) if name == "nasseg_ce_fwd" else (a[5], a[6])
        return P * (4 * C * (2 if name == "nasseg_ce_bwd" else 1) + 8)
    return 0


def algorithmic_flops(name, a):
    """multiply-add FLOPs (2 per MAC) of one launch of a dense-convolution entry point"""
    base = name.replace("nasseg_bf16_", "nasseg_")
    if base == "nasseg_conv_fwd":
        B, Hs, Ws, K, Ho, Wo, N, kh, kw = a[13], a[14], a[15], a[16], a[17], a[18], a[19], a[20], a[21]
        return 2 * B * Ho * Wo * N * K * kh * kw
    return 0


def bench_teacher(args, device):
    """--workload teacher: images/s of the teacher's inference forward and, for the kernel family with the largest
    share of its time, the achieved fp32 MFMA rate against the 157.3 TFLOP/s peak (MI355X_MICROARCH.md): the
    teacher's 256-2048-channel 1x1 / 3x3 convs are the one place on this path where the matrix pipe is the roofline."""
    from nas_segm_amd._lib import LaunchProfiler, lib
    from nas_segm_amd.kd.rf_lw import rf_lw152

    torch.manual_seed(0)
    net = rf_lw152(pretrained=False, num_classes=21).to(device).eval()
    x = torch.randn(args.batch, 3, args.height, args.width, device=device).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(args.warmup):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(x)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        lib.profiler = LaunchProfiler()
        net(x)
        net(x)
        torch.cuda.synchronize()
        records, lib.profiler = lib.profiler.records, None
    fam = {}
    per = len(records) // 2
    for i in range(per):
        name, a = records[i][0], records[i][1]
        ms = min(records[i][2].elapsed_time(records[i][3]), records[per + i][2].elapsed_time(records[per + i][3]))
        ent = fam.setdefault(kernel_family(name, a), [0, 0.0, 0, 0])
        ent[0] += 1
        ent[1] += ms
        ent[2] += algorithmic_flops(name, a)
        ent[3] += algorithmic_bytes(name, a)
    total_ms = sum(v[1] for v in fam.values())
    top, (n, ms, fl, nb) = max(fam.items(), key=lambda kv: kv[1][1])
    peak = 157.3
    tf = fl / 1e12 / (ms / 1e3) if ms > 0 else 0.0
    roof = {"bound": "mfma", "kernel": top, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
            "traffic": None, "algorithmic_flops_per_launch": fl / max(n, 1), "avg_launch_ms": ms / max(n, 1),
            "launches_per_step": n, "share_of_kernel_time": ms / total_ms if total_ms else None,
            "hbm_gbs_of_that_family": nb / 1e9 / (ms / 1e3) if ms > 0 else None,
            "launch_duration": "HIP events on the launch stream, minimum of 2 samples per launch",
            "families": sorted(((k, v[0], round(v[1], 3), round(v[2] / 1e12 / (v[1] / 1e3), 1) if v[1] > 0 else 0.0)
                                for k, v in fam.items()), key=lambda r: -r[2])[:6],
            "nasseg_calls_per_step": per}
    return {"metric": "images/sec (inference forward) KD teacher rf_lw152 {}x{} bs={}".format(
                args.width, args.height, args.batch),
            "value": args.batch * args.steps / elapsed, "unit": "images/sec", "n_gpus": 1, "rccl_ranks": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (randn images; random-init weights of the published architecture)",
            "config": {"workload": "teacher: {} - {}x3x{}x{}".format(WORKLOADS["teacher"][6], args.batch,
                                                                  args.height, args.width),
                       "logits_shape": list(out.shape)},
            "roofline": roof, "cpu_baseline": None}


def build_model(device, workload="headline"):
    from nas_segm_amd.engine import RankParallel, Segmenter
    from nas_segm_amd.nn.encoders import mbv2
    from nas_segm_amd.nn.micro_decoders import MicroDecoder, TemplateDecoder

    kind, genotype, classes = WORKLOADS[workload][:3]
    torch.manual_seed(0)  # random-init weights of the published architecture (no checkpoints offline)
    if kind == "sampled":
        with open(os.path.join(ROOT, "tests", "golden", "controller.json")) as fh:
            samples = json.load(fh)["wacv"]["samples"]
        genotype = samples[int(os.environ.get("RANK", "0")) % len(samples)]["config"]
        enc = mbv2(pretrained=False, return_layers=[1, 2])
        dec = TemplateDecoder(enc.out_sizes, classes, genotype, agg_size=48, repeats=1)
    elif kind == "micro_search":
        enc = mbv2(pretrained=False)
        dec = MicroDecoder(list(enc.out_sizes), classes, genotype, agg_size=48, repeats=1, aux_cell=True)
    elif kind == "template":
        enc = mbv2(pretrained=False, return_layers=[1, 2])
        dec = TemplateDecoder(enc.out_sizes, classes, genotype, agg_size=64, repeats=2)
    else:
        enc = mbv2(pretrained=False)
        dec = MicroDecoder(list(enc.out_sizes), classes, genotype, agg_size=64, repeats=2)
    net = Segmenter(enc, dec).to(device)
    if kind == "sampled":
        return net, net  # independent candidates: nothing is broadcast or all-reduced
    return RankParallel(net), net


def synthetic_batch(batch, height, width, rank, device, classes=NUM_CLASSES):
    g = torch.Generator().manual_seed(rank)
    image = torch.randn(batch, 3, height, width, generator=g)
    mask = torch.randint(0, classes, (batch, height, width), generator=g)
    mask[:, height // 2: height // 2 + 5, :] = 255
    image = image.to(device).contiguous(memory_format=torch.channels_last)
    return image, mask.to(device)


def cpu_baseline(height, width):
    """The oracle (CPU restatement of the reference graph) on the host cores:
    1 warm-up + the median of 5 timed train-mode forward+backward passes of WACV arch0 at
    1x3xHxW (the metric's unit; fewer if 30 s do not suffice), plus the eval forward of
    BASELINE.md section 4."""
    from oracle import engine as oeng
    from oracle import nets as onets

    from nas_segm_amd.engine import Segmenter
    from nas_segm_amd.nn.encoders import mbv2
    from nas_segm_amd.nn.micro_decoders import TemplateDecoder

    # torch's CPU convolutions stop scaling (and oversubscribe badly) far below the
    # 128-256 hardware threads of the GPU box's host: use at most 32 threads and say so
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    enc = mbv2(pretrained=False, return_layers=[1, 2])
    dec = TemplateDecoder(enc.out_sizes, NUM_CLASSES, WACV_ARCH0, agg_size=64, repeats=2)
    net = Segmenter(enc, dec)
    params = {k for k, _ in net.named_parameters()}
    sd = {k: (v.clone().requires_grad_(True) if k in params else v.clone())
          for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, height, width, generator=g)
    t = torch.randint(0, NUM_CLASSES, (1, height, width), generator=g)

    def fwd_bwd():
        for v in sd.values():
            v.grad = None
        out = onets.segmenter(sd, x, "template", WACV_ARCH0, [24, 32], (1, 2), True, repeats=2)
        oeng.train_loss(out, t).backward()

    def fwd_eval():
        with torch.no_grad():
            onets.segmenter(sd, x, "template", WACV_ARCH0, [24, 32], (1, 2), False, repeats=2)

    def timed(fn, budget_s, max_n=5):
        """1 warm-up, then up to max_n timed passes within a wall-clock budget"""
        t_start = time.perf_counter()
        t0 = time.perf_counter()
        fn()
        warm = time.perf_counter() - t0
        ts = []
        while len(ts) < max_n and (time.perf_counter() - t_start) + warm < budget_s:
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        if not ts:
            ts = [warm]
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    med, n_fb = timed(fwd_bwd, 30.0)
    fmed, n_f = timed(fwd_eval, 10.0)
    # BASELINE config 1 / BASELINE.md section 4: the eval forward at 1x3x321x321 (reference tests/test_inference.py path)
    x_full, x = x, torch.randn(1, 3, 321, 321, generator=g)
    c1med, n_c1 = timed(fwd_eval, 5.0, max_n=9)
    x = x_full
    cpu_name = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": 1.0 / med, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "median of {} train-mode fwd+bwd passes of WACV arch0 at 1x3x{}x{} fp32 after 1 warm-up "
                      "({} eval-forward passes for fwd_only), torch CPU threads = cores".format(
                          n_fb, height, width, n_f),
            "fwd_only_images_per_sec": 1.0 / fmed, "cpu_model": cpu_name,
            "config1_321x321_fwd_images_per_sec": 1.0 / c1med,
            "config1_sample": "median of {} eval-forward passes of WACV arch0 at 1x3x321x321 (BASELINE config 1)".format(n_c1)}


def lib_hash():
    """first 16 hex digits of the SHA-256 of the kernel library in use"""
    import hashlib

    from nas_segm_amd._lib import LIB_PATH
    with open(LIB_PATH, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def pmc_summary():
    """The committed rocprofv3 --pmc summary of this command (tools/gpu_pmc.sh: FETCH_SIZE and
    WRITE_SIZE in separate passes) - used ONLY if it was collected with the kernel library that is
    running now (its header carries the library's hash): counters of other kernels would be stale
    numbers.  Returns (None, None) otherwise, else ({family: HBM bytes per launch}, HBM bytes per
    step).  Units in the file: KB; on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced
    reads (MI355X_MICROARCH.md, HBM section) and is doubled here."""
    path = os.path.join(ROOT, "profiles", "pmc_fetch_write_latest.txt")
    fams, step_bytes = {}, None
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None, None
    head = lines[0].split() if lines else []
    if len(head) < 3 or head[0] != "#" or head[1] != "lib" or head[2] != lib_hash():
        return None, None
    try:
        if "step_bytes" in head:
            step_bytes = float(head[head.index("step_bytes") + 1])
        for line in lines[1:]:
            parts = [q.strip() for q in line.split(",")]
            if len(parts) == 4 and parts[0] != "kernel family":
                fams[parts[0]] = (2.0 * float(parts[2]) + float(parts[3])) * 1024.0
    except ValueError:
        return None, None
    return fams, step_bytes


def collect_pmc(passthrough, timeout_s=300):
    """HBM traffic measured in THIS run: two extra, untimed rocprofv3 passes of this command
    (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` separately - they do not fit one pass - with
    `--kernel-trace` only, as MI355X_MICROARCH.md's HBM section prescribes), 3 steps each.
    Returns ({kernel family: HBM bytes per launch}, HBM bytes per step) with FETCH_SIZE doubled
    for gfx950 (units: KB), or (None, None) when rocprofv3 is missing / a pass fails."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None
    out = tempfile.mkdtemp(prefix="nasseg_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    n_steps = 3  # (--steps 2 --warmup 1)
    res = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(out, c), "-o",
                   "run", "--", sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--no-roofline", "--pmc", "0", "--secondary", "0", "--graph", "0"]
            cmd += list(passthrough)  # (host-launched: the same kernels as a replay, and no capture-time trial replays)
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                return None, None
            fam = collections.defaultdict(lambda: [0, 0.0])
            for f in glob.glob(os.path.join(out, c, "**", "*counter_collection*.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != c:
                        continue
                    name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
                    base = re.split(r"[<(]", re.sub(r"^void ", "", name))[0]
                    fam[base][0] += 1
                    fam[base][1] += float(row["Counter_Value"])
            if not fam:
                return None, None
            res[c] = fam
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None, None
    finally:
        shutil.rmtree(out, ignore_errors=True)
    fams, total = {}, 0.0
    for k in set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"]):
        nf, vf = res["FETCH_SIZE"].get(k, [0, 0.0])
        nw, vw = res["WRITE_SIZE"].get(k, [0, 0.0])
        nbytes = (2.0 * vf + vw) * 1024.0
        total += nbytes
        fams[k] = nbytes / max(nf, nw, 1)
    return fams, total / n_steps


def kernel_family(name, a):
    """the __global__ function (rocprofv3's kernel name) an entry-point call dispatches to: the
    roofline is quoted per kernel family so that it can be held against the rocprof summary and
    the PMC traffic of the same name under profiles/"""
    base = name.replace("nasseg_bf16_", "nasseg_")

    def pointwise_kernel(B, Ho, Wo, N, K, mode):
        # (conv_fwd.hip:conv_dispatch) the library's own answer for a 1x1, stride-1 call
        from nas_segm_amd import functional as NF

        return ("conv_fwd_kernel", "conv_pw_kernel", "conv_pwn_kernel")[
            NF.lib.query("nasseg_conv_pointwise_kernel", B, Ho, Wo, N, K, mode)]

    if base == "nasseg_conv_fwd":
        # (conv_fwd.hip:conv_dispatch) 3x3, stride 1, plain forward form: the LDS-tiled kernel where the library says so
        B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil, transposed = a[13:26]
        pro = bool(a[5] or a[6] or a[7])
        if not transposed and not pro and kh == 3 and kw == 3:
            from nas_segm_amd import functional as NF

            # (the library's own answer, as for the pointwise kernels: dilation <= 3, statistics rows included)
            if NF.lib.query("nasseg_conv_fwd_lds3x3", B, Ho, Wo, N, K, kh, kw, stride, pad, dil, int(bool(a[26]))):
                return "conv3x3_lds_kernel"
        if kh == 1 and kw == 1 and stride == 1 and pad == 0 and (Hs, Ws) == (Ho, Wo) and K % 4 == 0 and N % 4 == 0:
            return pointwise_kernel(B, Ho, Wo, N, K, 1)
        return "conv_fwd_kernel"
    if base == "nasseg_conv_bwd_data_bn":
        B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil = a[12:24]
        if kh == 1 and kw == 1 and stride == 1 and pad == 0 and (Hs, Ws) == (Ho, Wo) and K % 4 == 0 and N % 4 == 0:
            return pointwise_kernel(B, Ho, Wo, N, K, 2)
        return "conv_fwd_kernel"
    if base == "nasseg_conv_pw_bwd_bn" and a[21] > 64:
        return "conv_pw_bwd_wide_kernel"
    fam = {"nasseg_sepconv_fwd": "sepconv_fwd_kernel",
           "nasseg_conv_pw_bwd_bn": "conv_pw_bwd_kernel", "nasseg_dwconv_bwd_bn": "dw3x3_bwd_bn_kernel",
           "nasseg_conv_wgrad": "conv_wgrad_kernel", "nasseg_conv_wgrad_bn": "conv_wgrad_bn_kernel",
           "nasseg_dwconv": "dw_fwd_strip", "nasseg_dwconv_bwd_data_bn": "dw_fwd_strip",
           "nasseg_dwconv_wgrad": "dw_wgrad_strip", "nasseg_dwconv_wgrad_bn": "dw_wgrad_strip",
           "nasseg_bn_bwd_apply": "bn_bwd_apply_kernel", "nasseg_affine_act": "affine_act_kernel",
           "nasseg_bn_stats": "colred_kernel", "nasseg_bn_bwd_reduce": "colred_kernel",
           "nasseg_irdw_fwd": "irdw_fwd_kernel", "nasseg_irdw_bwd": "irdw_bwd_kernel",
           "nasseg_irdw_stats": "ir_moments_kernel"}
    return fam.get(base, base)


def roofline_from_profile(records, n_steps=2):
    """records: LaunchProfiler.records of ``n_steps`` identical steps -> (per entry point rows,
    per kernel family groups).  A launch's duration is the MINIMUM over the steps of the bracket
    of HIP events around it: the bracket also contains whatever the stream waited for between the
    two event records (a host that falls behind leaves the queue empty), which a second sample
    of the same launch does not repeat."""
    per = len(records) // n_steps
    steps = [records[i * per:(i + 1) * per] for i in range(n_steps)]
    aligned = len(records) == per * n_steps and all(
        [r[0] for r in st] == [r[0] for r in steps[0]] for st in steps)
    launches = []
    if aligned:
        for i in range(per):
            name, args = steps[0][i][0], steps[0][i][1]
            ms = min(st[i][2].elapsed_time(st[i][3]) for st in steps)
            launches.extend([(name, args, ms)] * n_steps)
    else:  # (the steps did not issue the same launches: keep every sample as it is)
        launches = [(r[0], r[1], r[2].elapsed_time(r[3])) for r in records]
    rows, groups = {}, {}
    for name, args, ms in launches:
        nbytes = algorithmic_bytes(name, args)
        r = rows.setdefault(name, {"kernel": name, "launches": 0, "ms": 0.0, "bytes": 0})
        g = groups.setdefault(kernel_family(name, args), {"ms": 0.0, "bytes": 0, "launches": 0, "entries": set()})
        for ent in (r, g):
            ent["launches"] += 1
            ent["ms"] += ms
            ent["bytes"] += nbytes
        g["entries"].add(name)
    rows = sorted(rows.values(), key=lambda r: -r["ms"])
    for r in rows:
        r["gbs"] = (r["bytes"] / 1e9) / (r["ms"] / 1e3) if r["ms"] > 0 else 0.0
    return rows, groups, launches


def secondary_cvpr321(device, rank, steps=30, warmup=5):
    """BASELINE config 2 beside the headline (north_star: "images/sec on synthetic 321x321 and 2048x1024 batches ...
    as absolute and fraction-of-roofline"): CVPR arch0, 21 classes, 16x3x321x321, fwd + bwd + clip + optimisers, the
    whole step recorded once and replayed (engine.graphed: what train_segmenter does by itself at this size), laid out
    as stages of independent lanes (engine/graph_dag.py).  Timed like the headline: `warmup` untimed steps, `steps`
    timed ones between device synchronisations.  The roofline figures come from two host-launched steps with HIP
    events around every entry point (the dominant kernel family's algorithmic bytes / its launch time)."""
    from nas_segm_amd._lib import LaunchProfiler, lib
    from nas_segm_amd.engine.graphed import GraphedSegmenterStep
    from nas_segm_amd.engine.trainer import segmenter_step

    wl = WORKLOADS["cvpr321"]
    segmenter, net = build_model(device, "cvpr321")
    segmenter.train()
    optim_enc = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    optim_dec = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
    image, mask = synthetic_batch(wl[3], wl[4], wl[5], rank, device, wl[2])
    graphed = GraphedSegmenterStep(segmenter, image, mask, optim_enc, optim_dec, 255, 3.0, 3.0, -1,
                                   capture_optimisers=True)
    for _ in range(warmup):
        loss = graphed.step(image, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = graphed.step(image, mask)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lay = graphed.layout or {}
    out = {"workload": "cvpr321: {} - MobileNetV2 encoder + MicroDecoder, {}x3x{}x{}, train_segmenter step".format(
               wl[6], wl[3], wl[4], wl[5]),
           "value": wl[3] * steps / elapsed, "unit": "images/sec", "ms_per_step": 1e3 * elapsed / steps,
           "steps": steps, "warmup": warmup, "dtype": "f32", "loss": float(loss),
           "launch": "hipGraph(whole step), {}".format(
               "{} lanes, {} line graphs, {} forks per step".format(lay.get("lanes"), lay.get("parts"), lay.get("forks"))
               if graphed.plan is not None else "one line"),
           "line_ms_per_step": lay.get("line_ms"), "roofline": None}
    lib.profiler = LaunchProfiler()
    try:
        for _ in range(2):
            segmenter_step(segmenter, image, mask, optim_enc, optim_dec, 255, 3.0, 3.0, -1)
        torch.cuda.synchronize()
        rows, groups, launches = roofline_from_profile(lib.profiler.records)
    finally:
        lib.profiler = None
    if rows:
        name, top = max(groups.items(), key=lambda kv: kv[1]["ms"])
        gbs = top["bytes"] / 1e9 / (top["ms"] / 1e3)
        total_ms = sum(r["ms"] for r in rows)
        out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": gbs / HBM_PEAK_GBS, "calls_per_step": len(launches) / 2,
                           "tiny_launches_per_step": sum(1 for _, _, ms in launches if ms < 0.010) / 2,
                           "tiny_ms_per_step": sum(ms for _, _, ms in launches if ms < 0.010) / 2,
                           "share_of_kernel_time": top["ms"] / total_ms,
                           "algorithmic_bytes_per_step": sum(r["bytes"] for r in rows) / 2,
                           # the whole step's algorithmic bytes over the replayed step time
                           "step_algorithmic_gbs": sum(r["bytes"] for r in rows) / 2 / 1e9 / (elapsed / steps)}
    return out


def _free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE in the environment): re-run this
    command as N ranks of one node - one process per GPU, rendezvous on 127.0.0.1 - through
    torch.distributed.run, exactly the line the module docstring shows.  Rank 0's stdout (the one
    JSON line) passes through; the exit status is the launcher's."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")  # (torchrun would set 1: the optimisers' host side is threaded)
    if "--same-device" in argv:
        # debug mode, several ranks on ONE device: a replayed hipGraph fans its branches out over the
        # process's hardware queues (4 by default); two processes doing that on one GPU oversubscribe its
        # queues and stall for seconds at a time (profiles/r03_dp_graph_stall_hwqueues.txt: 5-58 s stalls
        # with 4 queues per process, none with 1 or 2).  One process per GPU - the real layout - has the
        # device's queues to itself.
        env.setdefault("GPU_MAX_HW_QUEUES", "2")
    return subprocess.call(cmd, env=env)


def verified_world(device, backend):
    """the number of ranks that REALLY take part in collectives: every rank contributes a one to an
    all-reduce on the device the step runs on (RCCL when backend is nccl) - reported as
    ``rccl_ranks`` so that a line printed by a run that silently fell back to one rank cannot
    pass for an N-GPU measurement"""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    one = torch.ones(1, device=device if backend == "nccl" or device.type == "cuda" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    n = int(round(float(one.item())))
    if n != dist.get_world_size():
        raise SystemExit("bench.py: all-reduce of ones gave {} on a world of {}".format(n, dist.get_world_size()))
    return n


def launch_selftest(args, world, rank):
    """--launch-selftest: the launcher, the rendezvous and one real collective WITHOUT a GPU (gloo
    on the CPU) - what tests/test_distributed_cpu.py runs here, where there is no device."""
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = verified_world(torch.device("cpu"), "gloo")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "rccl_ranks": n, "requested": args.gpus,
                          "backend": "gloo"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS),
                    help="headline = BASELINE.json metric (default); the others are informational")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (0 = the workload's default)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="debug: all ranks share GPU 0 (with --backend gloo on a 1-GPU box)")
    ap.add_argument("--graph", type=int, default=-1, choices=(-1, 0, 1, 2),
                    help="-1 (default) = what the engine does by itself (engine.graphed.auto_graph: train_segmenter "
                         "replays forward+loss+backward of steps up to AUTO_GRAPH_MAX_PIXELS image pixels per rank, "
                         "laid out in lanes - mode 1 - and launches larger ones from the host - mode 0); 0 = launch "
                         "every kernel from the host; 1 = replay forward+loss+backward from a hipGraph (all-reduce, "
                         "clip, optimisers outside); 2 = whole step in the graph (1 GPU)")
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16"),
                    help="storage type of activations and their gradients (arithmetic, statistics, "
                         "parameters and parameter gradients are fp32 either way); the BASELINE metric is f32")
    ap.add_argument("--fused-optim", type=int, default=0, choices=(0, 1),
                    help="torch.optim fused=True implementations of SGD / Adam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", type=int, default=-1, choices=(-1, 0, 1),
                    help="also time BASELINE config 2 (CVPR arch0 16x3x321x321, replayed) after the headline and "
                         "report it as \"secondary\": 1 = yes, 0 = no, -1 = yes for the default 1-GPU headline run")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--shapes", type=int, default=0,
                    help="with --breakdown: also print the N most expensive (entry point, shape) rows")
    ap.add_argument("--pmc", type=int, default=-1, choices=(-1, 0, 1),
                    help="HBM traffic of the step from rocprofv3 PMC counters, collected in two extra untimed "
                         "passes of this command: 1 = yes, 0 = no (the committed profiles/ summary is used while "
                         "it matches the library), -1 = yes for the default 1-GPU headline run")
    ap.add_argument("--step-times", action="store_true",
                    help="debug: every rank prints the host time of each timed step to stderr")
    ap.add_argument("--sync-each-step", action="store_true",
                    help="debug: read the loss on the host after every step (the engine's loss.item())")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="only launch the ranks, rendezvous and all-reduce on the CPU (gloo): no GPU needed")
    args = ap.parse_args()
    args.graph_flag_given = any(a == "--graph" or a.startswith("--graph=") for a in sys.argv[1:])

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started bare: become the launcher of N ranks (one per GPU) and pass rank 0's line through
        if not args.launch_selftest and not args.same_device:
            n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if n_dev < args.gpus:
                raise SystemExit("bench.py: --gpus {} but {} HIP device(s) visible".format(args.gpus, n_dev))
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus {} but WORLD_SIZE={} (one rank per GPU: launch with "
                         "--nproc-per-node {})".format(args.gpus, world, args.gpus))
    if args.launch_selftest:
        return launch_selftest(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    if args.same_device:
        local_rank = 0
        if args.backend == "nccl":
            # (RCCL refuses two ranks on one device - "Duplicate GPU detected": the debug mode stages through gloo)
            if rank == 0:
                sys.stderr.write("bench.py: --same-device: RCCL needs a device per rank, using --backend gloo\n")
            args.backend = "gloo"
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank {} has no GPU of its own ({} visible)".format(
            local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        # nccl = RCCL over xGMI; binding the group to its device creates the communicator here,
        # not lazily inside the first collective
        extra = dict(device_id=device) if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **extra)
    rccl_ranks = verified_world(device, args.backend)

    import nas_segm_amd  # noqa: F401
    from nas_segm_amd._lib import LaunchProfiler, lib
    from nas_segm_amd.engine.trainer import segmenter_step

    wl = WORKLOADS[args.workload]
    args.batch = args.batch or wl[3]
    args.height = args.height or wl[4]
    args.width = args.width or wl[5]
    if args.workload == "teacher":
        if world > 1:
            raise SystemExit("bench.py: the teacher workload is a one-GPU inference measurement")
        print(json.dumps(bench_teacher(args, device)))
        return
    segmenter, net = build_model(device, args.workload)
    segmenter.train()
    # default_args.py:57-66: SGD(lr 1e-3, mom 0.9, wd 1e-5) encoder, Adam(lr 3e-3, wd 1e-5) decoder
    # (torch.optim is the reference's optimiser too - SURVEY section 8a O1; `fused` selects its
    # single-kernel multi-tensor implementation instead of the foreach one)
    fused = dict(fused=True) if args.fused_optim else {}
    # (plain objects, as src/utils/solvers.py makes them: the engine steps those with nasseg_optim_step, also inside a
    # hipGraph; torch's own implementations - NASSEG_NATIVE_OPTIM=0 or --fused-optim 1 - need capturable=True there)
    torch_steps = bool(args.fused_optim) or os.environ.get("NASSEG_NATIVE_OPTIM", "1") == "0"
    optim_enc = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5, **fused)
    optim_dec = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5,
                                 capturable=args.graph == 2 and torch_steps, **fused)
    image, mask = synthetic_batch(args.batch, args.height, args.width, rank, device, wl[2])
    if args.dtype == "bf16":
        image = image.to(torch.bfloat16)

    def eager_step():
        return segmenter_step(segmenter, image, mask, optim_enc, optim_dec, 255, 3.0, 3.0, -1)

    if args.workload == "depth480":
        from nas_segm_amd import functional as NF
        from nas_segm_amd.engine.trainer import _clip_and_step, _zero_grads

        with torch.no_grad():
            probe = segmenter(image)
            probe = probe[0] if isinstance(probe, tuple) else probe
        depth = (torch.rand(probe.shape, device=device).contiguous(memory_format=torch.channels_last)
                 * 10.0).to(image.dtype)
        del probe

        def eager_step():  # noqa: F811 - regression head: berHu instead of softmax/NLL
            out = segmenter(image)
            out = out[0] if isinstance(out, tuple) else out
            loss = NF.berhu_loss(out, depth)
            _zero_grads(segmenter, (optim_enc, optim_dec))
            loss.backward()
            segmenter.sync_gradients()
            _clip_and_step([(list(net.encoder.parameters()), 3.0, optim_enc),
                            (list(net.decoder.parameters()), 3.0, optim_dec)])
            return loss

    if args.workload == "task0":
        # populate the (per-rank shard of the) feature cache with the encoder, then time decoder-only
        # steps on it: populate_task0 / make_task0_step of the engine, batches drawn like train_task0
        import numpy as np

        from nas_segm_amd.engine.trainer import make_task0_step, populate_task0

        n_cache = 16 * args.batch
        g = torch.Generator().manual_seed(100 + rank)
        loader = [{"image": torch.randn(args.batch, 3, args.height, args.width, generator=g),
                   "mask": torch.randint(0, wl[2], (args.batch, args.height, args.width), generator=g)}
                  for _ in range(n_cache // args.batch)]
        Xy = populate_task0(segmenter, loader, None, n_cache, do_kd=False)
        assert not isinstance(Xy, int), "populate_task0 failed"
        del loader
        segmenter.train()
        rng = np.random.RandomState(rank)
        env_graph = os.environ.get("NASSEG_GRAPH")  # (the user's setting is put back below)
        if args.graph_flag_given:
            os.environ["NASSEG_GRAPH"] = {0: "0", 1: "1", 2: "1"}[args.graph]
        task0_step = make_task0_step(Xy, segmenter, optim_dec, args.batch, 255, 3.0, 0.15)
        task0_eager = task0_step
        if getattr(task0_step, "__self__", None) is not None:  # (a stepper's bound method: replayed)
            os.environ["NASSEG_GRAPH"] = "0"
            task0_eager = make_task0_step(Xy, segmenter, optim_dec, args.batch, 255, 3.0, 0.15)
        if env_graph is None:
            os.environ.pop("NASSEG_GRAPH", None)
        else:
            os.environ["NASSEG_GRAPH"] = env_graph

        def eager_step():  # noqa: F811
            return task0_eager(rng.permutation(n_cache)[:args.batch])

    graph_layout = None
    graph_auto = args.graph < 0
    if graph_auto and args.workload != "task0":
        from nas_segm_amd.engine.graphed import auto_graph
        args.graph = 1 if auto_graph(segmenter, args.batch * args.height * args.width) else 0
    step = eager_step
    if args.workload == "task0":
        args.graph = 2 if getattr(task0_step, "__self__", None) is not None else 0

        def step():  # noqa: F811
            return task0_step(rng.permutation(n_cache)[:args.batch])

        stepper0 = getattr(task0_step, "__self__", None)
        if stepper0 is not None and getattr(stepper0, "plan", None) is not None:
            graph_layout = stepper0.layout
            if rank == 0:
                sys.stderr.write("graph layout: {}\n".format(stepper0.layout))
    elif args.graph:
        from nas_segm_amd.engine.graphed import GraphedSegmenterStep
        t_capture = time.perf_counter()
        if args.workload == "depth480":
            graphed = GraphedSegmenterStep(segmenter, image, depth, optim_enc, optim_dec, 255, 3.0, 3.0, -1,
                                           capture_optimisers=args.graph == 2, loss_fn=NF.berhu_loss)
            mask = depth
        else:
            graphed = GraphedSegmenterStep(segmenter, image, mask, optim_enc, optim_dec, 255, 3.0, 3.0, -1,
                                           capture_optimisers=args.graph == 2)

        def step():
            return graphed.step(image, mask)

        graph_layout = getattr(graphed, "layout", None) if getattr(graphed, "plan", None) is not None else None
        if rank == 0 and getattr(graphed, "layout", None):
            sys.stderr.write("graph layout ({:.2f} s to warm up, record, lay out and time the layouts): {}\n".format(
                time.perf_counter() - t_capture, graphed.layout))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        loss = step()
        if args.sync_each_step:
            float(loss)
        if args.step_times:
            stamps.append(time.perf_counter())
    fence()
    elapsed = time.perf_counter() - t0
    if args.step_times:
        sys.stderr.write("rank {} step ms (host, {}): {}  | fence +{:.1f}\n".format(
            rank, "synced" if args.sync_each_step else "not synced",
            " ".join("{:.1f}".format(1e3 * (b - a)) for a, b in zip(stamps, stamps[1:])),
            1e3 * (t0 + elapsed - stamps[-1])))
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_value = float(loss.item())
    peak_gib = torch.cuda.max_memory_allocated(device) / 2.0 ** 30  # (warm-up + timed steps; before the roofline passes)
    reward = None
    if args.workload == "search713":
        # the candidate's reward on one batch: fused upsample -> argmax -> confusion matrix on the GPU,
        # IoU arithmetic of the Cython module on the host (engine/inference.py)
        from nas_segm_amd.engine.inference import validate
        reward = float(validate(segmenter, [{"image": image, "mask": mask.to(torch.uint8)}], 0, 0,
                                num_classes=wl[2], print_every=10 ** 9, omit_classes=[]))
        segmenter.train()

    roof = None
    rows = []
    if not args.no_roofline:
        # two extra untimed steps with per-launch HIP events; every rank takes them (the step
        # contains the gradient all-reduce), only rank 0 records
        if rank == 0:
            lib.profiler = LaunchProfiler()
        eager_step()
        eager_step()
        torch.cuda.synchronize()
        records = lib.profiler.records if rank == 0 else []
        rows, groups, launches = roofline_from_profile(records)
        lib.profiler = None
        if rows:
            total_ms = sum(r["ms"] for r in rows)
            # depthwise forward + backward launches (plain, with the fused BN-backward sums, and the
            # one-kernel backward between two BatchNorms)
            dwr = [r for r in rows if r["kernel"].replace("nasseg_bf16_", "nasseg_")
                   in ("nasseg_dwconv", "nasseg_dwconv_bwd_data_bn", "nasseg_dwconv_bwd_bn")]
            dw = ([{"gbs": sum(r["bytes"] for r in dwr) / 1e9 / (sum(r["ms"] for r in dwr) / 1e3)}]
                  if dwr and sum(r["ms"] for r in dwr) > 0 else [])
            name, top = max(groups.items(), key=lambda kv: kv[1]["ms"])
            gbs = top["bytes"] / 1e9 / (top["ms"] / 1e3)
            # HBM traffic from the PMC counters: only from a summary collected with THIS build of the
            # kernels and this command (fp32 headline), else null
            fams = step_bytes = None
            pmc_source = None
            if args.pmc == 1 or (args.pmc == -1 and os.environ.get("NASSEG_BENCH_PMC", "1") != "0"
                                 and world == 1 and args.workload == "headline" and args.batch == wl[3]
                                 and (graph_auto or not args.graph)):
                passthrough = ["--workload", args.workload, "--dtype", args.dtype, "--batch", str(args.batch),
                               "--height", str(args.height), "--width", str(args.width)]
                torch.cuda.synchronize()
                fams, step_bytes = collect_pmc(passthrough) if world == 1 else (None, None)
                pmc_source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run" if fams else None
            if fams is None and args.dtype == "f32" and args.workload == "headline" and args.batch == wl[3]:
                fams, step_bytes = pmc_summary()
                pmc_source = "profiles/pmc_fetch_write_latest.txt (same library hash)" if fams else None
            unfused = {"headline": 24.98e9, "arch1": 35.12e9}.get(args.workload)
            roof = {"bound": "hbm", "kernel": name, "entry_points": sorted(top["entries"]),
                    "top5": [{"kernel": r["kernel"], "gbs": round(r["gbs"], 1),
                              "frac": round(r["gbs"] / HBM_PEAK_GBS, 3),
                              "share": round(r["ms"] / total_ms, 3)} for r in rows[:5]],
                    "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "frac_of_achievable": gbs / HBM_ACHIEVABLE_GBS, "achievable": HBM_ACHIEVABLE_GBS,
                    "launch_duration": "HIP events on the launch stream, minimum of 2 samples per launch",
                    "traffic": fams.get(name) if fams else None, "traffic_source": pmc_source,
                    "algorithmic_bytes_per_launch": top["bytes"] / top["launches"],
                    "avg_launch_ms": top["ms"] / top["launches"], "launches_per_step": top["launches"] / 2,
                    "share_of_kernel_time": top["ms"] / total_ms,
                    "nasseg_calls_per_step": len(launches) / 2,
                    # launches that cost their dispatch whatever they move (finalisers, row sums, small maps)
                    "tiny_launches_per_step": sum(1 for _, _, ms in launches if ms < 0.010) / 2,
                    "tiny_launch_ms_per_step": sum(ms for _, _, ms in launches if ms < 0.010) / 2,
                    "step_traffic": step_bytes,
                    "step_traffic_vs_unfused": (step_bytes / (unfused * args.batch)
                                                if step_bytes and unfused else None),
                    "step_hbm_gbs": (step_bytes / 1e9 / (elapsed / args.steps) if step_bytes else None),
                    "pmc_lib": lib_hash() if fams else None,
                    "depthwise_gbs": dw[0]["gbs"] if dw else None,
                    "depthwise_frac": dw[0]["gbs"] / HBM_PEAK_GBS if dw else None}
        if args.breakdown:
            for r in rows:
                sys.stderr.write("{kernel:28s} n={launches:5d} {ms:9.3f} ms {gbs:9.1f} GB/s\n".format(**r))
            for fam_name, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
                sys.stderr.write("family {:26s} n={:5d} {:9.3f} ms {:9.1f} GB/s\n".format(
                    fam_name, g["launches"], g["ms"], g["bytes"] / 1e6 / g["ms"] if g["ms"] > 0 else 0.0))
            if args.shapes:
                by_shape = {}
                for lname, a, t in launches:
                    key = (lname,) + tuple(v for v in a if isinstance(v, (int, float)) and abs(v) < (1 << 31))
                    ent = by_shape.setdefault(key, [0, 0.0, 0])
                    ent[0] += 1
                    ent[1] += t
                    ent[2] += algorithmic_bytes(lname, a)
                top_shapes = sorted(by_shape.items(), key=lambda kv: -kv[1][1])[: args.shapes]
                for key, (n, ms, nb) in top_shapes:
                    sys.stderr.write("{:9.3f} ms n={:3d} {:8.1f} GB/s  {} {}\n".format(
                        ms, n, nb / 1e6 / ms if ms > 0 else 0.0, key[0], list(key[1:])))
    if world > 1:
        fence()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "headline":
        cpu = cpu_baseline(args.height, args.width)
        # the GPU's inference forward at the same 1x3xHxW, beside the CPU's fwd_only figure
        segmenter.eval()
        with torch.no_grad():
            one = image[:1].contiguous(memory_format=torch.channels_last)
            for _ in range(3):
                segmenter(one)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                segmenter(one)
            torch.cuda.synchronize()
            cpu["gpu_fwd_only_images_per_sec"] = 10.0 / (time.perf_counter() - t0)
            # ... and BASELINE config 1's shape, 1x3x321x321 (host-launch-bound at this size)
            small = torch.randn(1, 3, 321, 321, device=device).to(image.dtype).contiguous(
                memory_format=torch.channels_last)
            for _ in range(3):
                segmenter(small)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                segmenter(small)
            torch.cuda.synchronize()
            cpu["config1_321x321_gpu_fwd_images_per_sec"] = 20.0 / (time.perf_counter() - t0)
        segmenter.train()

    secondary = None
    if rank == 0 and (args.secondary == 1 or (args.secondary == -1 and world == 1 and args.workload == "headline"
                                              and args.batch == wl[3] and (graph_auto or not args.graph)
                                              and args.dtype == "f32")):
        # (after everything the headline needs: its timed region, roofline passes and CPU baseline are done)
        del segmenter, net, optim_enc, optim_dec
        secondary = secondary_cvpr321(device, rank)

    if rank == 0:
        imgs = args.batch * world * args.steps
        out = {
            "metric": "images/sec ({}) {} {}x{} bs={}/GPU".format(
                "decoder-only fwd+bwd+optimizer step on cached encoder features" if args.workload == "task0"
                else "fwd+bwd+optimizer step",
                "WACV arch0" if args.workload == "headline" else args.workload, args.width, args.height,
                args.batch),
            "value": imgs / elapsed, "unit": "images/sec", "n_gpus": world, "rccl_ranks": rccl_ranks,
            "collective_backend": (args.backend if world > 1 else None), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 storage / f32 arithmetic",
            "data": "synthetic (randn images, randint labels with a 255 band; random-init weights)",
            "config": {"workload": "{}: {} - MobileNetV2 encoder + searched decoder, "
                                   "{}x3x{}x{} per GPU, train_segmenter step".format(
                                       args.workload, wl[6], args.batch, args.height, args.width),
                       "global_batch": args.batch * world, "parallelism": "dp{}".format(world),
                       "loss": loss_value, "reward": reward,
                       "max_memory_allocated_gib": round(peak_gib, 2),
                       "launch": ("host", "hipGraph(fwd+loss+bwd)", "hipGraph(whole step)" if args.workload != "task0"
                                  else "hipGraph(gather+fwd+loss+bwd), chosen by engine.graphed.auto_graph")[args.graph]
                       + (", chosen by engine.graphed.auto_graph" if graph_auto and args.workload != "task0" else "")
                       + (", {} lanes, {} line graphs, {} forks per step (line {} ms)".format(
                           graph_layout.get("lanes"), graph_layout.get("parts"), graph_layout.get("forks"),
                           graph_layout.get("line_ms")) if graph_layout and graph_layout.get("parts") else "")},
            "roofline": roof, "cpu_baseline": cpu, "secondary": secondary,
        }
        if cpu:
            out["gpu_over_cpu"] = out["value"] / cpu["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
