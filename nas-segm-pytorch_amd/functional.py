"""Differentiable host-side wrappers of the nasseg HIP kernels.

Each function here is one ``torch.autograd.Function`` whose forward and backward
are calls into libnasseg_hip.so (see include/nasseg.h).  PyTorch supplies device
memory (caching allocator), the current HIP stream and the autograd tape only;
no ATen compute op runs on the hot path.  Tensors keep the reference's NCHW
*shape* but live in ``torch.channels_last`` memory, which is the NHWC layout the
kernels address directly.
"""
import ctypes
import os
import weakref

import torch

from ._lib import LaunchProfiler, NassegError, current_stream, lib, ptr, require_device

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
RED_SUM, RED_SUMSQ, RED_DOT2, RED_DOT1 = 0, 1, 3, 4


# ---------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------
_BF16_PREFIX = "nasseg_bf16_"


def _k(name, t):
    """Entry point for activations stored like ``t``: nasseg_<op> (fp32) or its bfloat16 twin
    nasseg_bf16_<op> (same arguments; include/nasseg.h)."""
    if t.dtype == torch.float32:
        return name
    return _BF16_PREFIX + name[7:]


def _to_dtype(t, dtype):
    """t.to(dtype) between fp32 and bf16 as a nasseg launch (nasseg_to_bf16 / nasseg_from_bf16: the same rounding as
    torch's; an ATen kernel inside a recorded step would be a barrier for engine/graph_dag.py); anything else:
    torch's .to()"""
    if t.dtype == dtype:
        return t
    t = t.contiguous()
    if t.is_cuda and t.numel() and (t.dtype, dtype) in ((torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32)):
        y = torch.empty_like(t, dtype=dtype)
        lib.call("nasseg_to_bf16" if dtype == torch.bfloat16 else "nasseg_from_bf16", ptr(t), ptr(y), t.numel(),
                 current_stream())
        return y
    return t.to(dtype)


def _cl(x):
    """NHWC-contiguous view/copy of a 4-D NCHW-shaped activation (fp32 or bf16 storage)."""
    require_device(x)
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise NassegError("nasseg activations are fp32 or bf16 (got {})".format(x.dtype))
    if x.dim() != 4:
        raise NassegError("expected a 4-D activation, got shape {}".format(tuple(x.shape)))
    return x.contiguous(memory_format=torch.channels_last)


def _new(like, B, C, H, W):
    return torch.empty((B, C, H, W), device=like.device, dtype=like.dtype,
                       memory_format=torch.channels_last)


def _ws(like, n):
    return torch.empty((max(int(n), 1),), device=like.device, dtype=torch.float32)


def _vec(like, n):
    return torch.empty((int(n),), device=like.device, dtype=torch.float32)


# BatchNorm backward whose apply kernel adds up the partial rows of its sums itself (nasseg_bn_bwd_apply_rows): on the
# small maps of the CVPR cells the sums come as 8 - 128 rows (the first stage of the reduction over gradient and conv
# output, or the statistics rows of a fused backward-data kernel), and the launch that summed them - colred_finalize /
# rows_group_sum, ~5 us on the dependency chain of every BatchNorm backward, 81 + 60 of the ~1000 kernels of a replayed
# CVPR 321x321 step - is replaced by one more round trip to L2 at the head of the apply kernel.  Rows beyond
# nasseg_bn_bwd_apply_rows_max_bytes() keep the summing launch.  NASSEG_APPLY_ROWS=0 restores it everywhere (A/B).
APPLY_ROWS = os.environ.get("NASSEG_APPLY_ROWS", "1") != "0"


def _rows_small(nrows, C):
    """few enough rows of 2 C floats for every workgroup of the apply kernel to add them up itself"""
    return (APPLY_ROWS and C % 4 == 0 and C <= 1024 and 0 < nrows
            and nrows * 2 * C * 4 <= lib.query("nasseg_bn_bwd_apply_rows_max_bytes"))


def _bn_bwd_apply(g, z, scale, shift, mean, invstd, sums, M, C, training, act, dz, rows=None):
    """dz of a BatchNorm (+ activation) backward from the summed sums, or - rows = (buffer, count) - from their
    partial rows (``sums`` then RECEIVES the totals: they are the BatchNorm's parameter gradients)"""
    if rows is not None:
        lib.call(_k("nasseg_bn_bwd_apply_rows", g), ptr(g), ptr(z), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                 ptr(rows[0]), rows[1], ptr(sums), M, C, int(training), act, ptr(dz), current_stream())
    else:
        lib.call(_k("nasseg_bn_bwd_apply", g), ptr(g), ptr(z), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                 ptr(sums), M, C, int(training), act, ptr(dz), current_stream())
    return dz


def _bn_bwd_reduce(g, z, scale, shift, mean, invstd, act, sums, M, C, for_apply):
    """{sum g', sum g' * xhat} of a BatchNorm backward over the M pixels of g, z: into ``sums`` (returns None), or -
    when the caller goes on to _bn_bwd_apply (for_apply) and the first stage leaves few rows - only the rows, which
    are returned as (buffer, count) for the apply kernel to add up"""
    ws = _ws(z, lib.query("nasseg_colred_workspace", 1, M, C))
    if for_apply:
        nrows = lib.query("nasseg_colred_rows", 1, M, C)
        if _rows_small(nrows, C):
            lib.call(_k("nasseg_bn_bwd_reduce_rows", g), ptr(g), C, ptr(z), C, M, C, ptr(scale), ptr(shift), ptr(mean),
                     ptr(invstd), act, ptr(ws), current_stream())
            return ws, nrows
    lib.call(_k("nasseg_bn_bwd_reduce", g), ptr(g), C, ptr(z), C, M, C, ptr(scale), ptr(shift), ptr(mean),
             ptr(invstd), act, ptr(sums), ptr(ws), current_stream())
    return None


# ---------------------------------------------------------------------------
# deferred finalisation of weight gradients
# ---------------------------------------------------------------------------
class deferred_wgrad(object):
    """``with F.deferred_wgrad(): loss.backward()``

    A weight gradient is a two-stage reduction (per-slab partial sums, then their sum) and only
    the optimiser reads it.  Inside this context the backward-weight calls stop after stage one
    and the second stages of ALL layers run when the context exits - one launch per 16 layers
    (nasseg_wgrad_finalize_many) instead of one small launch per layer on the backward chain.
    Opt-in, because between backward and the exit the gradient tensors are allocated but not yet
    written: the caller must have cleared the gradients and every weight must be used once in
    the graph (autograd then just adopts the new tensor; an accumulation - or the copy it makes
    of a tensor somebody else still references, which is why only the ADDRESS is queued here -
    would read it too early), and must not touch ``param.grad`` before the exit.
    Same arithmetic in the same order: results are bit-identical."""

    active = False
    pending = []   # (partials, address of the gradient tensor, finalisation dims)
    grouped = {}   # (entry point, dtype) -> [(tensors the kernels read, launch arguments)]: first
    #                stages of small layers, launched side by side at the exit
    #                (nasseg_conv_wgrad_many / nasseg_dwconv_wgrad_many)
    side_ok = False  # first stages may go to a second stream (WGRAD_STREAM; never while a hipGraph is captured)
    side_used = {}   # device -> that stream, once a launch went there

    def __init__(self, enabled=True, params=None, second_stream=True):
        """params: the parameters being trained; when given, the exit verifies that every
        deferred gradient is the tensor autograd adopted as some ``param.grad`` (it would be a
        copy - of unwritten memory - had a condition above been violated) and fails loudly.
        second_stream: first stages may be launched on a second stream (WGRAD_STREAM below)."""
        self.enabled = bool(enabled)
        self.params = params
        self.second_stream = bool(second_stream)

    def __enter__(self):
        self.prev = deferred_wgrad.active, deferred_wgrad.side_ok
        deferred_wgrad.active = self.enabled
        deferred_wgrad.side_ok = bool(self.enabled and self.second_stream and WGRAD_STREAM and torch.cuda.is_available()
                                      and not torch.cuda.is_current_stream_capturing())
        return self

    @staticmethod
    def _launch_group(entry, dtype, calls, stream):
        flat = [v for c in calls for v in c[1]]
        table = (ctypes.c_int64 * len(flat))(*flat)
        name = entry if dtype == torch.float32 else entry.replace("nasseg_", "nasseg_bf16_", 1)
        if lib.recorder is not None:  # (a step being recorded: c[0] = (input, dz, prologue scale, shift, partials))
            lib.recorder.annotate(reads=[t for c in calls for t in c[0][:4]], writes=[c[0][4] for c in calls])
        lib.call(name, len(calls), table, stream)

    @staticmethod
    def _finalize(todo, stream):
        """second stages of the layers in ``todo`` (partials, address of the gradient, dims) on ``stream``"""
        n = len(todo)
        parts = (ctypes.c_void_p * n)(*[ptr(ws) for ws, _, _ in todo])
        outs = (ctypes.c_void_p * n)(*[addr for _, addr, _ in todo])
        dims = (ctypes.c_int * (5 * n))()
        for j, (_, _, d) in enumerate(todo):
            dims[5 * j:5 * j + 5] = d
        if lib.recorder is not None:  # (dims: rows, taps, N, K, flat - the gradient tensor has taps * N * K floats)
            lib.recorder.annotate(reads=[ws for ws, _, _ in todo],
                                  writes=[(addr, 4 * d[1] * d[2] * d[3]) for _, addr, d in todo])
        lib.call("nasseg_wgrad_finalize_many", n, parts, outs, dims, stream)

    def __exit__(self, exc_type, exc, tb):
        deferred_wgrad.active, deferred_wgrad.side_ok = self.prev
        todo, deferred_wgrad.pending = deferred_wgrad.pending, []
        groups, deferred_wgrad.grouped = deferred_wgrad.grouped, {}
        sides, deferred_wgrad.side_used = deferred_wgrad.side_used, {}
        if exc_type is None:
            for (entry, dtype), calls in groups.items():
                self._launch_group(entry, dtype, calls, current_stream())
                todo.extend(c[2] for c in calls)
        # the second stream's launches fill the partial sums finalised below: join (also after an exception)
        for dev, side in sides.items():
            torch.cuda.current_stream(dev).wait_stream(side)
        if exc_type is None:
            if todo:
                self._finalize(todo, current_stream())
            if self.params is not None and todo:
                adopted = set(p.grad.data_ptr() for p in self.params if p.grad is not None)
                if any(addr not in adopted for _, addr, _ in todo):
                    raise NassegError("deferred_wgrad: autograd copied a weight gradient before it was "
                                      "finalised (gradients not cleared, or a weight used twice?)")
        return False


# largest x + dy footprint (bytes) of a layer whose backward-weight launch is grouped
_GROUP_WGRAD_BYTES = 48 << 20

# Weight gradients on a second stream.  Inside deferred_wgrad nothing on the backward chain waits for a weight
# gradient: its first stage reads the layer's input and the gradient w.r.t. its output, writes partial sums, and the
# second stage runs at the exit.  Launched on the stream of the chain, those kernels sit between the backward-data
# kernels - or, the grouped small layers, behind the whole backward - although many of them have too few workgroups
# to fill 256 CUs; on a stream of their own (which first waits for what the chain has launched so far) the GPU runs
# them beside the chain, and the chain's stream waits for that stream once, at the exit.  Bits of NASSEG_WGRAD_STREAM:
# 1 = the launches of large layers, 2 = the grouped small layers, _SIDE_GROUP at a time as they come.  Measured on one
# box (profiles/r05_ab_second_half_same_box.txt): either bit alone is level, both together +0.5 % on the headline
# step and +1.2 - 1.5 % on WACV arch1.  0: everything on the chain's stream (A/B).  Never while a hipGraph is being
# captured (one second stream inside a capture made the replay slower, two crash this runtime: DESIGN_HISTORY.md),
# and only where the caller asks for it (deferred_wgrad(second_stream=...): a step that is launch-bound on the host
# gains nothing from more launches - CVPR 321x321 from the host 750.5 / 745.8).  The second stages stay on the chain's
# stream, after the join: sixteen at a time on the second stream as backward goes measured level (281.1 / 281.4).
# Same kernels on the same data: bit-identical.
WGRAD_STREAM = int(os.environ.get("NASSEG_WGRAD_STREAM", "3"))
_SIDE_STREAMS = {}
_SIDE_GROUP = int(os.environ.get("NASSEG_WGRAD_SIDE_GROUP", "8"))
# (in a step being recorded for lanes: measured slower at 8 and level at 16 - more launches of the grouped kernels -,
#  profiles/r06_ab_lanes.txt; kept as a switch)
_RECORD_GROUP = int(os.environ.get("NASSEG_WGRAD_RECORD_GROUP", "1000000"))


def _wgrad_stream(t, keep):
    """the stream a first-stage weight-gradient launch over ``t`` goes to, after it has been made to wait for the
    current one: the second stream (``keep`` - the tensors the launch touches, or the queued calls of a grouped launch -
    is then recorded on it for the allocator), else the current"""
    if not (deferred_wgrad.side_ok and t.is_cuda and (WGRAD_STREAM & 1 or isinstance(keep, list))):
        return current_stream()
    side = deferred_wgrad.side_used.get(t.device)
    if side is None:
        side = _SIDE_STREAMS.get(t.device)
        if side is None:
            side = _SIDE_STREAMS[t.device] = torch.cuda.Stream(t.device)
            LaunchProfiler.streams[side.cuda_stream] = side
        deferred_wgrad.side_used[t.device] = side
    side.wait_stream(torch.cuda.current_stream(t.device))
    # What the launch reads and writes must not be handed out again before the second stream is through with it -
    # and must not stay allocated until the end of backward either (round 5 kept references until the exit: peak
    # memory of a large step grew towards activations + every dz).  record_stream tells the caching allocator
    # exactly that: the block is reusable once the second stream has passed the point where it was freed.
    for item in (keep if isinstance(keep, tuple) else [x for c in keep for x in c[0]]):
        if item is not None:
            item.record_stream(side)
    return side.cuda_stream


def _group_wgrad(entry, cur, tensors, desc, fin):
    """queue a small layer's first stage (``fin``: what its second stage needs, queued once the first is launched);
    with WGRAD_STREAM & 2, launch the queue on the second stream when it holds _SIDE_GROUP layers (a list as ``keep``
    marks such a launch for _wgrad_stream)"""
    key = (entry, cur.dtype)
    calls = deferred_wgrad.grouped.setdefault(key, [])
    calls.append((tensors, desc, fin))
    if WGRAD_STREAM & 2 and deferred_wgrad.side_ok and cur.is_cuda and len(calls) >= _SIDE_GROUP:
        del deferred_wgrad.grouped[key]
        deferred_wgrad._launch_group(entry, cur.dtype, calls, _wgrad_stream(cur, calls))
        deferred_wgrad.pending.extend(c[2] for c in calls)
    elif lib.recorder is not None and len(calls) >= _RECORD_GROUP:
        # a step being recorded for replay in lanes (engine/graph_dag.py): _SIDE_GROUP layers at a time as backward
        # goes, on the recording stream - the layout then runs such a launch beside the NEXT piece of the backward
        # chain; one launch of all of them behind the chain would be a stage of its own
        del deferred_wgrad.grouped[key]
        deferred_wgrad._launch_group(entry, cur.dtype, calls, current_stream())
        deferred_wgrad.pending.extend(c[2] for c in calls)


def _dw_wgrad(cur, dz, w, psc, psh, pact, geom):
    """Weight gradient of a depthwise conv (geom = B, H, W, C, Ho, Wo, k, stride, pad, dil); see
    _dense_wgrad."""
    B, H, W, C, Ho, Wo, k, stride, pad, dil = geom
    dwt = torch.empty_like(w)
    ws = _ws(cur, lib.query("nasseg_dwconv_wgrad_workspace", B, C, Ho, Wo, k))
    if deferred_wgrad.active and (B * H * W * C + B * Ho * Wo * C) * cur.element_size() <= _GROUP_WGRAD_BYTES:
        desc = (ptr(cur), ptr(dz), ptr(ws), ptr(psc) or 0, ptr(psh) or 0, pact) + tuple(geom)
        _group_wgrad("nasseg_dwconv_wgrad_many", cur, (cur, dz, psc, psh, ws), desc, _fin_entry(ws, dwt, k * k, C, 1, 0))
        return dwt
    st = _wgrad_stream(cur, (cur, dz, psc, psh, ws))  # (before this layer is queued for finalisation)
    lib.call(_k("nasseg_dwconv_wgrad", cur), ptr(cur), ptr(dz), _finish_wgrad(ws, dwt, k * k, C, 1, 0), ptr(ws),
             ptr(psc), ptr(psh), pact, *geom, st)
    return dwt


def _dense_wgrad(cur, dz, w, psc, psh, pact, geom):
    """Weight gradient of a dense conv (geom = B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil),
    ``cur`` read through the prologue (psc, psh, pact).  Immediate, or - inside deferred_wgrad -
    with its second stage deferred and, for small maps, its first stage grouped as well."""
    B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil = geom
    dwt = torch.empty_like(w)
    ws = _ws(cur, lib.query("nasseg_conv_wgrad_workspace", B, Ho, Wo, N, K, kh, kw))
    flat = int(lib.query("nasseg_conv_fwd_pack_mode", K, kh, kw) == 2)
    # (layers that nasseg_conv_wgrad runs on its LDS-tiled 3x3 kernel are not grouped: the grouped launch always
    #  uses the generic kernel, and deferred / immediate finalisation must give the same bits)
    if (deferred_wgrad.active and (B * Hs * Ws * K + B * Ho * Wo * N) * cur.element_size() <= _GROUP_WGRAD_BYTES
            and not (psc is None and psh is None and not pact
                     and lib.query("nasseg_conv_wgrad_lds3x3", B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil))):
        desc = (ptr(cur), K, ptr(dz), N, ptr(ws), ptr(psc) or 0, ptr(psh) or 0, pact) + tuple(geom)
        _group_wgrad("nasseg_conv_wgrad_many", cur, (cur, dz, psc, psh, ws), desc,
                     _fin_entry(ws, dwt, kh * kw, N, K, flat))
        return dwt
    st = _wgrad_stream(cur, (cur, dz, psc, psh, ws))  # (before this layer is queued for finalisation)
    lib.call(_k("nasseg_conv_wgrad", cur), ptr(cur), K, ptr(dz), N, _finish_wgrad(ws, dwt, kh * kw, N, K, flat),
             ptr(ws), ptr(psc), ptr(psh), pact, *geom, st)
    return dwt


def _pw_bwd_slabs(kind, cur, z, w, stride, pad, need_dw, need_dx, i, ops):
    """> 0: this op's whole backward (BatchNorm backward on load, weight gradient, input gradient)
    runs as ONE kernel, nasseg_conv_pw_bwd_bn (the value is its number of partial rows): a
    pointwise conv with a BatchNorm behind it, both gradients wanted, a map large enough to fill
    the GPU with slabs, and NO BatchNorm of the chain in front of it whose backward the separate
    backward-data kernel would fuse into its epilogue (an activation applied on load to the
    chain's input - pre_clf's ReLU - is handled: the kernel masks dx with its derivative)."""
    if not (FUSE_PW_BWD and kind == "dense" and need_dw and need_dx):
        return 0
    N, K, kh, kw = w.shape
    if not (kh == 1 and kw == 1 and stride == 1 and pad == 0):
        return 0
    if (cur.numel() + z.numel()) * cur.element_size() <= _PW_BWD_MIN_BYTES:
        return 0
    if K % 4 == 0 and i > 0 and ops[i - 1][4] and not (N >= K and K <= 64):
        # bn_prev in _ConvChain.backward: the producer's BatchNorm backward goes with the backward-data
        # kernel (statistics epilogue) - except where this conv widens (N >= K): there the one kernel plus
        # a bn_bwd_reduce pass over the K-channel gradient moves fewer bytes (4K + 2N per pixel) than
        # weight-gradient + backward-data with dz in between (3K + 4N): 16 -> 96 behind a BatchNorm,
        # which is what MobileNetV2's merged stem / stage-1 / stage-2 chain contains
        return 0
    B, _, H, W = cur.shape
    if K > 64 and not (_PW_BWD_WIDE and B * H * W >= _PW_BWD_WIDE_MIN_PIXELS):
        return 0
    return lib.query("nasseg_conv_pw_bwd_slabs", B, H, W, K, N)


FUSE_DW_BWD = os.environ.get("NASSEG_FUSE_DW_BWD", "1") != "0"
_DW_BWD_MIN_BYTES = int(os.environ.get("NASSEG_DW_BWD_MIN_BYTES", 24 << 20))
_FLAT_WGRAD_BN_MIN_BYTES = 24 << 20  # (output map of the stem above which its BatchNorm backward rides on the wgrad loads)


def _dw_bwd_rows(kind, cur, z, w, stride, pad, dil, need_dw, need_dx, i, ops):
    """> 0: this depthwise op's whole backward runs as ONE kernel, nasseg_dwconv_bwd_bn (the value
    is its number of partial rows): 3x3, between two BatchNorms of the chain (the previous op has
    one - InvertedResidual's expansion), both gradients wanted, a large map."""
    if not (FUSE_DW_BWD and kind == "dw" and need_dw and need_dx and i > 0 and ops[i - 1][4]):
        return 0
    if (cur.numel() + z.numel()) * cur.element_size() <= _DW_BWD_MIN_BYTES:
        return 0
    B, C, H, W = cur.shape
    return lib.query("nasseg_dwconv_bwd_bn_rows", B, C, H, W, w.shape[-1], stride, pad, dil)


def _flat_bn_ok(kind, need_dw, need_dx, psc, psh, pact, w, N, z):
    """the stem's weight gradient with the BatchNorm backward on its loads (nasseg_conv_wgrad_bn_flat)"""
    return (kind == "dense" and need_dw and not need_dx and psc is None and psh is None and not pact
            and w.shape[2] * w.shape[3] > 1 and N % 4 == 0
            and lib.query("nasseg_conv_fwd_pack_mode", w.shape[1], w.shape[2], w.shape[3]) == 2
            and z.numel() * z.element_size() > _FLAT_WGRAD_BN_MIN_BYTES)


def _wgrad_bn_ok(kind, cur, z, w, stride, pad, dil):
    """Can this layer's weight-gradient kernel apply the BatchNorm backward on load
    (nasseg_conv_wgrad_bn / nasseg_dwconv_wgrad_bn)?  Large maps only: the launches of small ones
    are grouped at the end of backward, after the backward-data kernels that need dz.  (The rule
    must not depend on whether finalisation is deferred: both ways give the same bits.)"""
    if (cur.numel() + z.numel()) * cur.element_size() <= _GROUP_WGRAD_BYTES:
        return False
    K, N = cur.shape[1], z.shape[1]
    if kind == "dw":
        return K % 4 == 0 and bool(lib.query("nasseg_dwconv_strip_ok", w.shape[-1], stride, dil))
    return (w.shape[2] == 1 and w.shape[3] == 1 and stride == 1 and pad == 0 and K % 4 == 0 and N % 4 == 0)


def _wgrad_bn(kind, cur, g, z, w, psc, psh, pact, bn, geom):
    """Weight gradient of a conv followed by BatchNorm, from the masked gradient ``g`` w.r.t. the
    BatchNorm output: returns (dw, dz); bn = (scale, shift, mean, invstd, sums, training, act) with
    act = the activation whose mask g still lacks (ACT_NONE: g arrived masked, or no activation)."""
    scale, shift, mean, invstd, sums, training, act = bn
    dz = torch.empty_like(z)
    dwt = torch.empty_like(w)
    if kind == "dw":
        B, H, W, C, Ho, Wo, k, stride, pad, dil = geom
        ws = _ws(cur, lib.query("nasseg_dwconv_wgrad_workspace", B, C, Ho, Wo, k))
        lib.call(_k("nasseg_dwconv_wgrad_bn", cur), ptr(cur), ptr(g), ptr(z), ptr(dz),
                 _finish_wgrad(ws, dwt, k * k, C, 1, 0), ptr(ws), ptr(psc), ptr(psh), pact, ptr(scale), ptr(shift),
                 ptr(mean), ptr(invstd), ptr(sums), int(training), act, *geom, current_stream())
    else:
        B, H, W, K, N = geom
        ws = _ws(cur, lib.query("nasseg_conv_wgrad_workspace", B, H, W, N, K, 1, 1))
        lib.call(_k("nasseg_conv_wgrad_bn", cur), ptr(cur), K, ptr(g), N, ptr(z), N, ptr(dz), N,
                 _finish_wgrad(ws, dwt, 1, N, K, 0), ptr(ws), ptr(psc), ptr(psh), pact, ptr(scale), ptr(shift),
                 ptr(mean), ptr(invstd), ptr(sums), int(training), act, B, H, W, K, N, current_stream())
    return dwt, dz


def _finish_wgrad(ws, dw, taps, N, K, flat):
    """Returns the pointer to pass as ``dw`` to a backward-weight entry point: the tensor itself,
    or NULL with the second stage queued when finalisation is deferred."""
    if not deferred_wgrad.active:
        return ptr(dw)
    deferred_wgrad.pending.append(_fin_entry(ws, dw, taps, N, K, flat))
    return None


def _fin_entry(ws, dw, taps, N, K, flat):
    return (ws, dw.data_ptr(), (ws.numel() // (taps * N * K), taps, N, K, flat))


def conv_out_size(size, k, stride, pad, dil):
    return (size + 2 * pad - dil * (k - 1) - 1) // stride + 1


def _colred(mode, a, lda, b, ldb, c, ldc, S, R, C, mul=1.0):
    """[S][nacc][C] sums; nacc = 2 for SUMSQ / DOT2."""
    nacc = 2 if mode in (RED_SUMSQ, RED_DOT2) else 1
    out = _vec(a, S * nacc * C)
    ws = _ws(a, lib.query("nasseg_colred_workspace", S, R, C))
    lib.call(_k("nasseg_colred", a), mode, ptr(a), lda, ptr(b), ldb, ptr(c), ldc, ptr(out), ptr(ws),
             S, R, C, float(mul), current_stream())
    return out


def _affine_act(x, scale, shift, res, act):
    B, C, H, W = x.shape
    y = _new(x, B, C, H, W)
    lib.call(_k("nasseg_affine_act", x), ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(y), x.numel(), C,
             act, current_stream())
    return y


def _axpby(a, b, alpha, beta, act=ACT_NONE):
    B, C, H, W = a.shape
    y = _new(a, B, C, H, W)
    lib.call(_k("nasseg_axpby", a), ptr(a), ptr(b), ptr(alpha), ptr(beta), ptr(y), a.numel(), C, act,
             current_stream())
    return y


def _act_bwd(dy, ref, act):
    dx = torch.empty_like(dy)
    lib.call(_k("nasseg_act_bwd", dy), ptr(dy), ptr(ref), ptr(dx), dy.numel(), act, current_stream())
    return dx


# ---------------------------------------------------------------------------
# depthwise convolution
# ---------------------------------------------------------------------------
class _DepthwiseConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, pad, dil, relu_in):
        x = _cl(x)
        w = weight.contiguous()
        B, C, H, W = x.shape
        K = w.shape[-1]
        if w.shape[0] != C or w.shape[1] != 1 or w.shape[2] != K:
            raise NassegError("depthwise weight {} does not match C={}".format(tuple(w.shape), C))
        Ho, Wo = conv_out_size(H, K, stride, pad, dil), conv_out_size(W, K, stride, pad, dil)
        if Ho <= 0 or Wo <= 0:
            raise NassegError("depthwise conv output would be empty")
        s = current_stream()
        wt = _vec(x, K * K * C)
        lib.call("nasseg_dw_pack_weight", ptr(w), ptr(wt), C, K, 0, s)
        y = _new(x, B, C, Ho, Wo)
        lib.call(_k("nasseg_dwconv", x), ptr(x), ptr(wt), ptr(y), None, None, ACT_RELU if relu_in else ACT_NONE,
                 None, None, ACT_NONE, B, H, W, C, Ho, Wo, K, stride, pad, dil, 0, None, s)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil, bool(relu_in))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, dil, relu_in = ctx.cfg
        dy = _cl(dy)
        B, C, H, W = x.shape
        K = w.shape[-1]
        Ho, Wo = dy.shape[2], dy.shape[3]
        s = current_stream()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = _vec(x, K * K * C)
            dx = _new(x, B, C, H, W)
            padb = dil * (K - 1) - pad
            if stride == 1 and padb >= 0:
                # correlation with the 180-degree rotated kernel
                lib.call("nasseg_dw_pack_weight", ptr(w), ptr(wt), C, K, 1, s)
                lib.call(_k("nasseg_dwconv", dy), ptr(dy), ptr(wt), ptr(dx), None, None, ACT_NONE, None, None,
                         ACT_NONE, B, Ho, Wo, C, H, W, K, 1, padb, dil, 0, None, s)
            else:
                lib.call("nasseg_dw_pack_weight", ptr(w), ptr(wt), C, K, 0, s)
                lib.call(_k("nasseg_dwconv", dy), ptr(dy), ptr(wt), ptr(dx), None, None, ACT_NONE, None, None,
                         ACT_NONE, B, Ho, Wo, C, H, W, K, stride, pad, dil, 1, None, s)
            if relu_in:
                dx = _act_bwd(dx, x, ACT_RELU)
        if ctx.needs_input_grad[1]:
            dw = _dw_wgrad(x, dy, w, None, None, ACT_RELU if relu_in else ACT_NONE,
                           (B, H, W, C, Ho, Wo, K, stride, pad, dil))
        return dx, dw, None, None, None, None


def depthwise_conv2d(x, weight, stride=1, padding=0, dilation=1, relu_in=False):
    """groups == channels convolution; ``relu_in`` fuses a preceding ReLU (DilConv)."""
    return _DepthwiseConv.apply(x, weight, int(stride), int(padding), int(dilation), bool(relu_in))


# ---------------------------------------------------------------------------
# dense convolution (1x1 and k x k) on the fp32 matrix cores
# ---------------------------------------------------------------------------
def _pack_dense(w, mode):
    """mode 'fwd' picks the layout nasseg_conv_fwd expects for this geometry; 1 = backward-data"""
    N, K, kh, kw = w.shape
    if mode == "fwd":
        mode = lib.query("nasseg_conv_fwd_pack_mode", K, kh, kw)
    if kh == 1 and kw == 1 and mode == 0:
        return w  # (N,K,1,1) contiguous already is [tap=0][N][K]
    wp = _vec(w, w.numel())
    lib.call("nasseg_conv_pack_weight", ptr(w), ptr(wp), N, K, kh, kw, mode, current_stream())
    return wp


_PACK_PLANS = {}
_PACK_SWEEP_AT = 1024  # plans; above it the entries of dead parameters are dropped


class _PackPlan(object):
    """Launch descriptor + destination buffer of one set of weights (see _pack_many).  `refs` are
    weak references to the source tensors: a plan whose parameters are gone is never served again
    (a new tensor may live at the same address) and is dropped at the next sweep."""
    __slots__ = ("n", "src", "dst", "dims", "views", "buf", "refs")

    def alive(self):
        return all(r() is not None for r in self.refs)


class PackMemo(list):
    """What ``packed_once`` remembers between the steps of ONE model / graphed stepper: the
    (key, plan) pairs of the chains seen last time - strong references, so the packed-weight
    buffers a captured hipGraph has baked in stay allocated for as long as their owner lives,
    whatever happens to the global cache - and the merged launch table built from them."""

    def __init__(self, *a):
        super(PackMemo, self).__init__(*a)
        self.merged = None


def _sweep_pack_plans():
    for k in [k for k, pl in _PACK_PLANS.items() if not pl.alive()]:
        del _PACK_PLANS[k]


class packed_once(object):
    """``with F.packed_once(memo): forward (+ backward) of one training step``

    Every conv chain re-packs its weights into the kernels' layouts with one small launch per
    chain and step (~50 launches on the headline network, each on the dependency path).  Within
    ONE step the parameters do not change, so inside this context the packs of all chains seen
    the last time ``memo`` (a ``PackMemo`` the caller keeps with the model) went through it are
    issued together at entry - nasseg_pack_weights takes any number of tensors - and the chains
    find their packed weights ready.  Chains met for the first time pack themselves as usual and
    are remembered in ``memo`` for the next entry.  Same kernel, same bytes: identical results.
    The context must not span a parameter update."""

    scope = None

    def __init__(self, memo):
        self.memo = memo

    def __enter__(self):
        self.prev, packed_once.scope = packed_once.scope, self
        self.done, self.seen, self.noted = set(), [], set()
        # only plans that still are the current ones of their keys (a plan rebuilt since has a
        # new buffer: whoever asks for that key will read the new one)
        entries = [(k, pl) for k, pl in self.memo if _PACK_PLANS.get(k) is pl]
        if entries:
            merged = getattr(self.memo, "merged", None)
            if merged is not None and (len(merged[4]) != len(entries)
                                       or any(a is not b[1] for a, b in zip(merged[4], entries))):
                merged = None
            if merged is None:
                plans = [pl for _, pl in entries]
                n = sum(pl.n for pl in plans)
                src = (ctypes.c_void_p * max(n, 1))()
                dst = (ctypes.c_void_p * max(n, 1))()
                dims = (ctypes.c_int * (7 * max(n, 1)))()
                j = 0
                for pl in plans:
                    for i in range(pl.n):
                        src[j], dst[j] = pl.src[i], pl.dst[i]
                        dims[7 * j:7 * j + 7] = pl.dims[7 * i:7 * i + 7]
                        j += 1
                merged = (n, src, dst, dims, plans)
                if isinstance(self.memo, PackMemo):
                    self.memo.merged = merged
            if merged[0]:
                lib.call("nasseg_pack_weights", merged[0], merged[1], merged[2], merged[3], current_stream())
            self.done.update(k for k, _ in entries)
        return self

    def __exit__(self, exc_type, exc, tb):
        packed_once.scope = self.prev
        if exc_type is None:
            self.memo[:] = self.seen
        return False


def _pack_many(like, items):
    """Re-pack several weights with ONE launch.  items: [(weight, kind)] or, for a slice of the
    input channels of a dense weight, [(weight, kind, koff, K)]; kind 'fwd' | 0 | 1 | 2 for dense
    weights (as _pack_dense; 5 = mode 1 with flipped taps, see _dense_backward_data), 'dw' |
    'dwflip' for depthwise ones.  Returns the packed tensors in order (the weight itself where
    its layout already is the packed one).

    The launch descriptor (pointer / shape tables) and the destination buffer of a given set of
    weights are built once and kept (_PackPlan): a chain packs the same parameters every step,
    and building the ctypes tables cost more host time than the launch.  The buffer is rewritten
    by every call; a forward's packed weights stay valid until the parameters change, i.e. for
    its own backward.  Plans are dropped only when their parameters are gone (never wholesale: a
    captured hipGraph holds their buffers' addresses - and, through its PackMemo, the plans)."""
    key = (like.device,) + tuple((it[0].data_ptr(), tuple(it[0].shape)) + tuple(it[1:]) for it in items)
    plan = _PACK_PLANS.get(key)
    if plan is not None and not plan.alive():
        plan = None  # (same addresses, other tensors: the old owner may still replay the old buffer)
    if plan is None:
        slots, descs, off = [], [], 0
        for item in items:
            w, kind = item[0], item[1]
            if kind in ("dw", "dwflip"):
                C, _, k, _ = w.shape
                d = (C, 1, k, k, 3 if kind == "dw" else 4, 0, 0)
                numel = w.numel()
            else:
                N, Ksrc, kh, kw = w.shape
                koff, K = (item[2], item[3]) if len(item) > 2 else (0, Ksrc)
                mode = lib.query("nasseg_conv_fwd_pack_mode", K, kh, kw) if kind == "fwd" else int(kind)
                if kh == 1 and kw == 1 and mode == 5:
                    mode = 1
                if kh == 1 and kw == 1 and mode == 0 and K == Ksrc:
                    slots.append(None)  # the weight itself
                    continue
                d = (N, K, kh, kw, mode, Ksrc if K != Ksrc else 0, koff)
                numel = N * K * kh * kw
            slots.append((off, numel))
            descs.append((w, d, off, numel))
            off += (numel + 3) // 4 * 4  # keep every packed tensor 16-byte aligned
        n = len(descs)
        plan = _PackPlan()
        plan.n = n
        plan.buf = _vec(like, off) if n else None
        plan.src = (ctypes.c_void_p * max(n, 1))(*[ptr(w) for w, _, _, _ in descs])
        plan.dst = (ctypes.c_void_p * max(n, 1))()
        plan.dims = (ctypes.c_int * (7 * max(n, 1)))()
        for j, (w, d, o, numel) in enumerate(descs):
            plan.dst[j] = ptr(plan.buf[o:o + numel])
            plan.dims[7 * j:7 * j + 7] = d
        plan.views = [None if sl is None else plan.buf[sl[0]:sl[0] + sl[1]] for sl in slots]
        plan.refs = [weakref.ref(it[0]) for it in items]
        if len(_PACK_PLANS) >= _PACK_SWEEP_AT:  # (parameters of discarded candidates)
            _sweep_pack_plans()
        _PACK_PLANS[key] = plan
        if packed_once.scope is not None:
            packed_once.scope.done.discard(key)  # (a new buffer: whatever was packed for this key is gone)
    scope = packed_once.scope
    if scope is not None and key not in scope.noted:
        scope.noted.add(key)
        scope.seen.append((key, plan))
    if plan.n and (scope is None or key not in scope.done):
        lib.call("nasseg_pack_weights", plan.n, plan.src, plan.dst, plan.dims, current_stream())
        if scope is not None:
            scope.done.add(key)  # (a weight used twice in the step is packed once)
    return [item[0] if v is None else v for item, v in zip(items, plan.views)]


def _dgrad_form(kh, kw, stride, pad, dil):
    """Backward-data of a stride-1 k x k conv is a forward conv over dy with flipped,
    role-swapped weights (pack kind 5) and pad' = dil*(k-1) - pad - the form the LDS-tiled
    3x3 kernel serves; everything else uses the transposed gather (pack kind 1)."""
    if stride == 1 and kh == kw and kh > 1 and dil * (kh - 1) - pad >= 0:
        return 5
    return 1


def _dense_dgrad_form(w, stride, pad, dil):
    """_dgrad_form for weight ``w`` (N,K,kh,kw); a forward conv over dy whose reduction
    axis (N) is so short that nasseg_conv_fwd would take its flat-packed path keeps the
    transposed form."""
    N, _, kh, kw = w.shape
    form = _dgrad_form(kh, kw, stride, pad, dil)
    if form == 5 and lib.query("nasseg_conv_fwd_pack_mode", N, kh, kw) != 0:
        return 1
    return form


def _dense_backward_data(dz, wb, form, x_shape, N, kh, kw, stride, pad, dil, res=None):
    """dx of a dense conv; wb packed with kind ``form`` (_dgrad_form); ``res`` (a map of dx's shape) is added in
    the epilogue."""
    B, K, H, W = x_shape
    Ho, Wo = dz.shape[2], dz.shape[3]
    dx = _new(dz, B, K, H, W)
    if form == 5:
        lib.call(_k("nasseg_conv_fwd", dz), ptr(dz), N, ptr(wb), ptr(dx), K, None, None, 0, None, None,
                 ACT_NONE, ptr(res), K if res is not None else 0, B, Ho, Wo, N, H, W, K, kh, kw, 1,
                 dil * (kh - 1) - pad, dil, 0, None, current_stream())
    else:
        lib.call(_k("nasseg_conv_fwd", dz), ptr(dz), N, ptr(wb), ptr(dx), K, None, None, 0, None, None,
                 ACT_NONE, ptr(res), K if res is not None else 0, B, Ho, Wo, N, H, W, K, kh, kw, stride, pad, dil, 1,
                 None, current_stream())
    return dx


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil):
        x = _cl(x)
        w = weight.contiguous()
        B, K, H, W = x.shape
        N, Kw, kh, kw = w.shape
        if Kw != K:
            raise NassegError("conv weight {} does not match C_in={}".format(tuple(w.shape), K))
        Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
        if Ho <= 0 or Wo <= 0:
            raise NassegError("conv output would be empty")
        y = _new(x, B, N, Ho, Wo)
        wp = _pack_dense(w, "fwd")
        lib.call(_k("nasseg_conv_fwd", x), ptr(x), K, ptr(wp), ptr(y), N, None, None, 0, None, ptr(bias),
                 ACT_NONE, None, 0, B, H, W, K, Ho, Wo, N, kh, kw, stride, pad, dil, 0, None,
                 current_stream())
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, dil, has_bias = ctx.cfg
        dy = _cl(dy)
        B, K, H, W = x.shape
        N, _, kh, kw = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        s = current_stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            form = _dense_dgrad_form(w, stride, pad, dil)
            (wp,) = _pack_many(dy, [(w, form)])
            dx = _dense_backward_data(dy, wp, form, (B, K, H, W), N, kh, kw, stride, pad, dil)
        if ctx.needs_input_grad[1]:
            dw = _dense_wgrad(x, dy, w, None, None, ACT_NONE, (B, H, W, K, Ho, Wo, N, kh, kw, stride, pad, dil))
        if has_bias and ctx.needs_input_grad[2]:
            db = _colred(RED_SUM, dy, N, None, 0, None, 0, 1, B * Ho * Wo, N)
        return dx, dw, db, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    """Dense convolution, groups == 1 (nn.Conv2d semantics, square stride/pad/dilation)."""
    return _Conv2d.apply(x, weight, bias, int(stride), int(padding), int(dilation))


# ---------------------------------------------------------------------------
# conv -> BN -> act chains with "normalise on read"
# ---------------------------------------------------------------------------
def _dw_backward_data(dz, wt, K, x_shape, stride, pad, dil, bn=None):
    """wt: the packed weight - flipped when the stride-1 correlation form applies, plain otherwise.
    bn = (z, scale, shift, mean, invstd, act) of the BatchNorm whose normalised output this conv
    read: the kernel then also masks the gradient with act' and emits the BatchNorm-backward
    partial sums.  Returns (dx, None | (rows buffer, number of rows))."""
    B, C, H, W = x_shape
    Ho, Wo = dz.shape[2], dz.shape[3]
    s = current_stream()
    dx = _new(dz, B, C, H, W)
    padb = dil * (K - 1) - pad
    if stride == 1 and padb >= 0:
        geom = (B, Ho, Wo, C, H, W, K, 1, padb, dil, 0)
    else:
        geom = (B, Ho, Wo, C, H, W, K, stride, pad, dil, 1)
    if bn is not None:
        nb = lib.query("nasseg_dwconv_bwd_data_bn_blocks", B, C, H, W, K, geom[7], geom[8], dil, geom[10])
        if nb > 0:
            z, scale, shift, mean, invstd, act = bn
            part = _ws(dz, (nb + 64) * 2 * C)
            lib.call(_k("nasseg_dwconv_bwd_data_bn", dz), ptr(dz), ptr(wt), ptr(dx), ptr(z), ptr(scale),
                     ptr(shift), ptr(mean), ptr(invstd), act, *geom, ptr(part), s)
            return dx, (part, nb)
    lib.call(_k("nasseg_dwconv", dz), ptr(dz), ptr(wt), ptr(dx), None, None, ACT_NONE, None, None,
             ACT_NONE, *geom, None, s)
    return dx, None


_IDENTITY = {}


def _identity_vectors(like, n):
    """(ones, zeros) of at least n floats on like's device - the "BatchNorm" that turns the fused
    backward-data epilogue into a plain act' mask (scale 1, shift 0, mean 0, invstd 1).  Created
    once per device and size class, never written."""
    key = (like.device, (n + 1023) // 1024)
    ent = _IDENTITY.get(key)
    if ent is None:
        m = key[1] * 1024
        ent = (torch.ones(m, device=like.device, dtype=torch.float32),
               torch.zeros(m, device=like.device, dtype=torch.float32))
        _IDENTITY[key] = ent
    return ent


# backward of pointwise conv + BatchNorm as one kernel where the chain allows it (csrc/conv_pwbwd.hip),
# for maps whose input + output exceed this many bytes (tools/kbench_pwbwd.py: 32 -> 32 at 4x128x256,
# 33 MB, one kernel 21 us / two kernels 26 us; 64 -> 64 at 4x32x64, 4 MB, 35 / 19 us - too few slabs)
FUSE_RES_GRAD = os.environ.get("NASSEG_FUSE_RES_GRAD", "1") != "0"  # (_ConvChain.backward: dx + dres in one epilogue)
FUSE_PW_BWD = os.environ.get("NASSEG_FUSE_PW_BWD", "1") != "0"
_PW_BWD_MIN_BYTES = int(os.environ.get("NASSEG_PW_BWD_MIN_BYTES", 24 << 20))
# the wide-input variant (K > 64: the four waves split N; round 3: chunks of the input and of the
# backward-data weight prefetched while the previous chunk is multiplied): 224 -> 64 at 4x256x512 (pre_clf,
# whose input gradient also carries the ReLU mask) 410-425 us against 710-730 us for the two kernels
# (tools/kbench_pwbwd.py), 192 -> 64 at 16x81x81 98 / 120 us, 192 -> 48 at 8x179x179 155 / 251 us; on maps
# too small for a slab per CU it loses (320 -> 64 at 16x11x11: 170 / 22 us), hence the pixel floor
_PW_BWD_WIDE = os.environ.get("NASSEG_PW_BWD_WIDE", "1") == "1"
_PW_BWD_WIDE_MIN_PIXELS = 1 << 16
# ConcatReduce's BatchNorm -> ReLU -> 1x1 conv on the concat slab as one node (the conv normalises on load)
FUSE_BN_RELU_CONV = os.environ.get("NASSEG_FUSE_BN_RELU_CONV", "1") != "0"
# Pool's 1x1 conv + BatchNorm -> 3x3 max pooling as one node (csrc/pool.hip: nasseg_maxpool_bn_fwd / _bwd)
FUSE_POOL_BN = os.environ.get("NASSEG_FUSE_POOL_BN", "1") != "0"
# ConcatReduce as one node fed by its producers' raw conv outputs (_CatReduce / Pending)
FUSE_CAT_REDUCE = os.environ.get("NASSEG_FUSE_CAT_REDUCE", "1") != "0"
# depthwise -> pointwise stages of a chain (SepConv, DilConv) as one kernel (csrc/sepconv.hip)
FUSE_SEPCONV = os.environ.get("NASSEG_FUSE_SEPCONV", "1") != "0"  # (the switch exists for A/B measurements)
# which pointwise forward / backward-data calls take the persistent kernel (include/nasseg.h:
# nasseg_conv_pw_min_pixels): unset = where it measured faster, 0 = wherever it can, a huge number = nowhere
_PW_MIN_PIXELS = os.environ.get("NASSEG_PW_MIN_PIXELS")
# ... and which take the N-split persistent kernel (nasseg_conv_pwn_mode): unset / 1 = where it measured faster,
# 0 = nowhere, 2 = wherever it can
_PWN_MODE = os.environ.get("NASSEG_PWN_MODE")


# elements of the stage's depthwise output above which a 5x5 stage runs as two kernels when the
# depthwise output has to be written anyway (training).  tools/kbench_sepconv.py on MI355X, fused
# vs separate in us: 32ch 128x256 21 / 29, 64ch 64x128 15 / 20, 64ch 128x256 46 / 48 - but
# 48ch 8x179x179 72 / 56, 24->64ch 256x512 93 / 79, 64ch dil-6 256x512 193 / 171: on large maps
# both forms are bandwidth-bound and the separate kernels keep more waves resident (the fused one
# holds the tile in LDS and its two phases do not overlap within a workgroup).  3x3 stages and
# inference (no depthwise output written) win at every size measured.
_SEPCONV_5X5_TRAIN_MAX = 10 << 20
# ... and, with 64 or more channels, already from 4 M elements (round 5, same table on MI355X: 64ch 128x256 45.1 / 42.6,
# 64ch 16x81x81 dilation 6 50.2 / 44.2 - but 64ch 64x128 14.5 / 19.6): the fused kernel's depthwise phase is bound by
# its vector-instruction issue (1721 VALU instructions per wave, tools/gpu.sh pmc), and the wider the tile's channel
# axis the fewer columns share a workgroup's weights
_SEPCONV_5X5_TRAIN_MAX_WIDE = 4 << 20


def _sepconv_ok(x, w_dw, w_pw, op_dw, op_pw, needs_grad):
    """Does nasseg_sepconv_fwd serve this depthwise conv (no BatchNorm behind it) followed by
    this pointwise conv - and is it the faster form here?"""
    B, C, H, W = x.shape
    k = w_dw.shape[-1]
    _, stride, pad, dil = op_dw[:4]
    N = w_pw.shape[0]
    if tuple(w_dw.shape) != (C, 1, k, k) or tuple(w_pw.shape) != (N, C, 1, 1):
        return False
    if op_pw[1:4] != (1, 0, 1):  # pointwise: stride 1, no padding
        return False
    Ho, Wo = conv_out_size(H, k, stride, pad, dil), conv_out_size(W, k, stride, pad, dil)
    if Ho <= 0 or Wo <= 0:
        return False
    limit = _SEPCONV_5X5_TRAIN_MAX_WIDE if (C >= 64 and needs_grad) else _SEPCONV_5X5_TRAIN_MAX
    if k == 5 and B * Ho * Wo * C > limit and (needs_grad or C % 16 != 0):
        return False
    return lib.query("nasseg_sepconv_blocks", B, C, Ho, Wo, N, k, stride, dil) > 0


# InvertedResidual's expansion never stored (csrc/irdw.hip; reference src/nn/layer_factory.py:125-158): a pointwise
# conv K -> N = 6 K + BatchNorm + activation in front of a 3x3 depthwise conv + BatchNorm, in a training step.  The
# expansion's statistics come from the moments of its INPUT (nasseg_irdw_stats), the depthwise forward and the
# one-kernel depthwise backward rebuild the expanded map on the matrix cores (nasseg_irdw_fwd / _bwd), the pointwise
# backward rebuilds it as it already did (nasseg_conv_pw_bwd_bn with z == NULL): the map - six times the block's input,
# 805 MB for 16 -> 96 at 4x512x1024 - is neither written nor read.  Where it pays (tools/kbench_irdw.py, MI355X, us for
# expansion + depthwise forward / depthwise backward): 16 -> 96 stride 2 at 4x512x1024 426 -> 230 / 404 -> 470,
# 24 -> 144 stride 1 at 4x256x512 257 -> 184 / 332 -> 318, 24 -> 144 stride 2 190 -> 135 / 164 -> 213 (level in time,
# taken for the bytes); not 32 -> 192 (no kernel of the pointwise backward rebuilds twelve channel tiles).  NASSEG_IRDW=0 switches it off (A/B); maps below NASSEG_IRDW_MIN_PIXELS keep the stored form (the
# extra launches of the statistics cost more than the bytes).
IRDW = os.environ.get("NASSEG_IRDW", "1") != "0"
_IRDW_MIN_PIXELS = int(os.environ.get("NASSEG_IRDW_MIN_PIXELS", 1 << 18))
# (stride 2 with K = 24: by the kernels alone the rebuilt backward loses what the forward wins - 190 -> 135 / 164 -> 213 us -
#  but replayed in lanes the step is level or ahead, 306.7 -> 308.5 images/s over three runs each, with 0.9 GB less HBM
#  traffic and 0.3 GiB less memory: profiles/r06_ab_irdw_s2_k24.txt)
_IRDW_S2_MAX_K = int(os.environ.get("NASSEG_IRDW_S2_MAX_K", 24))


def _irdw_ok(ops, i, weights, cur, pend, needs_in_grad, need_w, training):
    """op i is such an expansion, op i + 1 its depthwise conv, and every kernel involved serves the geometry"""
    if not IRDW or i + 1 >= len(ops):
        return False
    kind, stride, pad, dil, has_bn, act = ops[i][:6]
    kind2, stride2, pad2, dil2, has_bn2 = ops[i + 1][:5]
    w, w2 = weights[i], weights[i + 1]
    if not (kind == "dense" and kind2 == "dw" and has_bn and has_bn2 and training and ops[i + 1][6]):
        return False
    N, K, kh, kw = w.shape
    if not (kh == 1 and kw == 1 and stride == 1 and pad == 0 and w2.shape[-1] == 3 and pad2 == 1 and dil2 == 1
            and stride2 in (1, 2) and w2.shape[0] == N):
        return False
    if not (needs_in_grad and need_w):  # (the one-kernel backwards need both gradients)
        return False
    B, _, H, W = cur.shape
    if B * H * W < _IRDW_MIN_PIXELS or not (N > K and K % 4 == 0):
        return False
    if not ((stride2 == 1 and N <= 144) or (stride2 == 2 and K <= _IRDW_S2_MAX_K)):
        return False
    if pend is not None and pend[0] is None and pend[1] is None and not pend[2]:
        return False
    return (lib.query("nasseg_irdw_rows", B, H, W, K, N, stride2, 0) > 0
            and lib.query("nasseg_irdw_rows", B, H, W, K, N, stride2, 1) > 0
            and lib.query("nasseg_conv_pw_bwd_slabs", B, H, W, K, N) > 0
            and lib.query("nasseg_dwconv_bwd_bn_rows", B, N, H, W, 3, stride2, 1, 1) > 0)


class _ConvChain(torch.autograd.Function):
    """A run of convolutions (dense on the MFMA path or depthwise), each optionally followed
    by BatchNorm (+ReLU/ReLU6), as ONE autograd node in which a normalised activation that
    only feeds the next convolution is never written: the producer emits the raw conv output
    z plus the BatchNorm statistics (epilogue), the consumer applies act(scale*z + shift) as
    it loads its operand (prologue), backward recomputes the same on load.  Only the chain's
    final output is materialised (with the block's residual add fused in).

    cfg = (in_act0, ops); ops[i] = (kind, stride, pad, dil, has_bn, act, training, momentum, eps)
    with kind 'dense' | 'dw'; tensors = per op (weight, gamma, beta, running_mean,
    running_var, num_batches_tracked) with None where absent.
    """

    @staticmethod
    def forward(ctx, cfg, x, residual, *tensors):
        in_act0, ops = cfg[:2]
        res_is_x = residual is not None and residual is x  # (InvertedResidual: the block's input is its skip)
        x = _cl(x)
        s = current_stream()
        # the statistics vector a deferred tail hands out never has a gradient: without this autograd would
        # materialise a zero "gradient" for it in every backward (one fill launch per chain and step)
        ctx.set_materialize_grads(False)
        # (under no_grad ctx.needs_input_grad still reports the parameters' requires_grad flags: nothing
        #  will ever call backward then, and inference may fold every BatchNorm into its conv's epilogue.
        #  The grad mode is the CALLER's - cfg[2]: inside forward() autograd has switched it off)
        needs_grad = cfg[2] and any(ctx.needs_input_grad)
        res = _cl(residual) if residual is not None else None
        cur, pend = x, ((None, None, in_act0) if in_act0 else None)
        saved, meta = [], []
        n_ops = len(ops)
        in_pending = len(cfg) > 5 and cfg[5] is not None
        if in_pending:
            # the input is another chain's Pending: its BatchNorm + activation are op 0's prologue (tensors[-1]: the
            # producer's statistics vector); what backward returns for x is the gradient w.r.t. the ACTIVATED
            # input - exactly what a plain backward-data of op 0 computes
            K0 = x.shape[1]
            in_st = tensors[6 * n_ops]
            pend = (in_st[2 * K0:3 * K0], in_st[3 * K0:], cfg[5])
        # every layout of every weight of the chain (forward now, backward-data later) is
        # produced by one launch
        weights = [tensors[6 * i].contiguous() for i in range(n_ops)]
        items, bwd_slot = [], [None] * n_ops
        for i, op in enumerate(ops):
            items.append((weights[i], "dw" if op[0] == "dw" else "fwd"))
        if needs_grad:
            for i, (kind, stride, pad, dil) in enumerate(o[:4] for o in ops):
                if i == 0 and not ctx.needs_input_grad[1]:
                    continue
                if kind == "dw":
                    k = weights[i].shape[-1]
                    if stride == 1 and dil * (k - 1) - pad >= 0:
                        bwd_slot[i] = len(items)
                        items.append((weights[i], "dwflip"))
                    else:
                        bwd_slot[i] = i  # transposed gather reads the forward layout
                else:
                    # (a chain whose producer is a BatchNorm uses the fused transposed kernel)
                    fused = ((i > 0 and ops[i - 1][4]) or (i == 0 and in_act0)) and weights[i].shape[1] % 4 == 0
                    bwd_slot[i] = len(items)
                    items.append((weights[i], 1 if fused else _dense_dgrad_form(weights[i], stride, pad, dil)))
        packed = _pack_many(x, items)
        fused_dw = None  # (z_dw, ...) of a depthwise conv already computed together with the next op
        skip_op = False
        for i, (kind, stride, pad, dil, has_bn, act, training, momentum, eps) in enumerate(ops):
            if skip_op:  # (the depthwise half of an InvertedResidual expansion: done with op i - 1 below)
                skip_op = False
                continue
            w, gamma, beta, rm, rv, nbt = tensors[6 * i:6 * i + 6]
            w = weights[i]
            B, K, H, W = cur.shape
            last = i == n_ops - 1
            if (needs_grad and fused_dw is None and i + 1 < n_ops
                    and _irdw_ok(ops, i, weights, cur, pend, i > 0 or ctx.needs_input_grad[1],
                                 ctx.needs_input_grad[3 + 6 * i] and ctx.needs_input_grad[3 + 6 * (i + 1)], training)):
                # ---- expansion + depthwise with the expanded map never stored (csrc/irdw.hip) ----
                N = w.shape[0]
                psc, psh, pact = pend if pend is not None else (None, None, ACT_NONE)
                st1 = _vec(cur, 4 * N)
                mean1, invstd1, scale1, shift1 = st1[0:N], st1[N:2 * N], st1[2 * N:3 * N], st1[3 * N:]
                wsm = _ws(cur, lib.query("nasseg_irdw_stats_workspace", K))
                lib.call(_k("nasseg_irdw_stats", cur), ptr(cur), ptr(w), ptr(psc), ptr(psh), pact, B, H, W, K, N,
                         float(eps), float(momentum), ptr(gamma), ptr(beta), ptr(mean1), ptr(invstd1), ptr(scale1),
                         ptr(shift1), ptr(rm), ptr(rv), ptr(nbt), ptr(wsm), s)
                saved.extend([cur, psc, psh, None, st1, w, packed[bwd_slot[i]] if bwd_slot[i] is not None else None])
                meta.append((pact,))
                _, stride2, pad2, dil2, _, act2, _, momentum2, eps2 = ops[i + 1]
                w2, gamma2, beta2, rm2, rv2, nbt2 = tensors[6 * (i + 1):6 * (i + 1) + 6]
                w2 = weights[i + 1]
                Ho, Wo = conv_out_size(H, 3, stride2, 1, 1), conv_out_size(W, 3, stride2, 1, 1)
                z2 = _new(cur, B, N, Ho, Wo)
                st2 = _vec(cur, 4 * N)
                rows2 = lib.query("nasseg_irdw_rows", B, H, W, K, N, stride2, 0)
                part2 = _ws(cur, (rows2 + 64) * 2 * N)
                lib.call(_k("nasseg_irdw_fwd", cur), ptr(cur), ptr(w), ptr(packed[i + 1]), ptr(z2), ptr(psc), ptr(psh),
                         pact, ptr(scale1), ptr(shift1), act, B, H, W, K, N, Ho, Wo, stride2, ptr(part2), s)
                lib.call("nasseg_bn_finalize", ptr(part2), rows2, B * Ho * Wo, N, float(eps2), float(momentum2),
                         ptr(gamma2), ptr(beta2), ptr(st2[0:N]), ptr(st2[N:2 * N]), ptr(st2[2 * N:3 * N]),
                         ptr(st2[3 * N:]), ptr(rm2), ptr(rv2), ptr(nbt2), s)
                # (input None: the depthwise conv's input does not exist - backward rebuilds it from op i's)
                saved.extend([None, scale1, shift1, z2, st2, w2,
                              packed[bwd_slot[i + 1]] if bwd_slot[i + 1] is not None else None])
                meta.append((act,))
                cur, pend = z2, (st2[2 * N:3 * N], st2[3 * N:], act2)
                skip_op = True
                continue
            if (kind == "dw" and not has_bn and not last and FUSE_SEPCONV and ops[i + 1][0] == "dense"
                    and not (i + 2 == n_ops and res is not None and not needs_grad)
                    and _sepconv_ok(cur, w, weights[i + 1], ops[i], ops[i + 1], needs_grad)):
                # SepConv / DilConv stage: this depthwise conv and the pointwise conv that follows
                # run as ONE kernel (csrc/sepconv.hip) when op i+1 comes up; nothing to do here but
                # remember the stage's input and its prologue
                fused_dw = (cur, pend, w, packed[i], i)
                continue
            if fused_dw is not None:
                # the pointwise half of a fused stage: its input is the (virtual) depthwise output
                dk = fused_dw[2].shape[-1]
                _, ds_, dp_, dd_ = ops[fused_dw[4]][:4]
                H, W = conv_out_size(H, dk, ds_, dp_, dd_), conv_out_size(W, dk, ds_, dp_, dd_)
                pend = None  # (the prologue belongs to the depthwise half: kept in fused_dw)
            if kind == "dw":
                k = w.shape[-1]
                if w.shape[0] != K or w.shape[1] != 1:
                    raise NassegError("depthwise weight {} does not match C={}".format(tuple(w.shape), K))
                N, kh, kw = K, k, k
                strip = bool(lib.query("nasseg_dwconv_strip_ok", k, stride, dil))
                pro_ok = strip or (pend is not None and pend[0] is None and pend[2] == ACT_RELU)
                stats_ok = strip
            else:
                N, Kw, kh, kw = w.shape
                if Kw != K:
                    raise NassegError("conv weight {} does not match C_in={}".format(tuple(w.shape), K))
                pointwise = kh == 1 and kw == 1 and stride == 1 and pad == 0
                pro_ok = pointwise and K % 4 == 0 and N % 4 == 0
                stats_ok = N % 4 == 0
            Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
            if Ho <= 0 or Wo <= 0:
                raise NassegError("conv output would be empty")
            if pend is not None and not pro_ok:
                cur = (_affine_act(cur, pend[0], pend[1], None, pend[2]) if pend[0] is not None
                       else _axpby(cur, None, None, None, pend[2]))
                pend = None
            psc, psh, pact = pend if pend is not None else (None, None, ACT_NONE)
            M = B * Ho * Wo
            z = _new(cur, B, N, Ho, Wo)
            fold = has_bn and not training and not needs_grad and not (kind == "dw" and last and res is not None)
            stats = part = None
            nblk = 0
            if has_bn:
                if training and M <= 1:
                    raise ValueError("Expected more than 1 value per channel when training, got input "
                                     "size {}".format((B, N, Ho, Wo)))
                stats = _vec(cur, 4 * N)  # mean | invstd | scale | shift
                mean, invstd, scale, shift = (stats[0:N], stats[N:2 * N], stats[2 * N:3 * N],
                                              stats[3 * N:])
                if not training:
                    lib.call("nasseg_bn_eval_params", N, float(eps), ptr(gamma), ptr(beta), ptr(rm),
                             ptr(rv), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), s)
                elif stats_ok:
                    if fused_dw is not None:
                        nblk = lib.query("nasseg_sepconv_blocks", B, K, Ho, Wo, N, dk, ds_, dd_)
                    elif kind == "dw":
                        nblk = lib.query("nasseg_dwconv_stats_blocks", B, N, Ho, Wo, kh, stride, dil)
                    else:
                        nblk = lib.query("nasseg_conv_fwd_stats_rows", B, Ho, Wo, N, K, kh, kw, stride, pad, dil)
                    part = _ws(cur, (nblk + 64) * 2 * N)
            o_sc = o_sh = o_res = None
            o_act = ACT_NONE
            if fold:  # inference: BN (+act, +residual) folded into the conv's epilogue
                o_sc, o_sh, o_act = scale, shift, act
                if last and res is not None:
                    o_res = res
            wp = packed[i]
            if fused_dw is not None:
                # (cur is still the depthwise conv's input: geometry of the stage from op i-1)
                xin, dpend, dww, dwp, di = fused_dw
                fused_dw = None
                zdw = _new(xin, B, K, Ho, Wo) if needs_grad else None
                dsc, dsh, dact = dpend if dpend is not None else (None, None, ACT_NONE)
                lib.call(_k("nasseg_sepconv_fwd", xin), ptr(xin), ptr(dwp), ptr(w), ptr(zdw), ptr(z), ptr(dsc),
                         ptr(dsh), dact, ptr(o_sc), ptr(o_sh), o_act, B, xin.shape[2], xin.shape[3], K, Ho, Wo, N,
                         dk, ds_, dp_, dd_, ptr(part), s)
                if needs_grad:
                    saved.extend([xin, dsc, dsh, zdw, None, dww,
                                  packed[bwd_slot[di]] if bwd_slot[di] is not None else None])
                    meta.append((dact,))
                    cur = zdw
            elif kind == "dw":
                lib.call(_k("nasseg_dwconv", cur), ptr(cur), ptr(wp), ptr(z), ptr(psc), ptr(psh), pact, ptr(o_sc),
                         ptr(o_sh), o_act, B, H, W, K, Ho, Wo, kh, stride, pad, dil, 0, ptr(part), s)
            else:
                lib.call(_k("nasseg_conv_fwd", cur), ptr(cur), K, ptr(wp), ptr(z), N, ptr(psc), ptr(psh), pact,
                         ptr(o_sc), ptr(o_sh), o_act, ptr(o_res), N, B, H, W, K, Ho, Wo, N, kh, kw,
                         stride, pad, dil, 0, ptr(part), s)
            if needs_grad:
                saved.extend([cur, psc, psh, z, stats, w,
                              packed[bwd_slot[i]] if bwd_slot[i] is not None else None])
                meta.append((pact,))
            if fold:
                cur, pend = z, None
                if last and res is not None:
                    res = None  # consumed by the epilogue
                continue
            if has_bn:
                if training:
                    if part is not None:
                        lib.call("nasseg_bn_finalize", ptr(part), nblk, M, N, float(eps), float(momentum),
                                 ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                 ptr(rm), ptr(rv), ptr(nbt), s)
                    else:
                        ws = _ws(cur, lib.query("nasseg_colred_workspace", 1, M, N))
                        lib.call(_k("nasseg_bn_stats", z), ptr(z), N, M, N, float(eps), float(momentum),
                                 ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                 ptr(rm), ptr(rv), ptr(nbt), ptr(ws), s)
                cur, pend = z, (scale, shift, act)
            else:
                cur, pend = z, None
        pool = cfg[3] if len(cfg) > 3 else None
        defer = len(cfg) > 4 and cfg[4]
        tail = None
        pool_idx = None
        pool_fused = False
        if pool is not None:
            # Pool (src/nn/layer_factory.py:161-178): 3x3 max pooling behind the chain's last BatchNorm.  With the
            # BatchNorm still pending (training, or inference under autograd) the pooling applies it as it loads
            # the raw conv output - the normalised map is never written; folded into the conv's epilogue
            # (inference without grad) the pooling reads the finished map.  One node either way.
            pk, ps, pp = pool
            B, N, Hc, Wc = cur.shape
            Hp, Wp = conv_out_size(Hc, pk, ps, pp, 1), conv_out_size(Wc, pk, ps, pp, 1)
            if Hp <= 0 or Wp <= 0:
                raise NassegError("max pooling output would be empty")
            psc, psh = (pend[0], pend[1]) if pend is not None else (None, None)
            if (res is not None or (pend is not None and pend[2] != ACT_NONE) or pk != 3 or pp != 1
                    or ps not in (1, 2)):
                raise NassegError("conv_chain: the pooled tail serves conv + BatchNorm -> 3x3 max pooling only")
            y = _new(cur, B, N, Hp, Wp)
            if needs_grad:
                pool_idx = torch.empty((B, Hp, Wp, N), device=cur.device, dtype=torch.uint8)
            lib.call(_k("nasseg_maxpool_bn_fwd", cur), ptr(cur), ptr(psc), ptr(psh), ptr(y), ptr(pool_idx), B, Hc,
                     Wc, N, Hp, Wp, ps, pp, s)
            pool_fused = pend is not None
        elif pend is not None:
            if defer and res is None and pend[0] is not None:
                # deferred tail: the consumer applies act(scale*z + shift) as it loads (Pending below); what
                # comes back in backward is still the gradient w.r.t. the BatchNorm's activated output
                y, tail = cur, stats
            else:
                y = _affine_act(cur, pend[0], pend[1], res, pend[2])
        elif res is not None:
            y = _axpby(cur, res, None, None)
        else:
            y = cur
        if needs_grad:
            if pool is not None:
                saved.append(pool_idx)
            ctx.save_for_backward(*[t for t in saved])
            ctx.meta = (cfg, meta, residual is not None, tuple(x.shape), pool_fused, res_is_x)
            ctx.n_inputs = 3 + len(tensors)
        if defer:
            if tail is None:
                return y, None
            ctx.mark_non_differentiable(tail)
            return y, tail  # (mean | invstd | scale | shift of the last BatchNorm)
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:  # (no gradient reaches the chain's output: nothing to hand on)
            return (None,) * ctx.n_inputs
        cfg, meta, has_res, x_shape = ctx.meta[:4]
        pool_fused = ctx.meta[4] if len(ctx.meta) > 4 else False
        in_act0, ops = cfg[:2]
        pool = cfg[3] if len(cfg) > 3 else None
        sv = ctx.saved_tensors
        fused_in0 = bool(in_act0)  # (forward packed op 0's backward-data weights for the fused kernel)
        g = _cl(dy)
        s = current_stream()
        n_ops = len(ops)
        grads = [None] * (6 * n_ops)
        dres = g if (has_res and ctx.needs_input_grad[2]) else None
        # x is also the residual: both gradients go to the same tensor, and autograd would add them with a launch of
        # its own - where op 0's input gradient comes from the plain backward-data call, its epilogue adds dres
        fuse_res = (FUSE_RES_GRAD and len(ctx.meta) > 5 and ctx.meta[5] and dres is not None
                    and ctx.needs_input_grad[1] and not in_act0)
        pre = None  # BatchNorm-backward partial rows of op i that came with g (fused dgrad epilogue)
        g_masked = False  # g already carries act' of op i's activation (with or without such rows)
        if _TAIL_ROWS and pool is None:
            # a consumer of this chain's deferred tail (_CatReduce) has already masked the gradient and summed
            # it against the last BatchNorm's xhat: its rows come by the side (keyed by the gradient tensor itself)
            ent = _TAIL_ROWS.pop(dy.data_ptr(), None)
            # ... AND still holding what that consumer wrote: with a second consumer of the same Pending autograd may
            # have accumulated another gradient INTO this tensor (same object, same address) - the version
            # counter tells, and the chain then runs its own reduction over the summed gradient
            if ent is not None and ent[0]() is dy and dy._version == ent[3] and ops[-1][4]:
                pre = (ent[1], ent[2])
        if pool is not None:
            # the pooled tail: gradient w.r.t. the last BatchNorm's output by a gather over the windows -
            # together with that BatchNorm's backward sums when the pooling had applied it on load
            pk, ps, pp = pool
            pool_idx = sv[7 * n_ops]
            z_last, st_last = sv[7 * (n_ops - 1) + 3], sv[7 * (n_ops - 1) + 4]
            Bp, Np, Hz, Wz = z_last.shape
            g_full = _new(g, Bp, Np, Hz, Wz)
            nb = lib.query("nasseg_maxpool_bn_bwd_blocks", Bp, Hz, Wz, Np, pk, ps, pp) if pool_fused else 0
            if nb > 0:
                part = _ws(g, (nb + 64) * 2 * Np)
                lib.call(_k("nasseg_maxpool_bn_bwd", g), ptr(g), ptr(pool_idx), ptr(z_last), ptr(st_last[0:Np]),
                         ptr(st_last[Np:2 * Np]), ptr(g_full), ptr(part), Bp, Hz, Wz, Np, g.shape[2], g.shape[3],
                         ps, pp, s)
                pre = (part, nb)
            else:
                lib.call(_k("nasseg_pool_bwd", g), 0, ptr(g), ptr(pool_idx), ptr(g_full), Bp, Hz, Wz, Np,
                         g.shape[2], g.shape[3], pk, ps, pp, s)
            g = g_full
        masked_in0 = False  # dx already multiplied by in_act0' (one-kernel pointwise backward of op 0)
        for i in range(n_ops - 1, -1, -1):
            kind, stride, pad, dil, has_bn, act, training, momentum, eps = ops[i]
            cur, psc, psh, z, stats, w, wb = sv[7 * i:7 * i + 7]
            (pact,) = meta[i]
            # InvertedResidual's expansion that was never stored (forward: _irdw_ok): the depthwise op has no input
            # tensor, the expansion no output tensor - their one-kernel backwards rebuild it
            ir_dw = cur is None
            ir_pw = z is None
            if ir_pw:
                B, N, Ho, Wo = cur.shape[0], w.shape[0], cur.shape[2], cur.shape[3]
            else:
                B, N, Ho, Wo = z.shape
            M = B * Ho * Wo
            need_dw = ctx.needs_input_grad[3 + 6 * i]
            need_dx = i > 0 or ctx.needs_input_grad[1]
            fused_bn = None  # BatchNorm backward applied by the weight-gradient kernel on load
            if has_bn:
                mean, invstd, scale, shift = stats[0:N], stats[N:2 * N], stats[2 * N:3 * N], stats[3 * N:]
                sums = _vec(g, 2 * N)
                act_left = ACT_NONE if (pre is not None or g_masked) else act  # (the mask still to be applied to g)
                pw_bact = act_left
                go_on = need_dw or need_dx
                if ir_pw:
                    pw_nsl, dw_rows = lib.query("nasseg_conv_pw_bwd_slabs", B, Ho, Wo, cur.shape[1], N), 0
                elif ir_dw:
                    pw_nsl, dw_rows = 0, 1
                else:
                    pw_nsl = _pw_bwd_slabs(kind, cur, z, w, stride, pad, need_dw, need_dx, i, ops) if go_on else 0
                    dw_rows = (_dw_bwd_rows(kind, cur, z, w, stride, pad, dil, need_dw, need_dx, i, ops)
                               if go_on else 0)
                # the stem (small-K k x k conv, no gradient for the image): BatchNorm backward on load in the
                # weight-gradient kernel, dz never written (nasseg_conv_wgrad_bn_flat)
                flat_bn = go_on and not (ir_pw or ir_dw) and _flat_bn_ok(kind, need_dw, need_dx, psc, psh, pact, w, N, z)
                on_load = go_on and (pw_nsl > 0 or dw_rows > 0 or flat_bn)  # (a kernel below applies the BatchNorm backward)
                wgrad_bn = go_on and not on_load and need_dw and _wgrad_bn_ok(kind, cur, z, w, stride, pad, dil)
                plain_apply = go_on and not on_load and not wgrad_bn  # (dz by a bn_bwd_apply pass)
                lazy_rows = None  # the sums as rows the apply kernel adds up itself
                if pre is not None:
                    # g arrived masked, with its per-workgroup {sum g, sum g*xhat} rows
                    if plain_apply and _rows_small(pre[1], N):
                        lazy_rows = (pre[0], pre[1])
                    else:
                        lib.call("nasseg_rows_sum", ptr(pre[0]), pre[1], 2 * N, ptr(sums), s)
                else:
                    lazy_rows = _bn_bwd_reduce(g, z, scale, shift, mean, invstd, act_left, sums, M, N, plain_apply)
                if ctx.needs_input_grad[3 + 6 * i + 1]:
                    grads[6 * i + 1] = sums[N:2 * N]
                if ctx.needs_input_grad[3 + 6 * i + 2]:
                    grads[6 * i + 2] = sums[0:N]
                if not go_on:
                    g = None
                    break
                if on_load:
                    dz = None  # (the one-kernel backward / flat weight gradient below applies the BatchNorm backward on load)
                elif wgrad_bn:
                    # the weight-gradient kernel below computes dz while it loads g and z (masking
                    # g first if it did not arrive masked) and leaves it behind for the
                    # backward-data kernel
                    fused_bn = (scale, shift, mean, invstd, sums, training, act_left)
                    dz = None
                else:
                    dz = _bn_bwd_apply(g, z, scale, shift, mean, invstd, sums, M, N, training, act_left,
                                       torch.empty_like(z), lazy_rows)
            else:
                dz = g
                pw_nsl = dw_rows = 0
                flat_bn = False
                if not (need_dw or need_dx):
                    g = None
                    break
            if ir_dw:
                # ---- the depthwise conv behind a rebuilt expansion: nasseg_dwconv_bwd_bn with z1 = W1 x from op i - 1 ----
                x_in, xpsc, xpsh, _, st1, w1, _ = sv[7 * (i - 1):7 * (i - 1) + 7]
                (xpact,) = meta[i - 1]
                Bc, K1, H, W = x_in.shape
                K = N
                rows = lib.query("nasseg_irdw_rows", Bc, H, W, K1, K, stride, 1)
                dwt = torch.empty_like(w)
                ws = _ws(g, rows * 9 * K)
                part = _ws(g, (rows + 64) * 2 * K)
                g_in = _new(g, Bc, K, H, W)
                lib.call(_k("nasseg_irdw_bwd", x_in), ptr(x_in), ptr(w1), ptr(g), ptr(z), ptr(wb), int(stride == 1),
                         ptr(g_in), _finish_wgrad(ws, dwt, 9, K, 1, 0), ptr(ws), ptr(xpsc), ptr(xpsh), xpact,
                         ptr(st1[2 * K:3 * K]), ptr(st1[3 * K:]), ptr(st1[0:K]), ptr(st1[K:2 * K]), ops[i - 1][5],
                         ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums), int(training), pw_bact,
                         Bc, H, W, K1, K, Ho, Wo, stride, ptr(part), s)
                grads[6 * i] = dwt
                g, pre = g_in, (part, rows)
                g_masked = False
                continue
            Bc, K, H, W = cur.shape
            pre = None
            g_masked = False
            bn_prev = None
            if need_dx and i > 0 and ops[i - 1][4] and K % 4 == 0:
                # the producer of this conv's input is a BatchNorm of the chain: fuse the first
                # half of ITS backward into the backward-data kernel below
                zp, stp = sv[7 * (i - 1) + 3], sv[7 * (i - 1) + 4]
                bn_prev = (zp, stp[2 * K:3 * K], stp[3 * K:], stp[0:K], stp[K:2 * K], ops[i - 1][5])
            elif need_dx and i == 0 and in_act0 and K % 4 == 0 and (kind == "dw" or fused_in0):
                # the chain's input went through an activation on load (ReLU ahead of DilConv's
                # depthwise conv, of pre_clf's 1x1): the same epilogue with an identity BatchNorm
                # multiplies dx by act'(x) - no separate pass over dx and x
                if kind == "dw":
                    one, zero = _identity_vectors(cur, K)
                    bn_prev = (cur, one, zero, zero, one, in_act0)
                else:
                    bn_prev = (cur, None, None, None, None, in_act0)  # mask-only epilogue
            if kind == "dw" and dw_rows > 0:
                # 3x3 depthwise conv between two BatchNorms of the chain: its whole backward in one
                # pass (csrc/dwconv.hip: dw3x3_bwd_bn_kernel) - BatchNorm backward on load, weight
                # gradient, masked input gradient + the partial sums of the BatchNorm in front
                zp, psc_, psh_, pmu_, pis_, pact_ = bn_prev
                dwt = torch.empty_like(w)
                ws = _ws(cur, dw_rows * 9 * K)
                part = _ws(cur, (dw_rows + 64) * 2 * K)
                g_in = _new(cur, Bc, K, H, W)
                # (wb: what forward packed for this op's backward-data - rotated for stride 1, plain else)
                lib.call(_k("nasseg_dwconv_bwd_bn", cur), ptr(cur), ptr(g), ptr(z), ptr(wb), int(stride == 1),
                         ptr(g_in), _finish_wgrad(ws, dwt, 9, K, 1, 0), ptr(ws), ptr(psc_), ptr(psh_), ptr(pmu_),
                         ptr(pis_), pact_, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums), int(training),
                         pw_bact, Bc, H, W, K, Ho, Wo, 3, stride, pad, dil, ptr(part), s)
                grads[6 * i] = dwt
                g, pre = g_in, (part, dw_rows)
                continue
            if kind == "dw":
                k = w.shape[-1]
                if fused_bn is not None:
                    grads[6 * i], dz = _wgrad_bn("dw", cur, g, z, w, psc, psh, pact, fused_bn,
                                                 (Bc, H, W, K, Ho, Wo, k, stride, pad, dil))
                elif need_dw:
                    grads[6 * i] = _dw_wgrad(cur, dz, w, psc, psh, pact,
                                             (Bc, H, W, K, Ho, Wo, k, stride, pad, dil))
                g = None
                if need_dx:
                    g, pre = _dw_backward_data(dz, wb, k, (Bc, K, H, W), stride, pad, dil, bn_prev)
            else:
                _, _, kh, kw = w.shape
                if flat_bn:
                    dwt = torch.empty_like(w)
                    ws = _ws(cur, lib.query("nasseg_conv_wgrad_workspace", Bc, Ho, Wo, N, K, kh, kw))
                    lib.call(_k("nasseg_conv_wgrad_bn_flat", cur), ptr(cur), K, ptr(g), N, ptr(z), N,
                             _finish_wgrad(ws, dwt, kh * kw, N, K, 1), ptr(ws), ptr(scale), ptr(shift), ptr(mean),
                             ptr(invstd), ptr(sums), int(training), act_left, Bc, H, W, K, Ho, Wo, N, kh, kw,
                             stride, pad, dil, s)
                    grads[6 * i] = dwt
                    g = None
                    continue
                if pw_nsl > 0:
                    # pointwise conv + BatchNorm, nothing to fuse towards the producer: BatchNorm
                    # backward on load, weight gradient and input gradient in ONE kernel - dz is
                    # neither written nor read back (csrc/conv_pwbwd.hip)
                    nsl, bact_ = pw_nsl, pw_bact
                    dwt = torch.empty_like(w)
                    ws = _ws(cur, nsl * N * K)
                    g_in = _new(cur, Bc, K, H, W)
                    # op 0 of a chain that applies an activation to its input on load, or a widening conv
                    # behind a BatchNorm of the chain (_pw_bwd_slabs): dx is masked with act' here - the
                    # kernel takes the mask from the ACTIVATED input tile it holds - and what goes on to op
                    # i - 1 is the gradient w.r.t. its BatchNorm's output
                    behind_bn = i > 0 and ops[i - 1][4]
                    dx_act = pact if ((i == 0 and in_act0 and psc is None and psh is None) or behind_bn) else ACT_NONE
                    # ... and, K <= 64, comes with the per-slab sums of that BatchNorm's backward (dx_stats)
                    part = pmu_ = pis_ = None
                    skip_g = dres if (fuse_res and i == 0 and K % 4 == 0) else None  # (x is also the block's skip)
                    if behind_bn and K <= 64:
                        stp = sv[7 * (i - 1) + 4]
                        pmu_, pis_ = stp[0:K], stp[K:2 * K]
                        part = _ws(cur, (nsl + 64) * 2 * K)
                    # (z only where the kernel loads it: where it rebuilds z = W x the argument is NULL - an explicit
                    #  contract instead of a pointer the kernel ignores, and NULL is also what says "never stored")
                    z_arg = z if (z is not None and lib.query("nasseg_conv_pw_bwd_reads_z", Bc, H, W, K, N)) else None
                    lib.call(_k("nasseg_conv_pw_bwd_bn", cur), ptr(cur), ptr(g), ptr(z_arg), ptr(wb), ptr(g_in),
                             _finish_wgrad(ws, dwt, 1, N, K, 0), ptr(ws), ptr(psc), ptr(psh), pact, dx_act,
                             ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums), int(training), bact_,
                             Bc, H, W, K, N, ptr(pmu_), ptr(pis_), ptr(part), ptr(skip_g), s)
                    if skip_g is not None:
                        dres = None  # (it is inside dx)
                    grads[6 * i] = dwt
                    g = g_in
                    masked_in0 = bool(dx_act) and i == 0
                    g_masked = behind_bn
                    if part is not None:
                        pre = (part, nsl)
                    continue
                if fused_bn is not None:
                    grads[6 * i], dz = _wgrad_bn("dense", cur, g, z, w, psc, psh, pact, fused_bn,
                                                 (Bc, H, W, K, N))
                elif need_dw:
                    grads[6 * i] = _dense_wgrad(cur, dz, w, psc, psh, pact,
                                                (Bc, H, W, K, Ho, Wo, N, kh, kw, stride, pad, dil))
                g = None
                if need_dx:
                    if bn_prev is not None:
                        g = _new(cur, Bc, K, H, W)
                        zp, psc_, psh_, pmu_, pis_, pact_ = bn_prev
                        pw1 = kh == 1 and kw == 1 and stride == 1 and pad == 0
                        nb = (lib.query("nasseg_conv_fwd_stats_blocks", Bc, H, W, K, N, 2 * int(pw1))
                              if pmu_ is not None else 0)
                        part = _ws(cur, (nb + 64) * 2 * K) if nb else None
                        lib.call(_k("nasseg_conv_bwd_data_bn", dz), ptr(dz), N, ptr(wb), ptr(g), K, ptr(zp), K,
                                 ptr(psc_), ptr(psh_), ptr(pmu_), ptr(pis_), pact_, Bc, Ho, Wo, N, H, W,
                                 K, kh, kw, stride, pad, dil, ptr(part), s)
                        pre = (part, nb)
                    else:
                        g = _dense_backward_data(dz, wb, _dense_dgrad_form(w, stride, pad, dil),
                                                 (Bc, K, H, W), N, kh, kw, stride, pad, dil,
                                                 dres if (fuse_res and i == 0) else None)
                        if fuse_res and i == 0:
                            dres = None  # (it is inside dx)
        dx = None
        if g is not None and ctx.needs_input_grad[1]:
            dx = g
            if in_act0 and pre is None and not masked_in0:
                # the chain started with an activation applied on load and the backward-data
                # kernel had no fused mask for this geometry
                dx = _act_bwd(dx, sv[0], in_act0)
        return (None, dx, dres) + tuple(grads) + ((None,) if (len(cfg) > 5 and cfg[5] is not None) else ())


# BatchNorm-backward partial rows handed from a consumer's backward to the producer chain's, by the side of
# the gradient tensor: data_ptr -> (weakref to that tensor, rows, number of rows).  An entry is only honoured
# for the very tensor object it was made for (a dead or different object: the chain reduces as usual).
_TAIL_ROWS = {}
FUSE_TAIL_ROWS = os.environ.get("NASSEG_FUSE_TAIL_ROWS", "1") != "0"


class Pending(object):
    """A conv chain's raw last conv output whose BatchNorm (+ activation) is still to be applied:
    y = act(scale[c]*z + shift[c]).  A consumer that understands it (cat_reduce) applies the tail as it loads z -
    the normalised map is never written; anything else calls materialize().  Backward contract: the gradient
    that flows back into ``z``'s slot is the one w.r.t. y (the chain's backward is the same either way: it has
    always received dL/dy and redone mask and BatchNorm backward from z)."""

    __slots__ = ("z", "stats", "act", "_mat")

    def __init__(self, z, stats, act):
        self.z, self.stats, self.act = z, stats, int(act)  # stats: mean | invstd | scale | shift, C each
        self._mat = None

    scale = property(lambda self: self.stats[2 * self.z.shape[1]:3 * self.z.shape[1]])
    shift = property(lambda self: self.stats[3 * self.z.shape[1]:])
    shape = property(lambda self: self.z.shape)
    dtype = property(lambda self: self.z.dtype)
    device = property(lambda self: self.z.device)

    def size(self, *a):
        return self.z.size(*a)

    def dim(self):
        return self.z.dim()

    def materialize(self):
        # (memoised: a node with several consumers that cannot apply the tail themselves is normalised once)
        if self._mat is None:
            self._mat = _ApplyTail.apply(self.z, self.scale, self.shift, self.act)
        return self._mat


class _ApplyTail(torch.autograd.Function):
    """The pass a deferred tail avoided: y = act(scale*z + shift); the gradient goes through unchanged (see
    Pending: z's slot carries dL/dy)."""

    @staticmethod
    def forward(ctx, z, scale, shift, act):
        return _affine_act(_cl(z), scale, shift, None, act)

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None, None


def materialize(x):
    """A plain tensor from a tensor or a Pending."""
    return x.materialize() if isinstance(x, Pending) else x


# Nodes with several consumers (a cell's node read by several ops and sums, a decoder map read by several blocks and
# collect_all: src/nn/micro_decoders.py:95-121,237-251,380-398) go through ONE gradient junction instead of autograd's
# pairwise accumulation (an at::native add launch and three tensor passes per extra consumer) - and a junction over a
# Pending node also does the mask-and-reduce pass of its producer's BatchNorm backward.  NASSEG_JUNCTION=0: as before.
JUNCTION = os.environ.get("NASSEG_JUNCTION", "1") != "0"
_JUNCTION_MAX = 8  # gradients one nasseg_grad_junction launch adds


class _Junction(torch.autograd.Function):
    """Fan a node out to its consumers.  forward(z, stats, act, n_raw, n_fin): n_raw aliases of z for consumers that
    take the node as it is - a plain tensor (stats None), or a conv chain's raw output whose BatchNorm + activation they
    apply on load (the caller wraps them into Pending objects) - and n_fin aliases of the FINISHED map act(scale*z +
    shift), written once, for consumers that need it.  Backward: whatever gradients came back (all of them w.r.t. the
    finished value: Pending's contract) are added by one nasseg_grad_junction launch; over a pending node the sum is
    also multiplied by act' (a contribution that arrived masked stays as it is: the mask is 0 / 1) and comes with the
    producer's BatchNorm-backward sums as rows, handed to the chain's backward by the side of the gradient
    (_TAIL_ROWS) - which then skips its own pass over gradient and z."""

    @staticmethod
    def forward(ctx, z, stats, act, n_raw, n_fin):
        z = _cl(z)
        C = z.shape[1]
        fin = None
        if n_fin:
            fin = z if stats is None else _affine_act(z, stats[2 * C:3 * C], stats[3 * C:], None, act)
        outs = [z.view_as(z) for _ in range(n_raw)] + [fin.view_as(fin) for _ in range(n_fin)]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z if stats is not None else None, stats)
        ctx.cfg = (int(act), tuple(z.shape), z.dtype, z.device)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        z, stats = ctx.saved_tensors
        act, shape, dtype, device = ctx.cfg
        live = [_cl(g) for g in grads if g is not None]
        if not live:
            return None, None, None, None, None
        if len(live) == 1 and stats is None:
            return live[0], None, None, None, None
        B, C, H, W = shape
        s = current_stream()
        nrows = lib.query("nasseg_cat_src_blocks", B, H, W, C)
        if nrows <= 0:  # (a width the row layout does not serve: C / 4 > 256)
            out = live[0]
            for g in live[1:]:
                out = _axpby(out, g, None, None)
            return out, None, None, None, None
        # more than eight consumers: partial sums first (plain adds), the mask and the rows with the last launch
        while len(live) > _JUNCTION_MAX:
            head, live = live[:_JUNCTION_MAX], live[_JUNCTION_MAX:]
            part_sum = torch.empty_like(head[0])
            lib.call(_k("nasseg_grad_junction", head[0]), *([ptr(g) for g in head] + [len(head), None, None, ACT_NONE,
                     ptr(part_sum), None, B, H, W, C, s]))
            live.insert(0, part_sum)
        # (a pending node whose only gradient came from one consumer: the mask and the rows alone, n = 1)
        out = torch.empty(shape, device=device, dtype=dtype, memory_format=torch.channels_last)
        rows = _ws(out, (nrows + 64) * 2 * C) if (stats is not None and FUSE_TAIL_ROWS) else None
        args = [ptr(g) for g in live] + [None] * (_JUNCTION_MAX - len(live))
        lib.call(_k("nasseg_grad_junction", out), *(args + [len(live), ptr(z) if stats is not None else None,
                 ptr(stats), act, ptr(out), ptr(rows), B, H, W, C, s]))
        if rows is not None:
            for key in [k for k, e in _TAIL_ROWS.items() if e[0]() is None]:
                del _TAIL_ROWS[key]
            _TAIL_ROWS[out.data_ptr()] = (weakref.ref(out), rows, nrows, out._version)
        return out, None, None, None, None


def fan_out(x, n_raw, n_fin=0):
    """Handles of a node for its consumers: ``n_raw`` that take it as it is (x itself, or - x a Pending - Pending
    objects whose tail the consumer applies on load) followed by ``n_fin`` finished maps (plain tensors).  With one
    consumer in all, a width the junction kernel does not serve, no gradient to route, or NASSEG_JUNCTION=0 the
    handles are x itself / its memoised materialisation: autograd accumulates as it always did."""
    n = n_raw + n_fin
    pending = isinstance(x, Pending)
    z = x.z if pending else x
    if (n < 2 or not JUNCTION or z.dim() != 4 or z.shape[1] % 4 != 0 or not torch.is_grad_enabled()
            or not z.requires_grad):
        return [x] * n_raw + [materialize(x)] * n_fin if n_fin else [x] * n_raw
    if pending:
        outs = _Junction.apply(x.z, x.stats, x.act, n_raw, n_fin)
        return [Pending(t, x.stats, x.act) for t in outs[:n_raw]] + list(outs[n_raw:])
    return list(_Junction.apply(x, None, ACT_NONE, n, 0))


def conv_chain(x, ops, in_act0=ACT_NONE, residual=None, pool=None, defer_tail=False):
    """ops: list of (weight, stride, padding, dilation, depthwise, bn, act) where ``bn`` is None
    or (gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps).
    pool = (3, stride, 1): 3x3 max pooling of the chain's output (which must end in a BatchNorm without
    activation and without residual - the reference's Pool op), fused behind it.
    defer_tail: return a Pending (raw conv output + the last BatchNorm's scale/shift + activation) instead of
    the normalised output when the chain ends in a BatchNorm that is not folded (no residual, no pooling)."""
    cfg_ops, tensors = [], []
    for weight, stride, padding, dilation, depthwise, bn, act in ops:
        if bn is None:
            cfg_ops.append(("dw" if depthwise else "dense", int(stride), int(padding), int(dilation),
                            False, ACT_NONE, False, 0.0, 0.0))
            tensors.extend([weight, None, None, None, None, None])
        else:
            gamma, beta, rm, rv, nbt, training, momentum, eps = bn
            cfg_ops.append(("dw" if depthwise else "dense", int(stride), int(padding), int(dilation),
                            True, int(act), bool(training), float(momentum), float(eps)))
            tensors.extend([weight, gamma, beta, rm, rv, nbt if training else None])
    cfg = (int(in_act0), tuple(cfg_ops), torch.is_grad_enabled())
    if isinstance(x, Pending):
        if in_act0 != ACT_NONE or not cfg_ops:
            x = x.materialize()  # (an activation on top of a pending one: not fused)
        else:
            # cfg[3] pool, cfg[4] deferred tail, cfg[5] the pending input's activation (its statistics vector
            # rides behind the per-op tensors)
            defer = bool(defer_tail and residual is None and pool is None and cfg_ops[-1][4])
            full = cfg + ((int(pool[0]), int(pool[1]), int(pool[2])) if pool is not None else None, defer, x.act)
            out = _ConvChain.apply(full, x.z, residual, *(tensors + [x.stats]))
            if not defer:
                return out
            y, stats = out
            return y if stats is None else Pending(y, stats, cfg_ops[-1][5])
    if pool is not None:
        cfg = cfg + ((int(pool[0]), int(pool[1]), int(pool[2])),)
    elif defer_tail and residual is None and cfg_ops and cfg_ops[-1][4]:
        y, stats = _ConvChain.apply(cfg + (None, True), x, residual, *tensors)
        return y if stats is None else Pending(y, stats, cfg_ops[-1][5])
    return _ConvChain.apply(cfg, x, residual, *tensors)


def conv_bn_act(x, weight, gamma, beta, running_mean, running_var, num_batches_tracked, training,
                momentum=0.1, eps=1e-5, act=ACT_NONE, residual=None, stride=1, padding=0, dilation=1):
    """act(BN(conv(x))) (+ residual): a conv chain of one link (statistics in the conv epilogue;
    in inference without grad the BatchNorm is folded into the conv's epilogue - one kernel)."""
    bn = (gamma, beta, running_mean, running_var, num_batches_tracked, bool(training), momentum, eps)
    return conv_chain(x, [(weight, stride, padding, dilation, False, bn, int(act))], ACT_NONE, residual)


# ---------------------------------------------------------------------------
# BatchNorm (+ activation, + residual)
# ---------------------------------------------------------------------------
class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act,
                residual):
        x = _cl(x)
        B, C, H, W = x.shape
        M = B * H * W
        s = current_stream()
        stats = _vec(x, 4 * C)  # mean | invstd | scale | shift
        mean, invstd, scale, shift = stats[0:C], stats[C:2 * C], stats[2 * C:3 * C], stats[3 * C:]
        if training:
            if M <= 1:
                # same condition and exception class as torch.nn.functional.batch_norm
                raise ValueError(
                    "Expected more than 1 value per channel when training, got input size {}".format(
                        tuple(x.shape)))
            ws = _ws(x, lib.query("nasseg_colred_workspace", 1, M, C))
            lib.call(_k("nasseg_bn_stats", x), ptr(x), C, M, C, float(eps), float(momentum), ptr(gamma),
                     ptr(beta), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(running_mean),
                     ptr(running_var), ptr(nbt), ptr(ws), s)
        else:
            lib.call("nasseg_bn_eval_params", C, float(eps), ptr(gamma), ptr(beta),
                     ptr(running_mean), ptr(running_var), ptr(mean), ptr(invstd), ptr(scale),
                     ptr(shift), s)
        res = _cl(residual) if residual is not None else None
        y = _affine_act(x, scale, shift, res, act)
        ctx.save_for_backward(x, stats)
        ctx.cfg = (bool(training), act, residual is not None, gamma is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        training, act, has_res, affine = ctx.cfg
        dy = _cl(dy)
        B, C, H, W = x.shape
        M = B * H * W
        mean, invstd, scale, shift = stats[0:C], stats[C:2 * C], stats[2 * C:3 * C], stats[3 * C:]
        s = current_stream()
        sums = _vec(x, 2 * C)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        rows = _bn_bwd_reduce(dy, x, scale, shift, mean, invstd, act, sums, M, C, dx is not None and C % 4 == 0)
        if dx is not None:
            _bn_bwd_apply(dy, x, scale, shift, mean, invstd, sums, M, C, training, act, dx, rows)
        dgamma = sums[C:2 * C] if (affine and ctx.needs_input_grad[1]) else None
        dbeta = sums[0:C] if (affine and ctx.needs_input_grad[2]) else None
        dres = dy if (has_res and ctx.needs_input_grad[10]) else None
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, dres


def batch_norm_act(x, gamma, beta, running_mean, running_var, num_batches_tracked, training,
                   momentum=0.1, eps=1e-5, act=ACT_NONE, residual=None):
    """y = act(BN(x)) (+ residual).  Training mode updates the running buffers in place."""
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked,
                               bool(training), momentum, eps, int(act), residual)


# ---------------------------------------------------------------------------
# pooling
# ---------------------------------------------------------------------------
class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode, k, stride, pad):
        x = _cl(x)
        B, C, H, W = x.shape
        Ho, Wo = conv_out_size(H, k, stride, pad, 1), conv_out_size(W, k, stride, pad, 1)
        y = _new(x, B, C, Ho, Wo)
        idx = None
        if mode == 0:
            idx = torch.empty((B, Ho, Wo, C), device=x.device, dtype=torch.uint8)
        lib.call(_k("nasseg_pool_fwd", x), mode, ptr(x), ptr(y), ptr(idx), B, H, W, C, Ho, Wo, k, stride,
                 pad, current_stream())
        ctx.cfg = (mode, k, stride, pad, (B, C, H, W))
        ctx.idx = idx
        return y

    @staticmethod
    def backward(ctx, dy):
        mode, k, stride, pad, (B, C, H, W) = ctx.cfg
        dy = _cl(dy)
        dx = _new(dy, B, C, H, W)
        lib.call(_k("nasseg_pool_bwd", dy), mode, ptr(dy), ptr(ctx.idx), ptr(dx), B, H, W, C, dy.shape[2],
                 dy.shape[3], k, stride, pad, current_stream())
        return dx, None, None, None, None


def max_pool2d(x, kernel_size=3, stride=1, padding=1):
    return _Pool.apply(x, 0, int(kernel_size), int(stride), int(padding))


def avg_pool2d(x, kernel_size=3, stride=1, padding=1):
    """count_include_pad=False semantics."""
    return _Pool.apply(x, 1, int(kernel_size), int(stride), int(padding))


# ---------------------------------------------------------------------------
# bilinear resize, concat
# ---------------------------------------------------------------------------
class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x = _cl(x)
        B, C, H, W = x.shape
        y = _new(x, B, C, Ho, Wo)
        lib.call(_k("nasseg_bilinear_fwd", x), ptr(x), ptr(y), C, 0, B, H, W, C, Ho, Wo, ACT_NONE,
                 current_stream())
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _cl(dy)
        dx = _new(dy, B, C, H, W)
        nws = lib.query("nasseg_bilinear_bwd_workspace", B, H, W, C, dy.shape[2], dy.shape[3])
        lib.call(_k("nasseg_bilinear_bwd", dy), ptr(dy), C, 0, ptr(dx), B, H, W, C, dy.shape[2],
                 dy.shape[3], ptr(_ws(dy, nws)) if nws else None, current_stream())
        return dx, None, None


def bilinear_resize(x, size, align_corners=False):
    """nn.Upsample(size, mode='bilinear') / F.interpolate(..., align_corners=False); align_corners=True
    (the distillation teacher's decoder, src/kd/rf_lw/model_lw_v2.py:258) is forward-only."""
    Ho, Wo = int(size[0]), int(size[1])
    if tuple(x.shape[2:]) == (Ho, Wo):
        return x
    if align_corners:
        if x.requires_grad and torch.is_grad_enabled():
            raise NassegError("bilinear_resize(align_corners=True) has no backward (inference-only)")
        x = _cl(x)
        B, C, H, W = x.shape
        if C % 4 != 0:
            raise NassegError("bilinear_resize(align_corners=True): C % 4 != 0")
        y = _new(x, B, C, Ho, Wo)
        lib.call(_k("nasseg_bilinear_ac_fwd", x), ptr(x), ptr(y), B, H, W, C, Ho, Wo, current_stream())
        return y
    return _Bilinear.apply(x, Ho, Wo)


class _ConcatResize(torch.autograd.Function):
    """cat(dim=1) of tensors, each bilinearly resized to (Ho, Wo) when needed,
    written straight into the output slab, with an optional fused ReLU."""

    @staticmethod
    def forward(ctx, Ho, Wo, act, *xs):
        xs = [_cl(x) for x in xs]
        B = xs[0].shape[0]
        Ct = sum(x.shape[1] for x in xs)
        y = _new(xs[0], B, Ct, Ho, Wo)
        s = current_stream()
        off = 0
        shapes = []
        for x in xs:
            _, C, H, W = x.shape
            if x.shape[0] != B:
                raise NassegError("concat: batch sizes differ")
            if (H, W) == (Ho, Wo):
                lib.call(_k("nasseg_chan_copy", x), ptr(x), C, 0, ptr(y), Ct, off, None, 0, 0, B * Ho * Wo,
                         C, act, ACT_NONE, s)
            else:
                lib.call(_k("nasseg_bilinear_fwd", x), ptr(x), ptr(y), Ct, off, B, H, W, C, Ho, Wo, act, s)
            shapes.append((C, H, W))
            off += C
        ctx.shapes = shapes
        ctx.act = act
        if act != ACT_NONE:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _cl(dy)
        B, Ct, Ho, Wo = dy.shape
        s = current_stream()
        if ctx.act != ACT_NONE:
            (y,) = ctx.saved_tensors
            dy = _act_bwd(dy, y, ctx.act)
        grads = []
        off = 0
        for i, (C, H, W) in enumerate(ctx.shapes):
            if not ctx.needs_input_grad[3 + i]:
                grads.append(None)
                off += C
                continue
            dx = _new(dy, B, C, H, W)
            if (H, W) == (Ho, Wo):
                lib.call(_k("nasseg_chan_copy", dy), ptr(dy), Ct, off, ptr(dx), C, 0, None, 0, 0,
                         B * Ho * Wo, C, ACT_NONE, ACT_NONE, s)
            else:
                nws = lib.query("nasseg_bilinear_bwd_workspace", B, H, W, C, Ho, Wo)
                lib.call(_k("nasseg_bilinear_bwd", dy), ptr(dy), Ct, off, ptr(dx), B, H, W, C, Ho, Wo,
                         ptr(_ws(dy, nws)) if nws else None, s)
            grads.append(dx)
            off += C
        return (None, None, None) + tuple(grads)


def concat_resize(tensors, size, relu=False):
    return _ConcatResize.apply(int(size[0]), int(size[1]), ACT_RELU if relu else ACT_NONE, *tensors)


class _CatBNReluConv(torch.autograd.Function):
    """ConcatReduce's tail, cat(x, y) -> BatchNorm(2C) -> ReLU -> 1x1 conv (2C -> N)
    (src/nn/layer_factory.py:369-382), WITHOUT the concatenation: BatchNorm is per channel and
    the 1x1 conv is linear in its input channels, so

        out = W[:, :C] . relu(bn_lo(x)) + W[:, C:] . relu(bn_hi(y))

    Two pointwise convs, each applying its half of the BatchNorm on load (the second adds the
    first's output in its epilogue); neither the 2C-channel slab nor its normalised copy is ever
    written, and the backward needs no slicing: each half's backward-data kernel emits the
    masked gradient with its BatchNorm-backward sums.  Used for large maps (a few more, smaller
    launches than the slab path)."""

    @staticmethod
    def forward(ctx, x, y, gamma, beta, rm, rv, nbt, weight, training, momentum, eps, grad_mode=True):
        x, y = _cl(x), _cl(y)
        B, C, H, W = x.shape
        N = weight.shape[0]
        w = weight.contiguous()
        if tuple(y.shape) != (B, C, H, W) or tuple(w.shape) != (N, 2 * C, 1, 1):
            raise NassegError("cat_bn_relu_conv: shapes {} {} {}".format(
                tuple(x.shape), tuple(y.shape), tuple(w.shape)))
        M = B * H * W
        s = current_stream()
        needs_grad = grad_mode and any(ctx.needs_input_grad)  # (grad_mode: the caller's, see _ConvChain)
        stats = _vec(x, 8 * C)  # [mean | invstd | scale | shift] x [2C]
        mean, invstd, scale, shift = (stats[0:2 * C], stats[2 * C:4 * C], stats[4 * C:6 * C],
                                      stats[6 * C:8 * C])
        if training:
            if M <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input "
                                 "size {}".format((B, 2 * C, H, W)))
            ws = _ws(x, lib.query("nasseg_colred_workspace", 1, M, C))
            for h, t in enumerate((x, y)):
                lo, hi = h * C, (h + 1) * C
                lib.call(_k("nasseg_bn_stats", t), ptr(t), C, M, C, float(eps), float(momentum), ptr(gamma[lo:hi]),
                         ptr(beta[lo:hi]), ptr(mean[lo:hi]), ptr(invstd[lo:hi]), ptr(scale[lo:hi]),
                         ptr(shift[lo:hi]), ptr(rm[lo:hi]) if rm is not None else None,
                         ptr(rv[lo:hi]) if rv is not None else None,
                         ptr(nbt) if (nbt is not None and h == 0) else None, ptr(ws), s)
        else:
            lib.call("nasseg_bn_eval_params", 2 * C, float(eps), ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
                     ptr(mean), ptr(invstd), ptr(scale), ptr(shift), s)
        items = [(w, 0, 0, C), (w, 0, C, C)]
        if needs_grad:
            items += [(w, 1, 0, C), (w, 1, C, C)]
        packed = _pack_many(x, items)
        y1 = _new(x, B, N, H, W)
        lib.call(_k("nasseg_conv_fwd", x), ptr(x), C, ptr(packed[0]), ptr(y1), N, ptr(scale[0:C]), ptr(shift[0:C]),
                 ACT_RELU, None, None, ACT_NONE, None, 0, B, H, W, C, H, W, N, 1, 1, 1, 0, 1, 0, None, s)
        out = _new(x, B, N, H, W)
        lib.call(_k("nasseg_conv_fwd", y), ptr(y), C, ptr(packed[1]), ptr(out), N, ptr(scale[C:]), ptr(shift[C:]),
                 ACT_RELU, None, None, ACT_NONE, ptr(y1), N, B, H, W, C, H, W, N, 1, 1, 1, 0, 1, 0, None, s)
        if needs_grad:
            ctx.save_for_backward(x, y, stats, packed[2], packed[3])
            ctx.cfg = (bool(training), N)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, stats, wb_lo, wb_hi = ctx.saved_tensors
        training, N = ctx.cfg
        dout = _cl(dout)
        B, C, H, W = x.shape
        M = B * H * W
        s = current_stream()
        mean, invstd, scale, shift = (stats[0:2 * C], stats[2 * C:4 * C], stats[4 * C:6 * C],
                                      stats[6 * C:8 * C])
        need_w = ctx.needs_input_grad[7]
        need_bn = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        dbn = _vec(x, 4 * C) if need_bn else None  # [dbeta(2C) | dgamma(2C)]
        dw = torch.empty((N, 2 * C, 1, 1), device=x.device, dtype=torch.float32) if need_w else None
        nb = lib.query("nasseg_conv_fwd_stats_blocks", B, H, W, C, N, 2)
        grads_in = [None, None]
        for h, (t, wb) in enumerate(((x, wb_lo), (y, wb_hi))):
            lo, hi = h * C, (h + 1) * C
            need_dx = ctx.needs_input_grad[h]
            sc, sh, mu, isd = scale[lo:hi], shift[lo:hi], mean[lo:hi], invstd[lo:hi]
            if need_dx or need_bn:
                g = _new(x, B, C, H, W)
                part = _ws(x, (nb + 64) * 2 * C)
                lib.call(_k("nasseg_conv_bwd_data_bn", dout), ptr(dout), N, ptr(wb), ptr(g), C, ptr(t), C, ptr(sc),
                         ptr(sh), ptr(mu), ptr(isd), ACT_RELU, B, H, W, N, H, W, C, 1, 1, 1, 0, 1,
                         ptr(part), s)
                sums = _vec(x, 2 * C)
                lib.call("nasseg_rows_sum", ptr(part), nb, 2 * C, ptr(sums), s)
                if need_bn:  # sums = [sum g | sum g*xhat] -> rows (dbeta, dgamma) of dbn at columns lo..hi
                    lib.call(_k("nasseg_chan_copy", sums), ptr(sums), C, 0, ptr(dbn), 2 * C, lo, None, 0, 0, 2, C,
                             ACT_NONE, ACT_NONE, s)
                if need_dx:
                    dz = torch.empty_like(t)
                    lib.call(_k("nasseg_bn_bwd_apply", g), ptr(g), ptr(t), ptr(sc), ptr(sh), ptr(mu), ptr(isd),
                             ptr(sums), M, C, int(training), ACT_RELU, ptr(dz), s)
                    grads_in[h] = dz
            if need_w:
                dwh = _vec(x, N * C)
                ws = _ws(x, lib.query("nasseg_conv_wgrad_workspace", B, H, W, N, C, 1, 1))
                lib.call(_k("nasseg_conv_wgrad", t), ptr(t), C, ptr(dout), N, ptr(dwh), ptr(ws), ptr(sc), ptr(sh),
                         ACT_RELU, B, H, W, C, H, W, N, 1, 1, 1, 0, 1, s)
                lib.call(_k("nasseg_chan_copy", dwh), ptr(dwh), C, 0, ptr(dw), 2 * C, lo, None, 0, 0, N, C, ACT_NONE,
                         ACT_NONE, s)
        dgamma = dbn[2 * C:4 * C] if ctx.needs_input_grad[2] else None
        dbeta = dbn[0:2 * C] if ctx.needs_input_grad[3] else None
        return (grads_in[0], grads_in[1], dgamma, dbeta, None, None, None, dw, None, None, None, None)


def cat_bn_relu_conv(x, y, gamma, beta, running_mean, running_var, num_batches_tracked, weight,
                     training, momentum=0.1, eps=1e-5):
    """conv1x1(relu(batch_norm(cat([x, y], 1)))) without materialising the concatenation."""
    return _CatBNReluConv.apply(x, y, gamma, beta, running_mean, running_var, num_batches_tracked,
                                weight, bool(training), float(momentum), float(eps), torch.is_grad_enabled())


class _BNReluConv(torch.autograd.Function):
    """BatchNorm -> ReLU -> 1x1 conv over ONE tensor (ConcatReduce's tail on its concat slab,
    src/nn/layer_factory.py:369-382, below the size where the slab is avoided altogether): the conv applies
    the BatchNorm on load, so the normalised slab is never written; backward, the backward-data kernel emits
    the masked gradient with the BatchNorm-backward sums (no reduction pass over gradient and slab).  The
    single-input form of _CatBNReluConv."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, weight, training, momentum, eps, grad_mode):
        x = _cl(x)
        B, C, H, W = x.shape
        N = weight.shape[0]
        w = weight.contiguous()
        if tuple(w.shape) != (N, C, 1, 1):
            raise NassegError("bn_relu_conv: shapes {} {}".format(tuple(x.shape), tuple(w.shape)))
        M = B * H * W
        s = current_stream()
        needs_grad = grad_mode and any(ctx.needs_input_grad)
        stats = _vec(x, 4 * C)  # mean | invstd | scale | shift
        mean, invstd, scale, shift = stats[0:C], stats[C:2 * C], stats[2 * C:3 * C], stats[3 * C:]
        if training:
            if M <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input "
                                 "size {}".format((B, C, H, W)))
            ws = _ws(x, lib.query("nasseg_colred_workspace", 1, M, C))
            lib.call(_k("nasseg_bn_stats", x), ptr(x), C, M, C, float(eps), float(momentum), ptr(gamma), ptr(beta),
                     ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(rm), ptr(rv), ptr(nbt), ptr(ws), s)
        else:
            lib.call("nasseg_bn_eval_params", C, float(eps), ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
                     ptr(mean), ptr(invstd), ptr(scale), ptr(shift), s)
        items = [(w, 0)]
        if needs_grad:
            items.append((w, 1))
        packed = _pack_many(x, items)
        out = _new(x, B, N, H, W)
        lib.call(_k("nasseg_conv_fwd", x), ptr(x), C, ptr(packed[0]), ptr(out), N, ptr(scale), ptr(shift),
                 ACT_RELU, None, None, ACT_NONE, None, 0, B, H, W, C, H, W, N, 1, 1, 1, 0, 1, 0, None, s)
        if needs_grad:
            ctx.save_for_backward(x, stats, packed[1], w)
            ctx.cfg = (bool(training), N)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, stats, wb, w = ctx.saved_tensors
        training, N = ctx.cfg
        dout = _cl(dout)
        B, C, H, W = x.shape
        M = B * H * W
        s = current_stream()
        mean, invstd, scale, shift = stats[0:C], stats[C:2 * C], stats[2 * C:3 * C], stats[3 * C:]
        need_dx = ctx.needs_input_grad[0]
        need_bn = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx = dgamma = dbeta = dw = None
        if need_dx or need_bn:
            nb = lib.query("nasseg_conv_fwd_stats_blocks", B, H, W, C, N, 2)
            g = _new(x, B, C, H, W)
            part = _ws(x, (nb + 64) * 2 * C)
            lib.call(_k("nasseg_conv_bwd_data_bn", dout), ptr(dout), N, ptr(wb), ptr(g), C, ptr(x), C, ptr(scale),
                     ptr(shift), ptr(mean), ptr(invstd), ACT_RELU, B, H, W, N, H, W, C, 1, 1, 1, 0, 1, ptr(part), s)
            sums = _vec(x, 2 * C)
            lib.call("nasseg_rows_sum", ptr(part), nb, 2 * C, ptr(sums), s)
            if ctx.needs_input_grad[1]:
                dgamma = sums[C:2 * C]
            if ctx.needs_input_grad[2]:
                dbeta = sums[0:C]
            if need_dx:
                dx = torch.empty_like(x)
                lib.call(_k("nasseg_bn_bwd_apply", g), ptr(g), ptr(x), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                         ptr(sums), M, C, int(training), ACT_NONE, ptr(dx), s)
        if ctx.needs_input_grad[6]:
            dw = _dense_wgrad(x, dout, w, scale, shift, ACT_RELU, (B, H, W, C, H, W, N, 1, 1, 1, 0, 1))
        return dx, dgamma, dbeta, None, None, None, dw, None, None, None, None


class _CatReduce(torch.autograd.Function):
    """ConcatReduce whole (src/nn/layer_factory.py:369-382 with Adapt's resize, :338-350) as ONE node:
    cat(x, y) -> BatchNorm(2C) -> ReLU -> 1x1 conv.  Each input is written into its half of the slab by one
    launch (nasseg_cat_src_fwd) that resizes it when its size differs, applies the producer's pending
    BatchNorm + activation on load (Pending: the producers' normalised outputs are never written) and emits
    the slab's BatchNorm statistics as partial rows - no pass over the slab for them; the conv applies the
    slab's BatchNorm + ReLU on load (_BNReluConv).  Backward: the backward-data kernel leaves the masked
    gradient + the slab BatchNorm's sums; nasseg_cat_src_bwd then applies that BatchNorm's backward per input
    slice (no full-width slab gradient), and for a pending input of the slab's size also masks with its
    activation's derivative and emits the producer's BatchNorm-backward sums (_TAIL_ROWS).  The gradient
    returned for a Pending input is the one w.r.t. its activated output (the producer chain's backward takes
    it from there).

    cfg = (Ho, Wo, act_x, act_y, training, momentum, eps, grad_mode); xst / yst: the producers' statistics
    vectors (mean | invstd | scale | shift) or None."""

    @staticmethod
    def forward(ctx, cfg, x, y, xst, yst, gamma, beta, rm, rv, nbt, weight):
        Ho, Wo, act_x, act_y, training, momentum, eps, grad_mode = cfg
        x, y = _cl(x), _cl(y)
        B, C = x.shape[0], x.shape[1]
        Ct = 2 * C
        N = weight.shape[0]
        w = weight.contiguous()
        if y.shape[0] != B or y.shape[1] != C or tuple(w.shape) != (N, Ct, 1, 1):
            raise NassegError("cat_reduce: shapes {} {} {}".format(tuple(x.shape), tuple(y.shape), tuple(w.shape)))
        M = B * Ho * Wo
        s = current_stream()
        needs_grad = grad_mode and any(ctx.needs_input_grad)
        stats = _vec(x, 4 * Ct)  # mean | invstd | scale | shift
        mean, invstd, scale, shift = stats[0:Ct], stats[Ct:2 * Ct], stats[2 * Ct:3 * Ct], stats[3 * Ct:]
        if training and M <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input "
                             "size {}".format((B, Ct, Ho, Wo)))
        slab = _new(x, B, Ct, Ho, Wo)
        nblk = lib.query("nasseg_cat_src_blocks", B, Ho, Wo, C)
        part = _ws(x, (nblk + 64) * 2 * Ct) if training else None
        for off, (t, st, act) in enumerate(((x, xst, act_x), (y, yst, act_y))):
            sc, sh = (st[2 * C:3 * C], st[3 * C:]) if st is not None else (None, None)
            lib.call(_k("nasseg_cat_src_fwd", t), ptr(t), ptr(sc), ptr(sh), act if st is not None else ACT_NONE,
                     ptr(slab), Ct, off * C, ptr(part), B, t.shape[2], t.shape[3], C, Ho, Wo, s)
        if training:
            lib.call("nasseg_bn_finalize", ptr(part), nblk, M, Ct, float(eps), float(momentum), ptr(gamma),
                     ptr(beta), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(rm), ptr(rv), ptr(nbt), s)
        else:
            lib.call("nasseg_bn_eval_params", Ct, float(eps), ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
                     ptr(mean), ptr(invstd), ptr(scale), ptr(shift), s)
        items = [(w, 0)]
        if needs_grad:
            items.append((w, 1))
        packed = _pack_many(x, items)
        out = _new(x, B, N, Ho, Wo)
        lib.call(_k("nasseg_conv_fwd", slab), ptr(slab), Ct, ptr(packed[0]), ptr(out), N, ptr(scale), ptr(shift),
                 ACT_RELU, None, None, ACT_NONE, None, 0, B, Ho, Wo, Ct, Ho, Wo, N, 1, 1, 1, 0, 1, 0, None, s)
        if needs_grad:
            # (x / y: the producers' raw outputs - they hold them for their own backward anyway)
            ctx.save_for_backward(slab, stats, packed[1], w, x if xst is not None else None,
                                  y if yst is not None else None, xst, yst)
            ctx.cfg = (bool(training), N, tuple(x.shape), tuple(y.shape), act_x, act_y)
        return out

    @staticmethod
    def backward(ctx, dout):
        for key in [k for k, e in _TAIL_ROWS.items() if e[0]() is None]:
            del _TAIL_ROWS[key]  # (rows whose gradient tensor died unconsumed: a backward that stopped short)
        slab, stats, wb, w, zx, zy, xst, yst = ctx.saved_tensors
        training, N, x_shape, y_shape, act_x, act_y = ctx.cfg
        dout = _cl(dout)
        B, Ct, Ho, Wo = slab.shape
        C = Ct // 2
        s = current_stream()
        mean, invstd, scale, shift = stats[0:Ct], stats[Ct:2 * Ct], stats[2 * Ct:3 * Ct], stats[3 * Ct:]
        need_in = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        need_bn = ctx.needs_input_grad[5] or ctx.needs_input_grad[6]
        dx = dy = dgamma = dbeta = dw = None
        if need_in or need_bn:
            nb = lib.query("nasseg_conv_fwd_stats_blocks", B, Ho, Wo, Ct, N, 2)
            g = _new(slab, B, Ct, Ho, Wo)
            part = _ws(slab, (nb + 64) * 2 * Ct)
            lib.call(_k("nasseg_conv_bwd_data_bn", dout), ptr(dout), N, ptr(wb), ptr(g), Ct, ptr(slab), Ct,
                     ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ACT_RELU, B, Ho, Wo, N, Ho, Wo, Ct, 1, 1, 1, 0,
                     1, ptr(part), s)
            sums = _vec(slab, 2 * Ct)
            lib.call("nasseg_rows_sum", ptr(part), nb, 2 * Ct, ptr(sums), s)
            if ctx.needs_input_grad[5]:
                dgamma = sums[Ct:2 * Ct]
            if ctx.needs_input_grad[6]:
                dbeta = sums[0:Ct]
            if need_in:
                nrows = lib.query("nasseg_cat_src_blocks", B, Ho, Wo, C)
                grads = []
                for off, (need, shp, z, st, act) in enumerate(((ctx.needs_input_grad[1], x_shape, zx, xst, act_x),
                                                               (ctx.needs_input_grad[2], y_shape, zy, yst, act_y))):
                    if not need:
                        grads.append(None)
                        continue
                    _, _, H, W = shp
                    same = (H, W) == (Ho, Wo)
                    # the producer's mask and BatchNorm-backward sums ride along: directly when its output has the
                    # slab's size; behind a resize the sums are formed at the slab's size against the interpolated
                    # mask and nasseg_bilinear_bwd_act masks the gradient it transposes
                    # (a producer SMALLER than the slab keeps its own reduction pass: over its few pixels that is
                    #  cheaper than four taps of z per slab pixel - 3 launches of the headline step, +15 us each)
                    rows = (_ws(slab, (nrows + 64) * 2 * C)
                            if (st is not None and FUSE_TAIL_ROWS and H * W >= Ho * Wo) else None)
                    d = _new(slab, B, C, Ho, Wo)
                    lib.call(_k("nasseg_cat_src_bwd", g), ptr(g), ptr(slab), Ct, off * C, ptr(scale), ptr(mean),
                             ptr(invstd), ptr(sums), int(training), ptr(z) if rows is not None else None,
                             ptr(st) if rows is not None else None, act, ptr(d), ptr(rows), B, Ho, Wo, C,
                             H if rows is not None else Ho, W if rows is not None else Wo, s)
                    if not same:
                        full = _new(slab, B, C, H, W)
                        nws = lib.query("nasseg_bilinear_bwd_workspace", B, H, W, C, Ho, Wo)
                        ws = ptr(_ws(d, nws)) if nws else None
                        if rows is not None:
                            lib.call(_k("nasseg_bilinear_bwd_act", d), ptr(d), C, 0, ptr(z), ptr(st[2 * C:3 * C]),
                                     ptr(st[3 * C:]), act, ptr(full), B, H, W, C, Ho, Wo, ws, s)
                        else:
                            lib.call(_k("nasseg_bilinear_bwd", d), ptr(d), C, 0, ptr(full), B, H, W, C, Ho, Wo, ws, s)
                        d = full
                    if rows is not None:
                        _TAIL_ROWS[d.data_ptr()] = (weakref.ref(d), rows, nrows, d._version)
                    grads.append(d)
                dx, dy = grads
        if ctx.needs_input_grad[10]:
            dw = _dense_wgrad(slab, dout, w, scale, shift, ACT_RELU, (B, Ho, Wo, Ct, Ho, Wo, N, 1, 1, 1, 0, 1))
        return None, dx, dy, None, None, dgamma, dbeta, None, None, None, dw


def cat_reduce_ok(x, y, weight):
    """Shapes _CatReduce serves: both inputs of the same width C (a multiple of 4; what Adapt leaves), a 1x1
    conv over 2C channels with N % 4 == 0 outputs."""
    C = x.shape[1]
    return (FUSE_CAT_REDUCE and C == y.shape[1] and C % 4 == 0 and C // 4 <= 256 and weight.shape[0] % 4 == 0
            and x.shape[0] == y.shape[0])


def cat_reduce(x, y, size, gamma, beta, running_mean, running_var, num_batches_tracked, weight, training,
               momentum=0.1, eps=1e-5):
    """conv1x1(relu(batch_norm(cat(resize(x), resize(y))))) with ``size`` the common (H, W); x / y: tensors or
    Pending outputs of conv chains (their BatchNorm + activation are then applied as the slab is written)."""
    parts = []
    for t in (x, y):
        parts.append((t.z, t.stats, t.act) if isinstance(t, Pending) else (t, None, ACT_NONE))
    cfg = (int(size[0]), int(size[1]), parts[0][2], parts[1][2], bool(training), float(momentum), float(eps),
           torch.is_grad_enabled())
    return _CatReduce.apply(cfg, parts[0][0], parts[1][0], parts[0][1], parts[1][1], gamma, beta, running_mean,
                            running_var, num_batches_tracked, weight)


def bn_relu_conv(x, gamma, beta, running_mean, running_var, num_batches_tracked, weight, training,
                 momentum=0.1, eps=1e-5):
    """conv1x1(relu(batch_norm(x))) without materialising the normalised tensor."""
    return _BNReluConv.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, weight,
                             bool(training), float(momentum), float(eps), torch.is_grad_enabled())


# ---------------------------------------------------------------------------
# elementwise
# ---------------------------------------------------------------------------
class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _cl(a), _cl(b)
        if a.shape != b.shape:
            raise NassegError("add: shapes {} and {} differ".format(tuple(a.shape), tuple(b.shape)))
        return _axpby(a, b, None, None)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class _AddPending(torch.autograd.Function):
    """a + b where one or both are conv-chain outputs with a pending BatchNorm + activation (Pending): the tails
    are applied as the raw conv outputs are loaded (nasseg_add_act2) - one launch and three tensor passes instead
    of up to three launches and seven passes.  Backward: the gradient w.r.t. a sum's operands is the incoming one -
    and a Pending's slot carries the gradient w.r.t. its NORMALISED value (see Pending), so nothing is computed.
    (Masking the operands' gradients here with their chains' BatchNorm-backward rows by the side - nasseg_psum_bwd
    with unit coefficients - was tried in round 5: on the small maps this runs on it leaves 88 - 336 rows per
    operand, too many for the chains' apply kernels to add up themselves, and the launch count went UP by 4.)"""

    @staticmethod
    def forward(ctx, za, zb, sta, stb, act_a, act_b):
        za, zb = _cl(za), _cl(zb)
        if za.shape != zb.shape:
            raise NassegError("add: shapes {} and {} differ".format(tuple(za.shape), tuple(zb.shape)))
        C = za.shape[1]
        y = torch.empty_like(za)

        def vecs(st):
            return (None, None) if st is None else (st[2 * C:3 * C], st[3 * C:])

        (sa, ha), (sb, hb) = vecs(sta), vecs(stb)
        lib.call(_k("nasseg_add_act2", za), ptr(za), ptr(sa), ptr(ha), act_a, None, ptr(zb), ptr(sb), ptr(hb), act_b,
                 None, ptr(y), za.numel(), C, current_stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy, None, None, None, None


FUSE_PENDING_ADD = os.environ.get("NASSEG_PENDING_ADD", "1") != "0"


def add(a, b):
    """a + b; operands may be Pending (applied on load) where their width allows the vector kernel"""
    pa, pb = isinstance(a, Pending), isinstance(b, Pending)
    if (pa or pb) and FUSE_PENDING_ADD and a.shape[1] % 4 == 0:
        return _AddPending.apply(a.z if pa else a, b.z if pb else b, a.stats if pa else None, b.stats if pb else None,
                                 a.act if pa else ACT_NONE, b.act if pb else ACT_NONE)
    return _Add.apply(materialize(a), materialize(b))


class _ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        y = _axpby(x, None, None, None, ACT_RELU)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _act_bwd(_cl(dy), y, ACT_RELU)


def relu(x):
    return _ReLU.apply(x)


class _ParamSum(torch.autograd.Function):
    """a[c]*x + b[c]*y  (ParamSum, src/nn/layer_factory.py:353-366)."""

    @staticmethod
    def forward(ctx, x, y, a, b):
        x, y = _cl(x), _cl(y)
        if x.shape != y.shape:
            raise NassegError("psum: shapes {} and {} differ".format(tuple(x.shape), tuple(y.shape)))
        a, b = a.contiguous(), b.contiguous()
        out = _axpby(x, y, a, b)
        ctx.save_for_backward(x, y, a, b)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, y, a, b = ctx.saved_tensors
        dy = _cl(dy)
        B, C, H, W = x.shape
        dx = _axpby(dy, None, a, None) if ctx.needs_input_grad[0] else None
        dyy = _axpby(dy, None, b, None) if ctx.needs_input_grad[1] else None
        da = db = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            sums = _colred(RED_DOT2, dy, C, x, C, y, C, 1, B * H * W, C)
            da, db = sums[0:C], sums[C:2 * C]
        return dx, dyy, da, db


class _ParamSumPending(torch.autograd.Function):
    """a[c]*x + b[c]*y where x and / or y are conv-chain outputs whose last BatchNorm + activation is pending
    (Pending): forward applies the tails as it loads the raw conv outputs (nasseg_add_act2); backward is ONE kernel
    over the gradient (nasseg_psum_bwd) that leaves, per operand, the masked gradient its chain takes together
    with that chain's BatchNorm-backward sums (handed over by the side of the gradient, _TAIL_ROWS), and the rows of
    the coefficient gradients - instead of two scaling passes, a two-dot reduction and two mask-and-reduce passes."""

    @staticmethod
    def forward(ctx, za, zb, sta, stb, a, b, act_a, act_b):
        za, zb = _cl(za), _cl(zb)
        if za.shape != zb.shape:
            raise NassegError("psum: shapes {} and {} differ".format(tuple(za.shape), tuple(zb.shape)))
        C = za.shape[1]
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(za)

        def vecs(st):
            return (None, None) if st is None else (st[2 * C:3 * C], st[3 * C:])

        (sa, ha), (sb, hb) = vecs(sta), vecs(stb)
        lib.call(_k("nasseg_add_act2", za), ptr(za), ptr(sa), ptr(ha), act_a, ptr(a), ptr(zb), ptr(sb), ptr(hb), act_b,
                 ptr(b), ptr(y), za.numel(), C, current_stream())
        ctx.save_for_backward(za, zb, sta, stb, a, b)
        ctx.acts = (act_a, act_b)
        return y

    @staticmethod
    def backward(ctx, dy):
        for key in [k for k, e in _TAIL_ROWS.items() if e[0]() is None]:
            del _TAIL_ROWS[key]
        za, zb, sta, stb, a, b = ctx.saved_tensors
        act_a, act_b = ctx.acts
        dy = _cl(dy)
        B, C, H, W = za.shape
        s = current_stream()
        nrows = lib.query("nasseg_cat_src_blocks", B, H, W, C)
        ga = torch.empty_like(za) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(zb) if ctx.needs_input_grad[1] else None
        rows_a = _ws(za, (nrows + 64) * 2 * C) if (sta is not None and ga is not None and FUSE_TAIL_ROWS) else None
        rows_b = _ws(zb, (nrows + 64) * 2 * C) if (stb is not None and gb is not None and FUSE_TAIL_ROWS) else None
        cpart = _ws(za, (nrows + 64) * 2 * C)
        if (sta is not None and rows_a is None and ga is not None) or (stb is not None and rows_b is None and gb is not None):
            raise NassegError("psum: pending operands need NASSEG_FUSE_TAIL_ROWS")
        lib.call(_k("nasseg_psum_bwd", dy), ptr(dy), ptr(za), ptr(sta), act_a, ptr(a), ptr(ga), ptr(rows_a), ptr(zb),
                 ptr(stb), act_b, ptr(b), ptr(gb), ptr(rows_b), ptr(cpart), B, H, W, C, s)
        da = db = None
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            sums = _vec(za, 2 * C)
            lib.call("nasseg_rows_sum", ptr(cpart), nrows, 2 * C, ptr(sums), s)
            da, db = sums[0:C], sums[C:2 * C]
        for g, rows in ((ga, rows_a), (gb, rows_b)):
            if rows is not None:
                _TAIL_ROWS[g.data_ptr()] = (weakref.ref(g), rows, nrows, g._version)
        return ga, gb, None, None, da, db, None, None


FUSE_PENDING_PSUM = os.environ.get("NASSEG_PENDING_PSUM", "1") != "0"


def param_sum(x, y, a, b):
    """a[c]*x + b[c]*y (ParamSum); x / y may be Pending (same size, C % 4 == 0: applied on load)"""
    px, py = isinstance(x, Pending), isinstance(y, Pending)
    if (px or py) and FUSE_PENDING_PSUM and FUSE_TAIL_ROWS and x.shape[1] % 4 == 0 and torch.is_grad_enabled():
        return _ParamSumPending.apply(x.z if px else x, y.z if py else y, x.stats if px else None,
                                      y.stats if py else None, a, b, x.act if px else ACT_NONE, y.act if py else ACT_NONE)
    return _ParamSum.apply(materialize(x), materialize(y), a, b)


class _ChannelRepeat(torch.autograd.Function):
    """x.repeat(1, rep, 1, 1)  (Skip / Zero, src/nn/layer_factory.py:268-297)."""

    @staticmethod
    def forward(ctx, x, rep):
        x = _cl(x)
        B, C, H, W = x.shape
        y = _new(x, B, C * rep, H, W)
        s = current_stream()
        for r in range(rep):
            lib.call(_k("nasseg_chan_copy", x), ptr(x), C, 0, ptr(y), C * rep, r * C, None, 0, 0, B * H * W,
                     C, ACT_NONE, ACT_NONE, s)
        ctx.cfg = (rep, (B, C, H, W))
        return y

    @staticmethod
    def backward(ctx, dy):
        rep, (B, C, H, W) = ctx.cfg
        dy = _cl(dy)
        dx = _new(dy, B, C, H, W)
        lib.call(_k("nasseg_chan_fold", dy), ptr(dy), ptr(dx), B * H * W, C, rep, current_stream())
        return dx, None


def channel_repeat(x, rep):
    if rep == 1:
        # torch's repeat always copies; values are what matter downstream
        return _ChannelRepeat.apply(x, 1)
    return _ChannelRepeat.apply(x, int(rep))


def zeros(like, B, C, H, W):
    """A zero activation that is not connected to the autograd graph."""
    require_device(like)
    y = _new(like, B, C, H, W)
    lib.call(_k("nasseg_fill", y), ptr(y), y.numel(), 0.0, current_stream())
    return y


class _ZeroOf(torch.autograd.Function):
    """The reference's Zero op multiplies a view of x by 0.0 (src/nn/layer_factory.py:286-297):
    its output is zeros but stays CONNECTED to x, so everything upstream receives a zero
    gradient (not None: weight decay and momentum still act on those parameters, and a cell of
    'none' ops only still back-propagates)."""

    @staticmethod
    def forward(ctx, x, C, H, W):
        ctx.shape = tuple(x.shape)
        return zeros(x, x.shape[0], C, H, W)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        return zeros(dy, B, C, H, W), None, None, None


def zero_of(x, C, H, W):
    """zeros of shape (B, C, H, W) that depend on x with a zero gradient (Zero op)."""
    return _ZeroOf.apply(_cl(x), int(C), int(H), int(W))


# ---------------------------------------------------------------------------
# global average pooling and its broadcast
# ---------------------------------------------------------------------------
class _GlobalAvgPool(torch.autograd.Function):
    """The pooled (B, C, 1, 1) map is ALWAYS fp32, also when the activations are stored in
    bfloat16: what follows it in the reference's op (layer_factory.py:181-195) is a BatchNorm over
    those B values per channel, which amplifies the small differences between the samples' means
    - 8 bits of mantissa would leave nothing of them.  B*C values: storage cost nil."""

    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        B, C, H, W = x.shape
        out = _colred(RED_SUM, x, C, None, 0, None, 0, B, H * W, C, 1.0 / (H * W))
        ctx.shape = (B, C, H, W)
        ctx.dtype = x.dtype
        return out.view(B, C, 1, 1)  # (the reduction is fp32)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _to_dtype(dy.contiguous().view(B, C, 1, 1), ctx.dtype)
        # every pixel receives dy / (H*W): a broadcast with a scale
        scale = _vec(dy, C)
        lib.call("nasseg_fill", ptr(scale), C, 1.0 / (H * W), current_stream())
        dx = _new(dy, B, C, H, W)
        lib.call(_k("nasseg_bilinear_fwd", dy), ptr(dy), ptr(dx), C, 0, B, 1, 1, C, H, W, ACT_NONE,
                 current_stream())
        return _axpby(dx, None, scale, None)


def global_avg_pool(x):
    """x.mean(2, keepdim=True).mean(3, keepdim=True) -> (B, C, 1, 1), fp32 whatever x's storage."""
    return _GlobalAvgPool.apply(x)


class _Broadcast(torch.autograd.Function):
    """Bilinear interpolation from a 1x1 map = broadcast over (H, W); the output is stored as
    ``dtype`` (the activation storage of the network), the gradient w.r.t. v keeps v's dtype."""

    @staticmethod
    def forward(ctx, v, H, W, dtype):
        require_device(v)
        B, C = v.shape[0], v.shape[1]
        ctx.vdtype = v.dtype
        v = _to_dtype(v.contiguous().view(B, C, 1, 1), dtype)
        y = _new(v, B, C, H, W)
        lib.call(_k("nasseg_bilinear_fwd", v), ptr(v), ptr(y), C, 0, B, 1, 1, C, H, W, ACT_NONE,
                 current_stream())
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _cl(dy)
        dv = _colred(RED_SUM, dy, C, None, 0, None, 0, B, H * W, C)  # (fp32 sums)
        return _to_dtype(dv.view(B, C, 1, 1), ctx.vdtype), None, None, None


def broadcast_to(v, size, dtype=None):
    return _Broadcast.apply(v, int(size[0]), int(size[1]), dtype if dtype is not None else v.dtype)


# ---------------------------------------------------------------------------
# loss and reward
# ---------------------------------------------------------------------------
def _label_tensor(target):
    require_device(target)
    if target.dtype == torch.int64:
        return target.contiguous(), 8
    if target.dtype == torch.uint8:
        return target.contiguous(), 1
    raise NassegError("labels must be int64 or uint8 (got {})".format(target.dtype))


class _LogSoftmaxNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        logits = _cl(logits)
        B, C, H, W = logits.shape
        target, esz = _label_tensor(target)
        if tuple(target.shape) != (B, H, W):
            raise NassegError("loss: target {} does not match logits {}".format(
                tuple(target.shape), tuple(logits.shape)))
        out = _vec(logits, 2)
        ws = _ws(logits, lib.query("nasseg_ce_workspace"))
        lib.call(_k("nasseg_ce_fwd", logits), ptr(logits), ptr(target), esz, B * H * W, C, int(ignore_index),
                 ptr(out), ptr(ws), current_stream())
        ctx.save_for_backward(logits, target, out)
        ctx.cfg = (esz, int(ignore_index))
        return out[0]  # (a view, not a clone: a 4-byte copy node in a recorded step cannot be re-created, graph_dag.py)

    @staticmethod
    def backward(ctx, g):
        logits, target, out = ctx.saved_tensors
        esz, ignore = ctx.cfg
        B, C, H, W = logits.shape
        g = g.to(torch.float32).contiguous().view(1)
        d = torch.empty_like(logits)
        lib.call(_k("nasseg_ce_bwd", logits), ptr(logits), ptr(target), esz, ptr(out), ptr(g), B * H * W, C,
                 ignore, ptr(d), current_stream())
        return d, None, None


def log_softmax_nll(logits, target, ignore_index=255):
    """nn.NLLLoss2d(ignore_index)(nn.LogSoftmax()(logits), target) -> 0-dim tensor."""
    return _LogSoftmaxNLL.apply(logits, target, ignore_index)


class _BerHu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        require_device(pred, target)
        if pred.dtype not in (torch.float32, torch.bfloat16) or target.dtype != pred.dtype:
            raise NassegError("berhu: fp32 or bf16 tensors of one dtype expected")
        if pred.shape != target.shape:
            raise NassegError("berhu: shapes differ")
        p, t = pred.contiguous(), target.contiguous()
        out = _vec(p, 2)
        ws = _ws(p, lib.query("nasseg_ce_workspace"))
        lib.call(_k("nasseg_berhu_fwd", p), ptr(p), ptr(t), p.numel(), ptr(out), ptr(ws), current_stream())
        ctx.save_for_backward(p, t, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        p, t, out = ctx.saved_tensors
        g = g.to(torch.float32).contiguous().view(1)
        d = torch.empty_like(p)
        lib.call(_k("nasseg_berhu_bwd", p), ptr(p), ptr(t), ptr(out), ptr(g), p.numel(), ptr(d),
                 current_stream())
        return d, None


def berhu_loss(pred, target):
    """Reverse-Huber loss of the depth head (absent from the reference; oracle/losses.py)."""
    return _BerHu.apply(pred, target)


def nearest_label_resize(target, size, out=None):
    """F.interpolate(target[:, None].float(), size, mode='nearest').long()[:, 0]; ``out``: a
    contiguous int64 (B, Ho, Wo) tensor to write into (a slice of the task0 label cache)."""
    target, esz = _label_tensor(target)
    B, H, W = target.shape
    Ho, Wo = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty((B, Ho, Wo), device=target.device, dtype=torch.int64)
    elif (tuple(out.shape) != (B, Ho, Wo) or out.dtype != torch.int64 or not out.is_contiguous()
          or out.device != target.device):
        raise NassegError("nearest_label_resize: bad output tensor")
    lib.call("nasseg_nearest_label", ptr(target), esz, ptr(out), B, H, W, Ho, Wo, current_stream())
    return out


def copy_into(dst, src):
    """dst[...] = src for two densely stored activations of one shape and dtype (a slice of the
    task0 cache receiving an encoder output): one copy kernel on the current stream."""
    require_device(dst, src)
    if tuple(dst.shape) != tuple(src.shape) or dst.dtype != src.dtype:
        raise NassegError("copy_into: shapes / dtypes differ")
    src = src.contiguous(memory_format=torch.channels_last) if src.dim() == 4 else src.contiguous()
    C = src.shape[1] if src.dim() == 4 else 0
    if (src.dim() == 4 and C % 4 == 0 and src.dtype in (torch.float32, torch.bfloat16)
            and dst.is_contiguous(memory_format=torch.channels_last)):
        lib.call(_k("nasseg_chan_copy", src), ptr(src), C, 0, ptr(dst), C, 0, None, 0, 0,
                 src.numel() // C, C, ACT_NONE, ACT_NONE, current_stream())
    else:
        dst.copy_(src)  # (3-channel or integer maps: a plain device copy)
    return dst


def gather_rows(src, idx, out=None):
    """src[idx] along dim 0 for a densely stored tensor (NCHW-contiguous or channels_last: one
    sample = one contiguous row) with an int64 DEVICE index: the batch of the task0 feature cache,
    one copy kernel, no ATen indexing and no layout change (src/engine/trainer.py:128-137)."""
    require_device(src, idx)
    if idx.dtype != torch.int64 or idx.dim() != 1 or not idx.is_contiguous():
        raise NassegError("gather_rows: the index must be a contiguous 1-D int64 tensor")
    n = idx.shape[0]
    cl = src.dim() == 4 and src.is_contiguous(memory_format=torch.channels_last)
    if not (cl or src.is_contiguous()):
        raise NassegError("gather_rows: the source must be stored densely")
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), device=src.device, dtype=src.dtype,
                          memory_format=torch.channels_last if cl else torch.contiguous_format)
    row_bytes = (src.numel() // max(src.shape[0], 1)) * src.element_size()
    lib.call("nasseg_gather_rows", ptr(src), ptr(idx), ptr(out), n, row_bytes, src.shape[0], current_stream())
    return out


def argmax_confusion(logits, gt, n_classes, cm=None, out_size=None, return_preds=False):
    """Fused bilinear up-sampling -> argmax -> uint8 -> confusion-matrix update.

    logits (B,C,h,w) fp32 on device, gt (B,H,W) uint8 on device; ``cm`` is an
    int64 (n,n) device tensor that is accumulated into (created when None).
    """
    logits = _cl(logits.detach())
    if logits.dtype != torch.float32:
        logits = logits.float()  # the reward path interpolates and compares in fp32
    B, C, h, w = logits.shape
    preds = None
    if gt is not None:
        require_device(gt)
        if gt.dtype != torch.uint8:
            raise NassegError("gt must be uint8")
        gt = gt.contiguous()
        H, W = gt.shape[1], gt.shape[2]
        if cm is None:
            cm = torch.zeros((n_classes, n_classes), device=logits.device, dtype=torch.int64)
    else:
        H, W = (int(out_size[0]), int(out_size[1])) if out_size is not None else (h, w)
    if return_preds:
        preds = torch.empty((B, H, W), device=logits.device, dtype=torch.uint8)
    lib.call("nasseg_argmax_cm", ptr(logits), ptr(gt), ptr(preds), B, h, w, C, H, W,
             int(n_classes), ptr(cm) if gt is not None else None, current_stream())
    return (cm, preds) if return_preds else cm


def _apply_library_knobs():
    if _PW_MIN_PIXELS is not None:
        lib.query("nasseg_conv_pw_min_pixels", int(_PW_MIN_PIXELS))
        lib._memo.clear()
    if _PWN_MODE is not None:
        lib.query("nasseg_conv_pwn_mode", int(_PWN_MODE))
        lib._memo.clear()
    # NASSEG_DW_WGRAD_LDS=0: the strip kernel for 5x5 depthwise weight gradients too; NASSEG_CONV_DEEP_K=0: one
    # k-step per round trip on small maps as well; NASSEG_POOL_STRIP=0 / 2: stride-1 max pooling one gather per
    # element / two rows per thread; NASSEG_PW_RZ_MIN_PIXELS: maps from which the one-kernel pointwise backward rebuilds
    # z instead of loading it (A/B switches, include/nasseg.h)
    for env, fn in (("NASSEG_DW_WGRAD_LDS", "nasseg_dw_wgrad_lds"), ("NASSEG_CONV_DEEP_K", "nasseg_conv_deep_k"),
                    ("NASSEG_POOL_STRIP", "nasseg_pool_strip"),
                    ("NASSEG_PW_RZ_MIN_PIXELS", "nasseg_conv_pw_bwd_rz_min_pixels")):
        if os.environ.get(env) is not None:
            lib.query(fn, int(os.environ[env]))
            lib._memo.clear()


try:
    _apply_library_knobs()
except (OSError, NassegError):  # (no library yet - a CPU-only import before the build; set again after loading)
    pass
