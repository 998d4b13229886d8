"""Training steps replayed from a hipGraph.

A sampled candidate issues 700-1500 small launches per step (many of them at
11x11 ... 81x81 in the CVPR cells); issued one by one from Python the host, not
the GPU, sets the step time up to 713x713.  The whole forward + loss + backward of
``segmenter_step`` / of a decoder-only task0 step (engine/trainer.py; reference
src/engine/trainer.py:128-160,229-257) touches nothing on the host - every nasseg
entry point takes device pointers and the current stream, allocates nothing and keeps
BatchNorm's ``num_batches_tracked`` on the device - so it is captured ONCE per
candidate into a hipGraph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replayed
with one launch per step.  What stays outside the graph is what genuinely needs the
host or the network: copying the next batch (or the next batch's cache indices) into
the static input buffers, the RCCL gradient all-reduce, and (unless
``capture_optimisers``) clipping + optimiser steps.

Gradients are cleared (``grad = None``) before the capture, so backward writes them as
fresh tensors of the graph's private pool: every replay refills the same addresses and no
accumulation kernels are recorded.  When data parallel they go to ``RankParallel.sync_gradients()``
after the replay - the same flat fp32 bucket, status element and step count as a host-launched step,
so replaying and host-launching ranks (and a failing one) pair their collectives.

``train_segmenter`` / ``train_task0`` use these steppers by themselves where they win
(``auto_graph``): at most ``AUTO_GRAPH_MAX_PIXELS`` image pixels per step and rank (one process,
independent candidates per rank, or data-parallel replicas whose gradient all-reduce follows the
replay) - above that the step is GPU-bound and a
replay runs at the speed of host launches (measured, round 3: 4x1024x2048 262.1 eager / 261.0 replayed
images/s, 8x713x713 518-586 / 732, 8x480x640 438 / 675 (bf16), 16x321x321 711 / 1116; with bf16 storage the
4x1024x2048 step is at the edge - 13.3 ms of host work against 13.9 ms of GPU time: 265-287 eager, 287.7 replayed).
"""
import gc
import logging
import os
import weakref

import torch
from torch import nn

from .. import functional as F
from . import graph_dag
from .trainer_common import clip_and_step, inner

logger = logging.getLogger(__name__)

# NASSEG_GRAPH: "auto" (default) | "0" (always launch from the host) | "1" (always replay)
# (round 6: 32 M pixels.  Until round 5 a replay of the 4x1024x2048 step ran level with host launches - the step is
#  GPU-bound - and the limit stood at 6 M; laid out in lanes (engine/graph_dag.py) the replay is 8 % ahead: 279.5 ->
#  302.4 images/s on the headline step, 204.5 -> 222.3 on WACV arch1, profiles/r06_graph_modes.txt.  The price is the
#  graph's memory pool, which holds the SUM of the step's temporaries: 14.2 GiB instead of 6.2 at that size.)
AUTO_GRAPH_MAX_PIXELS = 32 << 20


def auto_graph(segmenter, n_pixels):
    """Should the engine replay this candidate's steps from a hipGraph?  n_pixels = B*H*W of the
    images one step consumes (per rank).  Data parallel the rule is the same - what is replayed
    is forward + loss + backward; packing the bucket, the RCCL all-reduce, clipping and the
    optimisers stay outside the graph - unless NASSEG_GRAPH_DP=0 (host launches on every rank)."""
    mode = os.environ.get("NASSEG_GRAPH", "auto")
    if mode == "0":
        return False
    if mode == "1":
        return True
    if getattr(segmenter, "world_size", 1) > 1 and os.environ.get("NASSEG_GRAPH_DP", "1") == "0":
        return False
    return n_pixels <= AUTO_GRAPH_MAX_PIXELS


class StaleCapture(F.NassegError):
    """a recorded step no longer matches its optimisers (``_GraphedStep.stale``): record a new one"""


def _capturable(optim):
    """An optimiser step may be baked into a graph only if none of its scalars live on
    the host: plain SGD qualifies, Adam only with ``capturable=True`` (its bias
    correction otherwise uses a host-side step count that a replay would freeze)."""
    if optim is None:
        return True
    if isinstance(optim, torch.optim.SGD):
        return True
    return all(bool(g.get("capturable", False)) for g in optim.param_groups)


class _GraphedStep(object):
    """Capture / replay machinery shared by the two steppers.  A subclass provides
    ``_forward_loss()`` (device tensors of ``self`` in, device scalar out), the modules whose
    parameters train (``self._trained``) and the clip / optimiser groups (``self.groups``)."""

    plan = None    # engine/graph_dag.Plan when the recorded step replays as stages of lanes (else: the line, self.graph)
    layout = None  # its summary

    def _init_common(self, segmenter, capture_optimisers, optimisers, warmup):
        self.segmenter = segmenter
        self.model = inner(segmenter)
        self.world = int(getattr(segmenter, "world_size", 1))
        # optimisers inside the graph: plain SGD / Adam through nasseg_optim_step (step counters on the device;
        # an object of this stepper's own - its tables are re-uploaded by the graph at every replay), others only
        # when torch can record them (``_capturable``)
        self._native = None
        if capture_optimisers and self.world == 1:
            from .optim_native import NativeStep

            self._native = NativeStep.build(self.groups)
        self.capture_optimisers = bool(capture_optimisers and self.world == 1
                                       and (self._native is not None or all(_capturable(o) for o in optimisers)))
        self._optimisers = [o for o in optimisers if o is not None]
        self._params = [p for m in self._trained for p in m.parameters()]
        self._pack_memo = F.PackMemo()  # (owns the packed-weight buffers the graph reads)
        # Data parallel, the replayed region ends where the eager step's does: the gradients are handed to
        # RankParallel.sync_gradients() - ONE bucket layout (every trainable parameter + the status element)
        # for replayed steps, host-launched steps and a failing rank's farewell alike.  A capture therefore
        # needs no agreement between the ranks: one that fails here (HIP out of memory, say) leaves THIS rank
        # launching from the host while its peers replay, and their collectives still pair.
        self._rank_parallel = weakref.ref(segmenter) if self.world > 1 else None
        self._capture(warmup)
        # The modules were needed to RECORD the step; replaying needs parameters, optimisers and
        # static tensors only.  Dropping them here means a stepper cached on its model (engine/
        # trainer.py) forms no reference cycle: the candidate's graph and its memory pool go away
        # with the candidate, by reference counting, not at some later garbage collection.
        self.segmenter = self.model = self._trained = None
        if hasattr(self, "decoder"):
            self.decoder = None

    # -- the captured region ---------------------------------------------------------
    def _fwd_bwd(self, with_optimisers):
        for p in self._params:
            p.grad = None
        with F.packed_once(self._pack_memo):  # (one re-pack launch for all chains, recorded too)
            loss = self._forward_loss()
            with F.deferred_wgrad(params=self._params):  # (gradients were cleared above)
                loss.backward()
        if with_optimisers:
            clip_and_step(self.groups, self._native)
        return loss.detach()

    def _bn_buffers(self):
        return [b for m in self.model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)
                for b in (m.running_mean, m.running_var, m.num_batches_tracked) if b is not None]

    def _capture(self, warmup):
        # The probe and warm-up passes (lazy initialisation must happen outside the
        # capture) leave no trace: running statistics and - when the optimisers are
        # inside the graph - parameters and optimiser state are put back afterwards,
        # ALSO when the warm-up or the capture raises (HIP out of memory, say): the engine then
        # launches this candidate from the host, and its BatchNorm statistics must not have
        # advanced by the warm-up's momentum updates.
        self.layout = self.plan = None
        buffers = self._bn_buffers()
        saved = [b.clone() for b in buffers]
        saved_params = saved_state = None
        if self.capture_optimisers:
            saved_params = [p.detach().clone() for p in self.model.parameters()]
            # optimiser state that exists already (a stepper rebuilt in the middle of a
            # candidate's training) is real momentum: keep a copy, not zeros
            saved_state = [dict((id(p), dict((k, v.clone() if torch.is_tensor(v) else v) for k, v in st.items()))
                                for p, st in optim.state.items()) for optim in self._optimisers]

        def restore():
            with torch.no_grad():
                if self.capture_optimisers:
                    for p, s in zip(self.model.parameters(), saved_params):
                        p.copy_(s)
                    for optim, before in zip(self._optimisers, saved_state):
                        for p, st in optim.state.items():
                            old = before.get(id(p))
                            for k, v in st.items():
                                if not torch.is_tensor(v):
                                    continue
                                if old is not None and torch.is_tensor(old.get(k)):
                                    v.copy_(old[k])
                                else:  # created lazily by the warm-up: back to "never stepped"
                                    v.zero_()
                for b, s in zip(buffers, saved):
                    b.copy_(s)
                if self._native is not None:
                    self._native.sync_steps()  # (device step counters := the restored state["step"])

        done = False
        try:
            # lanes > 1: the recorded line is laid out again with the real dependencies between its launches
            # (engine/graph_dag.py) - the graph object is kept after the capture; the last warm-up pass is timed
            # per call (HIP events), which is what the layout's cost model runs on
            lanes = graph_dag.LANES if hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph") else 1
            timers = []
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(max(1, warmup)):
                    timing = lanes > 1 and F.lib.profiler is None
                    if timing:
                        timers.append(F.LaunchProfiler())
                        F.lib.profiler = timers[-1]
                    try:
                        self._fwd_bwd(self.capture_optimisers)
                    finally:
                        if timing:
                            F.lib.profiler = None
            torch.cuda.current_stream().wait_stream(side)
            measured = None
            if timers:
                torch.cuda.synchronize()
                passes = [[(name, 1e3 * e0.elapsed_time(e1)) for name, _, e0, e1 in t.records] for t in timers]
                measured = passes[-1]
                for other in passes[:-1]:  # (the same calls in the same order: the shorter sample of each)
                    if [n for n, _ in other] == [n for n, _ in measured]:
                        measured = [(n, min(a, b)) for (n, a), (_, b) in zip(measured, other)]
                timers = passes = None
            restore()
            # No garbage collection while the stream is capturing: a collected cycle may hold device
            # tensors or another candidate's hipGraph, whose destruction inside a capture aborts the
            # process (torch >= 2.9 no longer collects before a capture by itself).  Collect now,
            # hold the collector off for the capture.
            self.graph = torch.cuda.CUDAGraph(keep_graph=True) if lanes > 1 else torch.cuda.CUDAGraph()
            self.layout = self.plan = None
            if self._native is not None:
                self._native.prepare_capture()
            gc.collect()
            gc_was_enabled = gc.isenabled()
            gc.disable()
            recorder = n_nodes = None
            try:
                with torch.cuda.graph(self.graph):
                    if lanes > 1:
                        with graph_dag.Recorder() as recorder:
                            self.loss = self._fwd_bwd(self.capture_optimisers)
                            n_nodes = recorder.nodes()
                    else:
                        self.loss = self._fwd_bwd(self.capture_optimisers)
            finally:
                if gc_was_enabled:
                    gc.enable()
            if self._native is not None:
                self._native.finish_capture()  # (its tables: uploaded once, here - the graph holds no copy nodes)
            if recorder is not None:
                recorder.release()
                raw = self.graph.raw_cuda_graph()
                if graph_dag.MODE == "rewire":
                    # (a failure here is a failed capture: the graph may be left without its order)
                    self.layout = graph_dag.lay_out(recorder, raw, n_nodes, lanes=lanes, durations=measured)
                    self.graph.instantiate()
                else:
                    self.graph.instantiate()
                    self._lay_out(recorder, raw, n_nodes, lanes, measured, restore)
            done = True
        finally:
            if not done:
                # whatever the warm-up changed before it (or the capture) failed is undone, and no
                # half-written gradient is left for the eager step that follows
                try:
                    torch.cuda.synchronize()
                    restore()
                finally:
                    for p in self._params:
                        p.grad = None
                    self.graph = None
        # capturing executes nothing: state is exactly as restored above.  The gradients the
        # capture left in ``param.grad`` are the static tensors every replay refills.
        self._static_grads = [(p, p.grad) for p in self._params if p.grad is not None]
        self._captured_hyper = self._native.hyper_values() if self._native is not None else None

    def _lay_out(self, recorder, raw, n_nodes, lanes, measured, restore):
        """the recorded step as stages of independent lanes (engine/graph_dag.py): the candidate layouts and the line
        as recorded are timed on the step itself - what those replays change is put back - and the fastest is kept"""
        import time

        def trial(run, reps=4):
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        try:
            timed = graph_dag.TRIALS
            plan, layout = graph_dag.lay_out_stages(recorder, raw, n_nodes, lanes=lanes, durations=measured,
                                                    trial=trial if timed else None)
            if plan is not None and timed:
                line = trial(self.graph.replay)
                layout["line_ms"] = round(1e3 * line, 3)
                if layout.get("ms") is None or 1e-3 * layout["ms"] >= line:
                    plan.close()
                    plan = None
            self.plan, self.layout = plan, layout
        except F.NassegError as e:  # (the recorded graph is untouched: replay the line)
            logger.warning("graph_dag: %s - replaying the step as recorded", e)
            self.plan, self.layout = None, {"mode": "line", "error": str(e)}
        finally:
            torch.cuda.synchronize()
            restore()

    def _sync_gradients(self):
        """the step's gradient collective - RankParallel's, so that the failure protocol (status element,
        ``sync_count``) covers replayed steps: a peer that left its step loop is seen by ``check_peers`` right
        after this step, and a failure AFTER this point (clipping, the optimisers) is told in the next one"""
        owner = self._rank_parallel() if self._rank_parallel is not None else None
        if owner is None:
            raise F.NassegError("graphed step: the data-parallel segmenter it was captured for is gone")
        owner.sync_gradients()

    def stale(self):
        """True when a replay would step with other hyper-parameters than the optimisers hold NOW: lr, weight decay,
        betas, eps and the clip norms are recorded BY VALUE (kernel arguments of nasseg_optim_step), so a schedule that
        edits param_group["lr"] needs a new capture.  The owner of a cached stepper asks before it replays and builds
        a new one (engine/trainer._cached_stepper); ``_replay`` itself refuses with StaleCapture."""
        return (self.capture_optimisers and self._native is not None
                and self._native.hyper_values() != self._captured_hyper)

    def _replay(self):
        native = self._native if self.capture_optimisers else None
        if native is not None:
            # lr, weight decay, betas, eps and the clip norms were recorded BY VALUE (kernel arguments of
            # nasseg_optim_step): a schedule that edits param_group["lr"] must not be ignored silently
            if self.stale():
                raise StaleCapture("graphed step: an optimiser's hyper-parameters changed after its step was "
                                   "captured (capture_optimisers=True bakes lr / weight decay / betas / eps / "
                                   "max_norm into the graph) - build a new stepper (engine/trainer.py's cache does: it "
                                   "asks stale() first), or keep the optimisers outside the graph "
                                   "(capture_optimisers=False reads param_groups every step)")
            if not native.host_steps_match():
                native.sync_steps()  # (the optimisers were stepped or reloaded outside this graph)
        if self.plan is not None:
            self.plan.run()  # (the step as stages of line graphs, engine/graph_dag.py)
        else:
            self.graph.replay()
        if native is not None:
            native.bump_host_steps()
        for p, g in self._static_grads:  # (an eager step in between may have re-pointed them)
            p.grad = g
        if not self.capture_optimisers:
            if self.world > 1:
                self._sync_gradients()
            clip_and_step(self.groups)
        return self.loss


class GraphedSegmenterStep(_GraphedStep):
    """``segmenter_step`` with forward/loss/backward replayed from a hipGraph.

    step(image, target) -> device loss (a static tensor, valid until the next step).
    Shapes are fixed at construction (a new candidate or a new crop size needs a new
    object - the reference rebuilds the segmenter per candidate anyway).
    """

    def __init__(self, segmenter, image, target, optim_enc, optim_dec, ignore_index=255,
                 enc_grad_clip=0.0, dec_grad_clip=0.0, aux_weight=-1, capture_optimisers=False,
                 warmup=2, loss_fn=None):
        """loss_fn(output, target) -> scalar replaces the softmax/NLL (+ aux heads) of the
        segmentation step, e.g. ``F.berhu_loss`` for a depth head; it must be capturable (device
        tensors in, device scalar out, no host synchronisation)."""
        model = inner(segmenter)
        self.optim_enc, self.optim_dec = optim_enc, optim_dec
        self.ignore_index = ignore_index
        self.aux_weight = aux_weight
        self.loss_fn = loss_fn
        self._trained = [model.encoder, model.decoder]
        self.groups = [
            (list(model.encoder.parameters()), enc_grad_clip, optim_enc),
            (list(model.decoder.parameters()), dec_grad_clip, optim_dec),
        ]
        self.image = image.detach().clone(memory_format=torch.channels_last)
        self.target = target.detach().clone()
        self._init_common(segmenter, capture_optimisers, (optim_enc, optim_dec), warmup)

    def _forward_loss(self):
        output = self.segmenter(self.image)
        aux_outs = []
        if isinstance(output, tuple):
            output, aux_outs = output
        if self.loss_fn is not None:
            return self.loss_fn(output, self.target)
        target = F.nearest_label_resize(self.target, output.size()[2:])
        loss = F.log_softmax_nll(output, target, self.ignore_index)
        if self.aux_weight > 0:
            for aux_out in aux_outs:
                aux_out = F.bilinear_resize(aux_out, target.size()[1:])
                loss = loss + F.log_softmax_nll(aux_out, target, self.ignore_index) * self.aux_weight
        return loss

    def matches(self, image, target):
        return (tuple(image.shape) == tuple(self.image.shape) and image.dtype == self.image.dtype
                and tuple(target.shape) == tuple(self.target.shape) and target.dtype == self.target.dtype)

    def step(self, image=None, target=None):
        if image is not None and image.data_ptr() != self.image.data_ptr():
            self.image.copy_(image, non_blocking=True)
        if target is not None and target.data_ptr() != self.target.data_ptr():
            self.target.copy_(target, non_blocking=True)
        return self._replay()

    __call__ = step


class GraphedTask0Step(_GraphedStep):
    """A decoder-only step on the device-resident feature cache (``train_task0``), replayed from
    a hipGraph: the batch is gathered from the cache by index INSIDE the graph
    (nasseg_gather_rows), so a step costs the host one copy of ``batch_size`` indices and one
    graph launch.

    step(indices) -> device loss; ``indices``: int64 tensor / array of ``batch_size`` cache rows.
    """

    def __init__(self, Xy_train, segmenter, optim_dec, batch_size, ignore_index=255, dec_grad_clip=0.0,
                 aux_weight=0, capture_optimisers=False, warmup=2):
        model = inner(segmenter)
        self.cache = Xy_train
        self.feat_keys = [k for k in Xy_train.keys() if k not in ("y", "kd_y", "out_size")]
        self.out_size = tuple(int(v) for v in Xy_train["out_size"])
        self.optim_dec = optim_dec
        self.ignore_index = ignore_index
        self.aux_weight = aux_weight
        self._trained = [model.decoder]
        self.groups = [(list(model.decoder.parameters()), dec_grad_clip, optim_dec)]
        self.decoder = model.decoder
        self.index = torch.arange(batch_size, device=Xy_train["y"].device, dtype=torch.int64)
        self._cache_rows = int(Xy_train["y"].shape[0])
        self._init_common(segmenter, capture_optimisers, (optim_dec,), warmup)

    def _forward_loss(self):
        feats = [F.gather_rows(self.cache[k], self.index) for k in self.feat_keys]
        target = F.gather_rows(self.cache["y"], self.index)
        output = self.decoder(feats)
        aux_outs = []
        if isinstance(output, tuple):
            output, aux_outs = output
        output = F.bilinear_resize(output, self.out_size)
        loss = F.log_softmax_nll(output, target, self.ignore_index)
        if self.aux_weight > 0:
            for aux_out in aux_outs:
                aux_out = F.bilinear_resize(aux_out, self.out_size)
                loss = loss + F.log_softmax_nll(aux_out, target, self.ignore_index) * self.aux_weight
        return loss

    def step(self, indices):
        idx = torch.as_tensor(indices, dtype=torch.int64)
        if tuple(idx.shape) != tuple(self.index.shape):
            raise F.NassegError("GraphedTask0Step: batches of {} indices (got {})".format(
                self.index.numel(), tuple(idx.shape)))
        if not idx.is_cuda and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self._cache_rows):
            # the reference's Xy_train[k][train_idx] raises here (trainer.py:132-137); the kernel's
            # clamp is a memory-safety net only
            raise IndexError("GraphedTask0Step: cache row index out of range [0, {})".format(self._cache_rows))
        self.index.copy_(idx, non_blocking=True)
        return self._replay()

    __call__ = step
