"""Training step replayed from a hipGraph.

A sampled candidate issues 700-1500 small launches per step (many of them at
11x11 ... 81x81 in the CVPR cells); issued one by one from Python the host, not
the GPU, sets the step time at 321x321.  The whole forward + loss + backward of
``segmenter_step`` (engine/trainer.py; reference src/engine/trainer.py:229-257)
touches nothing on the host - every nasseg entry point takes device pointers and
the current stream, allocates nothing and keeps BatchNorm's ``num_batches_tracked``
on the device - so it is captured ONCE per candidate into a hipGraph
(``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replayed with one launch per
step.  What stays outside the graph is what genuinely needs the host or the
network: copying the next batch into the static input buffers, the RCCL gradient
all-reduce, and (unless ``capture_optimisers``) clipping + optimiser steps.

Gradients are cleared (``grad = None``) before the capture, so backward writes them as
fresh tensors of the graph's private pool: every replay refills the same addresses and no
accumulation kernels are recorded.  When data parallel they are packed into one flat fp32
bucket (one multi-tensor copy) and all-reduced with one collective after the replay.
"""
import torch
import torch.distributed as dist
from torch import nn

from .. import functional as F
from .trainer import _clip_and_step, _inner


def _capturable(optim):
    """An optimiser step may be baked into a graph only if none of its scalars live on
    the host: plain SGD qualifies, Adam only with ``capturable=True`` (its bias
    correction otherwise uses a host-side step count that a replay would freeze)."""
    if optim is None:
        return True
    if isinstance(optim, torch.optim.SGD):
        return True
    return all(bool(g.get("capturable", False)) for g in optim.param_groups)


class GraphedSegmenterStep(object):
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
        self.segmenter = segmenter
        self.model = _inner(segmenter)
        self.optim_enc, self.optim_dec = optim_enc, optim_dec
        self.ignore_index = ignore_index
        self.aux_weight = aux_weight
        self.loss_fn = loss_fn
        self.world = int(getattr(segmenter, "world_size", 1))
        self.groups = [
            (list(self.model.encoder.parameters()), enc_grad_clip, optim_enc),
            (list(self.model.decoder.parameters()), dec_grad_clip, optim_dec),
        ]
        self.capture_optimisers = bool(capture_optimisers and self.world == 1
                                       and _capturable(optim_enc) and _capturable(optim_dec))
        self.image = image.detach().clone(memory_format=torch.channels_last)
        self.target = target.detach().clone()
        self.flat = self._views = self._used = None
        self._params = list(self.model.parameters())
        self._pack_memo = F.PackMemo()  # (owns the packed-weight buffers the graph reads)
        # A capture that fails on ONE rank (HIP out of memory, say) must fail on all of them: the
        # others would otherwise wait for it in the first gradient all-reduce.
        error = None
        try:
            self._capture(warmup)
        except RuntimeError as e:
            error = e
        if self.world > 1:
            flag = torch.tensor([1.0 if error is not None else 0.0], device=self.image.device)
            dist.all_reduce(flag, group=getattr(segmenter, "process_group", None))
            if error is None and float(flag) > 0:
                error = RuntimeError("GraphedSegmenterStep: the capture failed on a peer rank")
        if error is not None:
            raise error

    # -- the captured region ---------------------------------------------------------
    def _fwd_bwd(self, with_optimisers):
        for p in self._params:
            p.grad = None
        with F.packed_once(self._pack_memo):  # (one re-pack launch for all chains, recorded too)
            output = self.segmenter(self.image)
            aux_outs = []
            if isinstance(output, tuple):
                output, aux_outs = output
            if self.loss_fn is not None:
                loss = self.loss_fn(output, self.target)
            else:
                target = F.nearest_label_resize(self.target, output.size()[2:])
                loss = F.log_softmax_nll(output, target, self.ignore_index)
                if self.aux_weight > 0:
                    for aux_out in aux_outs:
                        aux_out = F.bilinear_resize(aux_out, target.size()[1:])
                        loss = loss + F.log_softmax_nll(aux_out, target, self.ignore_index) * self.aux_weight
            with F.deferred_wgrad(params=self._params):  # (gradients were cleared above)
                loss.backward()
        if with_optimisers:
            _clip_and_step(self.groups)
        return loss.detach()

    def _bn_buffers(self):
        return [b for m in self.model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)
                for b in (m.running_mean, m.running_var, m.num_batches_tracked) if b is not None]

    def _capture(self, warmup):
        # The probe and warm-up passes (lazy initialisation must happen outside the
        # capture) leave no trace: running statistics and - when the optimisers are
        # inside the graph - parameters and optimiser state are put back afterwards.
        buffers = self._bn_buffers()
        saved = [b.clone() for b in buffers]
        saved_params = None
        if self.capture_optimisers:
            saved_params = [p.detach().clone() for p in self.model.parameters()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._fwd_bwd(self.capture_optimisers)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            if self.capture_optimisers:
                # optimiser state was created lazily by the warm-up: back to "never stepped"
                for p, s in zip(self.model.parameters(), saved_params):
                    p.copy_(s)
                for optim in (self.optim_enc, self.optim_dec):
                    for st in (optim.state.values() if optim is not None else ()):
                        for v in st.values():
                            if torch.is_tensor(v):
                                v.zero_()
            for b, s in zip(buffers, saved):
                b.copy_(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._fwd_bwd(self.capture_optimisers)
        # capturing executes nothing: state is exactly as restored above.  The gradients the
        # capture left in ``param.grad`` are the static tensors every replay refills.
        self._static_grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]
        if self.world > 1:
            used = [p for p, _ in self._static_grads]
            self.flat = torch.zeros(sum(p.numel() for p in used), device=used[0].device,
                                    dtype=used[0].dtype)
            self._views, off = [], 0
            for p in used:
                self._views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def _all_reduce(self):
        torch._foreach_copy_(self._views, [g for _, g in self._static_grads])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM,
                        group=getattr(self.segmenter, "process_group", None))
        self.flat.div_(self.world)
        for (p, _), v in zip(self._static_grads, self._views):
            p.grad = v

    # -- per step ----------------------------------------------------------------------
    def step(self, image=None, target=None):
        if image is not None and image.data_ptr() != self.image.data_ptr():
            self.image.copy_(image, non_blocking=True)
        if target is not None and target.data_ptr() != self.target.data_ptr():
            self.target.copy_(target, non_blocking=True)
        self.graph.replay()
        for p, g in self._static_grads:  # (an eager step in between may have re-pointed them)
            p.grad = g
        if not self.capture_optimisers:
            if self.world > 1:
                self._all_reduce()
            _clip_and_step(self.groups)
        return self.loss

    __call__ = step
