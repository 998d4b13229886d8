"""Encoder+decoder container and the one-process-per-GPU replacement of
``nn.DataParallel`` (src/main_search.py:411-420,507)."""
import torch
import torch.distributed as dist
import torch.nn as nn


class Segmenter(nn.Module):
    """decoder(encoder(x)) - src/main_search.py:411-420."""

    def __init__(self, encoder, decoder):
        super(Segmenter, self).__init__()
        self.encoder = encoder
        self.decoder = decoder

    def forward(self, x):
        return self.decoder(self.encoder(x))


class RankParallel(nn.Module):
    """Data parallelism with one process per GPU over RCCL.

    Keeps the ``.module`` attribute engine code reaches through
    (``segmenter.module.encoder`` / ``.decoder``).  Each rank runs the full
    replica on its own shard of the batch; after backward, ``sync_gradients()``
    packs the gradients into ONE flat fp32 bucket (one multi-tensor copy), all-reduces it
    (no per-parameter collectives), divides by the world size and leaves every
    ``param.grad`` pointing into the bucket.
    BatchNorm statistics stay per rank, as under nn.DataParallel; parameters and
    buffers are broadcast from rank 0 when a candidate is (re)built.
    Works un-initialised too (world size 1): every collective becomes a no-op.
    """

    def __init__(self, module, process_group=None, broadcast=True):
        super(RankParallel, self).__init__()
        self.module = module
        self.process_group = process_group
        self._flat = None
        self._views = None
        self._used = None
        self._plist = None
        if broadcast:
            self.broadcast_parameters()

    @property
    def world_size(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group)
        return 1

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def broadcast_parameters(self, src=0):
        if self.world_size == 1:
            return
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src, group=self.process_group)

    def _parameters_once(self):
        # a candidate's module tree is fixed: walk it once, not twice per step
        if self._plist is None:
            self._plist = list(self.module.parameters())
        return self._plist

    def _build_bucket(self, used):
        total = sum(p.numel() for p in used)
        ref = used[0]
        self._flat = torch.zeros(total, device=ref.device, dtype=ref.dtype)
        self._used = used
        self._views = []
        off = 0
        for p in used:
            self._views.append(self._flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def attach_flat_grads(self):
        """Call instead of ``optimizer.zero_grad()``: clears every gradient so that autograd
        writes fresh tensors in backward (accumulating into pre-attached views would cost one
        small add kernel per parameter and step).  Which parameters the loss reaches is read off
        those fresh gradients in ``sync_gradients``: parameters autograd never touches keep
        ``grad is None`` - the optimisers skip those, as under the reference's nn.DataParallel."""
        for p in self._parameters_once():
            p.grad = None
        return None

    def sync_gradients(self):
        """Average the gradients over the ranks with ONE collective: the fresh gradients are
        packed into a flat fp32 bucket by one multi-tensor copy, the bucket is all-reduced and
        every reached ``param.grad`` is re-pointed at its slice of it."""
        ws = self.world_size
        if ws == 1:
            return
        # the parameters this backward reached: static per architecture and training stage
        # (decoder only on cached features, everything end to end), hence identical on every
        # rank; the bucket is rebuilt when the stage changes
        have = [p for p in self._parameters_once() if p.requires_grad and p.grad is not None]
        if (self._flat is None or len(have) != len(self._used)
                or any(a is not b for a, b in zip(have, self._used))):
            if not have:
                return
            self._build_bucket(have)
        src, dst = [], []
        for p, v in zip(self._used, self._views):
            if p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.process_group)
        self._flat.div_(ws)
        for p, v in zip(self._used, self._views):
            p.grad = v

    def reduce_confusion(self, cm):
        """Sum the int64 confusion matrix over ranks at the end of validation."""
        if self.world_size > 1:
            dist.all_reduce(cm, op=dist.ReduceOp.SUM, group=self.process_group)
        return cm
