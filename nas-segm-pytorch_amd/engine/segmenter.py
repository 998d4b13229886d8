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
    all-reduces ONE flat fp32 bucket that aliases every ``param.grad`` (no
    per-parameter collectives, no copy in or out) and divides by the world size.
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
        """Call instead of ``optimizer.zero_grad()``.  The first call only clears the
        gradients: which parameters a candidate's loss actually reaches is discovered from
        that first backward (``sync_gradients``), because parameters autograd never touches
        must keep ``grad is None`` - the optimisers skip those, as they do under the
        reference's nn.DataParallel - rather than receive a zero gradient plus weight decay.
        From then on every reached ``param.grad`` is a view into one zeroed flat bucket and
        autograd accumulates in place."""
        if self._flat is None:
            for p in self.module.parameters():
                p.grad = None
            return None
        self._flat.zero_()
        for p, v in zip(self._used, self._views):
            p.grad = v
        return self._flat

    def sync_gradients(self):
        """Sum gradients across ranks and average (one collective)."""
        ws = self.world_size
        if ws == 1:
            return
        # the parameters this backward reached: static per architecture and training stage
        # (decoder only on cached features, everything end to end), hence identical on every
        # rank; the bucket is rebuilt when the stage changes
        have = [p for p in self.module.parameters() if p.requires_grad and p.grad is not None]
        if (self._flat is None or len(have) != len(self._used)
                or any(a is not b for a, b in zip(have, self._used))):
            if not have:
                return
            self._build_bucket(have)
        aliased = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                      for p, v in zip(self._used, self._views))
        if not aliased:
            # first step, or grads were re-allocated (zero_grad(set_to_none)): pack them
            for p, v in zip(self._used, self._views):
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.process_group)
        self._flat.div_(ws)
        if not aliased:
            for p, v in zip(self._used, self._views):
                p.grad = v

    def reduce_confusion(self, cm):
        """Sum the int64 confusion matrix over ranks at the end of validation."""
        if self.world_size > 1:
            dist.all_reduce(cm, op=dist.ReduceOp.SUM, group=self.process_group)
        return cm
