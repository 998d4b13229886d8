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


class PeerFailure(RuntimeError):
    """A data-parallel peer reported a failure in the step just synchronised (see
    RankParallel.sync_gradients): every rank raises this at the same point of its own step
    sequence, so that the engine's ``try_except`` convention - a RuntimeError scores the candidate
    0 - abandons the candidate on ALL ranks together and the collectives stay paired."""


class RankFailure(RuntimeError):
    """A data-parallel rank's step loop ended with an exception that is not a RuntimeError (an
    IndexError from the feature cache, a ValueError from a loader ...).  In ONE process the reference lets
    those escape ``try_except`` (helpers/utils.py:172-187); with one process per GPU that would end this
    rank's process while its peers wait in the next collective, so the engine re-raises them as this
    RuntimeError (``__cause__`` = the original): every rank then scores the candidate 0 and moves on."""


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

    Failure protocol.  The reference runs in ONE process, where "RuntimeError => candidate
    scored 0" (helpers/utils.py:172-187) needs no coordination.  With one process per GPU a
    rank that leaves its step loop alone would leave its peers blocked in the next all-reduce.
    The bucket therefore has a fixed layout (every trainable parameter, whether or not the
    loss of the current stage reaches it - so a rank can take part in the collective without
    having run backward) plus ONE status element: a rank whose forward/backward raised calls
    ``sync_gradients(failed=True)`` - zeros and status 1 - and re-raises; its peers find a
    non-zero status next to their loss value (``check_peers``, no extra collective and no
    extra host synchronisation) and raise ``PeerFailure``.  The same slot rides on the
    confusion-matrix all-reduce of validation.
    """

    def __init__(self, module, process_group=None, broadcast=True, independent=False):
        """independent: this rank trains a candidate of its OWN (BASELINE config 4: one sampled
        architecture per GPU) - no parameter broadcast, no gradient or confusion-matrix
        all-reduce, whatever the process group's size."""
        super(RankParallel, self).__init__()
        self.module = module
        self.process_group = process_group
        self.independent = bool(independent)
        self._flat = None
        self._views = None
        self._plist = None
        self._status = None
        self.sync_count = 0  # gradient collectives this rank has taken part in
        if broadcast and not independent:
            self.broadcast_parameters()

    @property
    def world_size(self):
        if self.independent:
            return 1
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
        # a candidate's module tree is fixed: walk it once, not twice per step (the version
        # counter is bumped by the one API that swaps modules, TemplateDecoder._reset_clf)
        from ..nn.modules import TREE_VERSION

        if self._plist is None or self._plist[0] != TREE_VERSION[0]:
            self._plist = (TREE_VERSION[0], [p for p in self.module.parameters() if p.requires_grad])
            self._flat = None
        return self._plist[1]

    def _build_bucket(self):
        params = self._parameters_once()
        total = sum(p.numel() for p in params)
        ref = params[0]
        self._flat = torch.zeros(total + 1, device=ref.device, dtype=torch.float32)
        self._views = []
        off = 0
        for p in params:
            self._views.append(self._flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._status = self._flat[total:total + 1]

    def attach_flat_grads(self):
        """Call instead of ``optimizer.zero_grad()``: clears every gradient so that autograd
        writes fresh tensors in backward (accumulating into pre-attached views would cost one
        small add kernel per parameter and step).  Which parameters the loss reaches is read off
        those fresh gradients in ``sync_gradients``: parameters autograd never touches keep
        ``grad is None`` - the optimisers skip those, as under the reference's nn.DataParallel."""
        for p in self._parameters_once():
            p.grad = None
        return None

    def sync_gradients(self, failed=False):
        """Average the gradients over the ranks with ONE collective: the fresh gradients are
        packed into the flat fp32 bucket by one multi-tensor copy, the bucket is all-reduced and
        every reached ``param.grad`` is re-pointed at its slice of it.  ``failed``: this rank's
        forward/backward raised - it contributes zeros and sets the status element, so that the
        peers' collective completes and they learn about it (``check_peers``)."""
        ws = self.world_size
        if ws == 1:
            return
        params = self._parameters_once()
        if self._flat is None:
            self._build_bucket()
        self.sync_count += 1
        if failed:
            self._flat.zero_()
            self._status.fill_(1.0)
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.process_group)
            return
        src, dst, reached = [], [], []
        for p, v in zip(params, self._views):
            if p.grad is None:
                continue  # (its slice is never written: stays zero on every rank)
            reached.append((p, v))
            if p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        self._status.zero_()
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.process_group)
        self._flat.div_(ws)
        for p, v in reached:
            p.grad = v

    def step_values(self, loss):
        """(loss value, peers ok) with ONE device-to-host copy - the host synchronisation the
        reference's ``loss.item()`` makes every step anyway."""
        if self.world_size == 1 or self._status is None:
            return float(loss), True
        both = torch.stack([loss.detach().reshape(()).float(), self._status[0]]).tolist()
        return both[0], both[1] == 0.0

    def check_peers(self, loss):
        """float(loss); raises PeerFailure (on every healthy rank alike) when a peer failed in
        the step whose gradients were just synchronised"""
        value, ok = self.step_values(loss)
        if not ok:
            raise PeerFailure("a data-parallel peer failed in this step: candidate abandoned on all ranks")
        return value

    def epoch_status(self, failed=False):
        """The end-of-epoch handshake of a training epoch: one single-element all-reduce that every rank
        which reached the end of its step loop takes part in - and a rank whose LAST step failed after that
        step's gradient collective (clipping, the optimisers), with ``failed``: there is no next step whose
        collective could carry its status.  Raises PeerFailure on the healthy ranks."""
        if self.world_size == 1:
            return
        if self._flat is None:
            self._build_bucket()
        flag = torch.full((1,), 1.0 if failed else 0.0, device=self._flat.device, dtype=torch.float32)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.process_group)
        if not failed and float(flag) != 0.0:
            raise PeerFailure("a data-parallel peer failed in the epoch's last step: candidate abandoned on all ranks")

    def reduce_confusion(self, cm, failed=False):
        """Sum the int64 confusion matrix over ranks at the end of validation; one extra element
        carries the failure status (see the class docstring)."""
        if self.world_size > 1:
            buf = torch.zeros(cm.numel() + 1, device=cm.device, dtype=torch.int64)
            if failed:
                buf[-1] = 1
            else:
                buf[:-1].copy_(cm.reshape(-1))
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.process_group)
            if failed:
                return cm
            if int(buf[-1]) != 0:
                raise PeerFailure("a data-parallel peer failed during validation: candidate abandoned")
            cm.copy_(buf[:-1].view_as(cm))
        return cm
