"""Candidate training: end-to-end step (task1), decoder-only step on cached
encoder features (task0) and the feature cache itself.

Step semantics follow src/engine/trainer.py (populate_task0 :17-74, train_task0
:78-175, train_segmenter :179-283): LogSoftmax(dim 1) + NLL(ignore 255, mean over
valid pixels), auxiliary heads weighted by ``aux_weight``, per-sub-module
gradient-norm clipping, separate encoder / decoder optimisers, optional Polyak
averaging, one host sync per step for the loss value.  Differences are confined
to where the work runs: forward, backward and loss are nasseg HIP kernels;
gradients of data-parallel replicas are all-reduced over RCCL before clipping.
"""
import logging
import time
from collections import defaultdict

import numpy as np
import torch
from torch import nn

from .. import functional as F
from ..helpers.utils import AverageMeter, try_except
from ..nn.modules import TREE_VERSION
from .trainer_common import clip_and_step as _clip_and_step
from .trainer_common import inner as _inner

logger = logging.getLogger(__name__)


def _set_stage(loader, stage):
    ds = getattr(loader, "dataset", None)
    if ds is None:
        return
    try:
        ds.set_stage(stage)
    except AttributeError:
        sub = getattr(ds, "dataset", None)
        if sub is not None and hasattr(sub, "set_stage"):
            sub.set_stage(stage)


def _freeze_bn(module):
    for m in module.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()


def _ignore_index(segm_crit):
    return int(getattr(segm_crit, "ignore_index", 255))


def _to_device_image(image, device):
    # fp32 like the reference; a loader that already yields bfloat16 images selects the bf16
    # activation-storage twins of every kernel (include/nasseg.h)
    dtype = torch.bfloat16 if image.dtype == torch.bfloat16 else torch.float32
    return image.to(device=device, dtype=dtype, non_blocking=True).contiguous(
        memory_format=torch.channels_last)


def _model_device(module):
    return next(module.parameters()).device


def _labels(mask, device):
    if mask.dtype not in (torch.uint8, torch.int64):
        mask = mask.to(torch.int64)
    return mask.to(device, non_blocking=True)


def _polyak_update(params, avg_param, decay):
    with torch.no_grad():
        for p, avg_p in zip(params, avg_param):
            avg_p.mul_(decay).add_(p.data, alpha=1.0 - decay)


def _zero_grads(segmenter, optimisers):
    """every ``param.grad`` back to None (not zeros): deferred_wgrad relies on autograd ADOPTING
    the fresh gradient tensors of backward"""
    if hasattr(segmenter, "attach_flat_grads") and getattr(segmenter, "world_size", 1) > 1:
        segmenter.attach_flat_grads()
    else:
        for o in optimisers:
            if o is not None:
                o.zero_grad(set_to_none=True)


def _distributed(segmenter):
    return getattr(segmenter, "world_size", 1) > 1


def _loss_value(segmenter, loss):
    """the per-step host synchronisation (``loss.item()`` in the reference); data parallel, the
    same copy brings the peers' status: PeerFailure if one of them failed in this step"""
    if _distributed(segmenter):
        return segmenter.check_peers(loss)
    return loss.item()


def _syncs(segmenter):
    return getattr(segmenter, "sync_count", 0)


def _tell_peers(segmenter, exc, syncs_before=None, last_step=False):
    """Data parallel, this rank is leaving its step loop with ``exc``: take part - once per
    failure - in the gradient collective its peers are (or will be) waiting in, with zeros and
    the failure status, so that they abandon the candidate too (RankParallel's failure
    protocol).  ``syncs_before``: segmenter.sync_count at the start of the step - if this rank
    has already joined the step's collective (the failure came later: clipping, the optimiser)
    the flag reaches the peers in the NEXT step's collective and they all stop there; after the
    epoch's last step there is no such collective and the flag goes into the end-of-epoch handshake the
    peers are about to make (RankParallel.epoch_status).  A PeerFailure needs no telling: every healthy
    rank raised it at the same point."""
    from .segmenter import PeerFailure

    if not _distributed(segmenter) or isinstance(exc, PeerFailure) or getattr(exc, "_nasseg_peers_told", False):
        return
    try:
        exc._nasseg_peers_told = True
    except AttributeError:
        pass
    if last_step and syncs_before is not None and _syncs(segmenter) > syncs_before:
        segmenter.epoch_status(failed=True)  # (the peers are on their way to the end-of-epoch handshake)
        return
    segmenter.sync_gradients(failed=True)


def _as_rank_failure(segmenter, exc):
    """what the step loop re-raises: data parallel, an exception that ``try_except`` would let through (not
    a RuntimeError) becomes a RankFailure, so that this rank scores the candidate 0 like the peers it just
    told - instead of ending its process while they go on to the next collective"""
    from .segmenter import RankFailure

    if not _distributed(segmenter) or isinstance(exc, RuntimeError):
        return exc
    wrapped = RankFailure("{}: {}".format(type(exc).__name__, exc))
    wrapped.__cause__ = exc
    return wrapped


def _epoch_handshake(segmenter):
    """after the last step of a training epoch (data parallel only): see RankParallel.epoch_status"""
    if _distributed(segmenter):
        segmenter.epoch_status()


def _agreed_count(segmenter, n):
    """the smallest ``n`` over the data-parallel ranks (loaders / cache shards of unequal length
    would otherwise issue different numbers of gradient all-reduces); ``n`` itself otherwise"""
    if not _distributed(segmenter) or n is None:
        return n
    import torch.distributed as dist

    t = torch.tensor([int(n)], device=_model_device(_inner(segmenter)), dtype=torch.int64)
    if dist.get_backend(getattr(segmenter, "process_group", None)) == "gloo":
        t = t.cpu()
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=getattr(segmenter, "process_group", None))
    return int(t.item())


def _graphed():
    from . import graphed  # (graphed imports this module's helpers)

    return graphed


def _replays(segmenter, device, n_pixels):
    """does this step run as a hipGraph replay?  (device memory only; engine/graphed.py: auto_graph)"""
    return device.type == "cuda" and _graphed().auto_graph(segmenter, n_pixels)


def _bn_modes(module):
    return tuple(m.training for m in module.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))


_STEPPERS_PER_CANDIDATE = 2  # captured shapes kept per candidate (the usual batch + a last, smaller one)


def _trainable_signature(params, optimisers):
    """what a captured graph has baked in besides shapes: which parameters receive gradients and
    which tensors the optimisers step on (frozen / unfrozen parameters or edited param_groups
    between epochs need a new capture)"""
    return (tuple(p.requires_grad for p in params),
            tuple(tuple(id(q) for g in o.param_groups for q in g["params"]) for o in optimisers if o is not None))


def _cached_stepper(owner, slot, base_key, shape_key, build):
    """The hipGraph steppers of one candidate: ``base_key`` = everything a capture depends on
    except the batch's shape (optimisers, module tree, BatchNorm modes, trainable set ...) - when
    it changes all steppers are dropped; ``shape_key`` selects among at most
    _STEPPERS_PER_CANDIDATE captured shapes.  A shape beyond that (or one whose capture failed
    once) returns None: that batch is launched from the host instead of paying two warm-up
    passes and a capture for a batch that comes once per epoch."""
    ent = getattr(owner, slot, None)
    if ent is None or ent[0] != base_key:
        ent = (base_key, {})
        setattr(owner, slot, ent)
    steppers = ent[1]
    if shape_key in steppers:
        stepper = steppers[shape_key]
        if stepper is None or not stepper.stale():
            return stepper
        # (optimisers inside the graph and a schedule moved lr / weight decay / a clip norm: those are recorded by
        #  value - record the step again instead of failing in the middle of an epoch)
        del steppers[shape_key]
    if len(steppers) >= _STEPPERS_PER_CANDIDATE:
        return None
    try:
        stepper = build()
    except RuntimeError as e:  # e.g. HIP out of memory while capturing: launch from the host
        logger.warning(" hipGraph capture failed (%s): launching from the host", e)
        stepper = None
    steppers[shape_key] = stepper
    return stepper


def _task0_stepper(Xy_train, segmenter, optim_dec, batch_size, ignore, dec_grad_clip, aux_weight, freeze_bn):
    model = _inner(segmenter)
    base = (TREE_VERSION[0], id(optim_dec), ignore, dec_grad_clip, aux_weight, _bn_modes(model.decoder),
            _trainable_signature(list(model.decoder.parameters()), (optim_dec,)))
    shape = (batch_size, tuple((k, v.data_ptr(), tuple(v.shape)) for k, v in Xy_train.items() if k != "out_size"))
    return _cached_stepper(model, "_nasseg_task0_stepper", base, shape, lambda: _graphed().GraphedTask0Step(
        Xy_train, segmenter, optim_dec, batch_size, ignore, dec_grad_clip, aux_weight))


def _segmenter_stepper(segmenter, image, target, optim_enc, optim_dec, ignore, enc_grad_clip, dec_grad_clip,
                       aux_weight):
    model = _inner(segmenter)
    base = (TREE_VERSION[0], id(optim_enc), id(optim_dec), ignore, enc_grad_clip, dec_grad_clip, aux_weight,
            _bn_modes(model), _trainable_signature(list(model.parameters()), (optim_enc, optim_dec)))
    shape = (tuple(image.shape), image.dtype, tuple(target.shape), target.dtype)
    return _cached_stepper(model, "_nasseg_task1_stepper", base, shape, lambda: _graphed().GraphedSegmenterStep(
        segmenter, image, target, optim_enc, optim_dec, ignore, enc_grad_clip, dec_grad_clip, aux_weight))


@try_except
def populate_task0(segmenter, train_loader, kd_net, n_train, do_kd=False):
    """Run the encoder (eval, no grad, one image at a time) over ``n_train``
    samples and keep its feature maps, the nearest-resized labels and optionally
    the teacher's logits on the device.  Returns the cache dict
    {0..S-1: (N,C,h,w), 'y': (N,h,w) int64, ['kd_y'], 'out_size': (h,w)}.

    The cache is DEVICE-RESIDENT and NHWC: every entry keeps the reference's (N, C, h, w) shape
    and is stored channels_last, pre-allocated for ``n_train`` samples on the first batch and
    filled in place by copy kernels (no list of slices, no torch.stack, no layout change), so
    that a training step gathers its batch with one nasseg_gather_rows per entry.  Data
    parallel, every rank caches the samples of ITS loader: the cache is sharded."""
    cache = {}
    segmenter.eval()
    _set_stage(train_loader, "train")
    if hasattr(train_loader, "batch_sampler") and train_loader.batch_sampler is not None:
        train_loader.batch_sampler.batch_size = 1
    model = _inner(segmenter)
    device = _model_device(model)

    def slot(key, like, seen, b):
        """rows [seen, seen+b) of cache[key] (allocated on first use for n_train + b - 1 rows:
        the loop below stops at the first batch that reaches n_train)"""
        if key not in cache:
            shape = (n_train + b - 1,) + tuple(like.shape[1:])
            cache[key] = (torch.empty(shape, device=device, dtype=like.dtype, memory_format=torch.channels_last)
                          if like.dim() == 4 else torch.empty(shape, device=device, dtype=like.dtype))
        return cache[key][seen:seen + b]

    def store(key, t, seen):
        F.copy_into(slot(key, t, seen, t.shape[0]), t)

    with torch.no_grad():
        seen = 0
        for sample in train_loader:
            image = _to_device_image(sample["image"], device)
            b = image.size(0)
            feats = model.encoder(image)
            for i, f in enumerate(feats):
                store(i, f, seen)
            size = feats[0].size()[2:]
            labels = _labels(sample["mask"], device)
            F.nearest_label_resize(labels, size, out=slot(
                "y", torch.empty((1,) + tuple(size), dtype=torch.int64), seen, b))
            if do_kd:
                # (the teacher runs in fp32 whatever the candidate's activation storage)
                store("kd_y", F.bilinear_resize(kd_net(image if image.dtype == torch.float32 else image.float()),
                                                size), seen)
            seen += b
            if seen >= n_train:
                logger.info(" Populated Xy_train, N = {}".format(seen))
                for k in list(cache):
                    cache[k] = cache[k][:seen]
                cache["out_size"] = size
                break
        else:
            for k in list(cache):  # (loader exhausted first - as in the reference, no 'out_size' then)
                cache[k] = cache[k][:seen]
    return cache


def make_task0_step(Xy_train, segmenter, optim_dec, batch_size, ignore_index=255, dec_grad_clip=0.0, aux_weight=0,
                    freeze_bn=False, do_kd=False, kd_coeff=0.0, kd_crit=None):
    """step(batch_idx) -> device loss: one decoder-only training step on the cache rows
    ``batch_idx`` (a host array of ``batch_size`` indices).  Small batches are launch-bound, so the
    step is replayed from a hipGraph where that wins (engine/graphed.py: auto_graph; the stepper
    lives with the model for as long as cache, decoder, BatchNorm modes and optimiser stay the
    same); otherwise - and always with a distillation term or data parallel - it is launched from
    the host: gather the batch (one kernel per cache entry), decoder forward, bilinear resize to
    ``out_size``, softmax/NLL (+ aux heads), backward, [all-reduce], clip, optimiser."""
    decoder = _inner(segmenter).decoder
    feat_keys = [k for k in Xy_train.keys() if k not in ("y", "kd_y", "out_size")]
    out_size = tuple(Xy_train["out_size"])
    device = Xy_train["y"].device
    dec_params = list(decoder.parameters())
    pack_memo = F.PackMemo()
    n_rows = int(Xy_train["y"].shape[0])
    n_pixels = batch_size * int(Xy_train[feat_keys[0]].shape[2]) * int(Xy_train[feat_keys[0]].shape[3]) * 16
    if not do_kd and _replays(segmenter, device, n_pixels):
        stepper = _task0_stepper(Xy_train, segmenter, optim_dec, batch_size, ignore_index, dec_grad_clip,
                                 aux_weight, freeze_bn)
        if stepper is not None:
            return stepper.step

    def step(batch_idx):
        syncs = _syncs(segmenter)
        try:
            idx = torch.as_tensor(batch_idx, dtype=torch.int64)
            if not idx.is_cuda and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= n_rows):
                raise IndexError("train_task0: cache row index out of range [0, {})".format(n_rows))  # as Xy[k][idx]
            idx = idx.to(device, non_blocking=True)
            with F.packed_once(pack_memo):  # (one weight re-pack launch per step)
                feats = [F.gather_rows(Xy_train[k], idx) for k in feat_keys]
                target = F.gather_rows(Xy_train["y"], idx)
                output = decoder(feats)
                aux_outs = []
                if isinstance(output, tuple):
                    output, aux_outs = output
                output = F.bilinear_resize(output, out_size)
                loss = F.log_softmax_nll(output, target, ignore_index)
                if do_kd:
                    loss = loss + kd_coeff * kd_crit(output, F.gather_rows(Xy_train["kd_y"], idx))
                if aux_weight > 0:
                    for aux_out in aux_outs:
                        aux_out = F.bilinear_resize(aux_out, out_size)
                        loss = loss + F.log_softmax_nll(aux_out, target, ignore_index) * aux_weight
                _zero_grads(segmenter, (optim_dec,))
                with F.deferred_wgrad(params=dec_params, second_stream=False):  # (crops of the feature cache: launch-bound)
                    loss.backward()
            if _distributed(segmenter):
                # the feature cache is sharded: every rank steps on its own cached samples and the
                # decoder gradients are averaged (the reference runs this stage on one GPU)
                segmenter.sync_gradients()
            _clip_and_step([(dec_params, dec_grad_clip, optim_dec)])
        except Exception as e:
            if _syncs(segmenter) == syncs:
                _tell_peers(segmenter, e)  # (the peers are waiting in this step's collective)
            raise  # (a later failure - clipping, the optimiser - is the caller's to announce: train_task0)
        return loss

    return step


@try_except
def train_task0(Xy_train, segmenter, optim_dec, epoch, segm_crit, kd_crit, batch_size, freeze_bn,
                do_kd, kd_coeff, dec_grad_clip, do_polyak, avg_param=None, polyak_decay=0.9,
                aux_weight=0):
    """Decoder-only epoch over the cached encoder features (trainer.py:78-175)."""
    decoder = _inner(segmenter).decoder
    # (data parallel the cache is sharded: every rank must issue the same number of gradient
    #  all-reduces, so the shards agree on the smallest of their sizes first)
    n_examples = _agreed_count(segmenter, Xy_train[0].size(0))
    batch_size = min(batch_size, n_examples)
    n_passes = n_examples // batch_size
    indices = np.arange(n_examples)
    batch_time, losses = AverageMeter(), AverageMeter()
    decoder.train()
    if freeze_bn:
        _freeze_bn(decoder)
    np.random.shuffle(indices)
    step = make_task0_step(Xy_train, segmenter, optim_dec, batch_size, _ignore_index(segm_crit), dec_grad_clip,
                           aux_weight, freeze_bn, do_kd, kd_coeff, kd_crit)
    for i in range(n_passes):
        start = time.time()
        syncs = _syncs(segmenter)
        try:
            loss = step(indices[i * batch_size:(i + 1) * batch_size])
        except Exception as e:
            _tell_peers(segmenter, e, syncs, i == n_passes - 1)
            raise _as_rank_failure(segmenter, e)
        losses.update(_loss_value(segmenter, loss))
        batch_time.update(time.time() - start)
        if do_polyak:
            _polyak_update(decoder.parameters(), avg_param, polyak_decay)
    _epoch_handshake(segmenter)
    logger.info(" Train epoch: {}\tAvg. Loss: {:.3f}\tAvg. Time: {:.3f}".format(
        epoch, losses.avg, batch_time.avg))


def segmenter_step(segmenter, image, target, optim_enc, optim_dec, ignore_index=255,
                   enc_grad_clip=0.0, dec_grad_clip=0.0, aux_weight=-1):
    """One end-to-end training step on device tensors; returns the (device) loss.

    forward -> nearest-resize labels to the logits' size -> fused log-softmax/NLL
    (+ weighted aux heads) -> backward -> gradient all-reduce (if data parallel)
    -> per-sub-module norm clipping -> optimiser steps.
    """
    model = _inner(segmenter)
    cached = getattr(model, "_nasseg_step_params", None)
    if cached is None or cached[0] != TREE_VERSION[0]:
        # a candidate's module tree is fixed: walk it once, not every step (and again when a
        # sub-module was swapped - TemplateDecoder._reset_clf bumps the version)
        cached = (TREE_VERSION[0], (list(model.encoder.parameters()), list(model.decoder.parameters())))
        model._nasseg_step_params = cached
        model._nasseg_pack_memo = F.PackMemo()
    groups = cached[1]
    syncs = _syncs(segmenter)
    # the parameters are constant until the optimiser steps below: all chains' weights are
    # re-packed by one launch at the start of the step
    try:
        with F.packed_once(model._nasseg_pack_memo):
            output = segmenter(image)
            aux_outs = []
            if isinstance(output, tuple):
                output, aux_outs = output
            target = F.nearest_label_resize(target, output.size()[2:])
            loss = F.log_softmax_nll(output, target, ignore_index)
            if aux_weight > 0:
                for aux_out in aux_outs:
                    aux_out = F.bilinear_resize(aux_out, target.size()[1:])
                    loss = loss + F.log_softmax_nll(aux_out, target, ignore_index) * aux_weight
            _zero_grads(segmenter, (optim_enc, optim_dec))
            # gradients were just cleared: the second stages of all weight-gradient reductions run
            # batched when backward is through
            # (a step small enough to be worth replaying from a hipGraph is launch-bound when it is not: no second stream)
            side = image.shape[0] * image.shape[2] * image.shape[3] > _graphed().AUTO_GRAPH_MAX_PIXELS
            with F.deferred_wgrad(params=groups[0] + groups[1], second_stream=side):
                loss.backward()
        finish_step(segmenter, groups, optim_enc, optim_dec, enc_grad_clip, dec_grad_clip)
    except Exception as e:
        if _syncs(segmenter) == syncs:
            _tell_peers(segmenter, e)  # (the peers are waiting in this step's collective)
        raise  # (a later failure - clipping, the optimiser - is the caller's to announce: train_segmenter)
    return loss


def finish_step(segmenter, groups, optim_enc, optim_dec, enc_grad_clip, dec_grad_clip):
    """What follows backward in a training step (src/engine/trainer.py:255-268): average the
    gradients over the data-parallel ranks (one RCCL all-reduce; the reference's DataParallel
    sums replica gradients on GPU 0), clip the encoder's and the decoder's gradient norms
    separately, step both optimisers.  groups = (encoder parameters, decoder parameters)."""
    if hasattr(segmenter, "sync_gradients"):
        segmenter.sync_gradients()
    _clip_and_step([(groups[0], enc_grad_clip, optim_enc), (groups[1], dec_grad_clip, optim_dec)])


@try_except
def train_segmenter(segmenter, train_loader, optim_enc, optim_dec, epoch, segm_crit, freeze_bn,
                    enc_grad_clip, dec_grad_clip, do_polyak, print_every=10, aux_weight=-1,
                    avg_param=None, polyak_decay=0.99):
    """End-to-end epoch (trainer.py:179-283)."""
    _set_stage(train_loader, "train")
    segmenter.train()
    if freeze_bn:
        _freeze_bn(segmenter)
    batch_time, losses = AverageMeter(), AverageMeter()
    ignore = _ignore_index(segm_crit)
    device = _model_device(_inner(segmenter))
    # data parallel: the ranks agree on the number of steps (loaders of unequal length would leave
    # the longer ones waiting in an all-reduce), and a rank that fails ANYWHERE in its step - the
    # loader, the copy to the device, the optimiser - tells its peers before it leaves (_tell_peers)
    n_steps = _agreed_count(segmenter, len(train_loader) if hasattr(train_loader, "__len__") else None)
    batches = iter(train_loader)
    i = 0
    while n_steps is None or i < n_steps:
        start = time.time()
        syncs = _syncs(segmenter)
        try:
            try:
                sample = next(batches)
            except StopIteration:
                break
            image = _to_device_image(sample["image"], device)
            target = _labels(sample["mask"], device)
            stepper = None
            if _replays(segmenter, device, image.shape[0] * image.shape[2] * image.shape[3]):
                # launch-bound sizes: forward + loss + backward replayed from a hipGraph captured on
                # this candidate's first batch (bit-identical to the eager step)
                stepper = _segmenter_stepper(segmenter, image, target, optim_enc, optim_dec, ignore,
                                             enc_grad_clip, dec_grad_clip, aux_weight)
            if stepper is not None:
                loss = stepper.step(image, target)
            else:
                loss = segmenter_step(segmenter, image, target, optim_enc, optim_dec, ignore,
                                      enc_grad_clip, dec_grad_clip, aux_weight)
            if do_polyak:
                _polyak_update(segmenter.parameters(), avg_param, polyak_decay)
        except Exception as e:
            _tell_peers(segmenter, e, syncs, n_steps is not None and i == n_steps - 1)
            raise _as_rank_failure(segmenter, e)
        losses.update(_loss_value(segmenter, loss))
        batch_time.update(time.time() - start)
        if i % print_every == 0:
            logger.info(" Train epoch: {} [{}/{}]\tAvg. Loss: {:.3f}\tAvg. Time: {:.3f}".format(
                epoch, i, len(train_loader), losses.avg, batch_time.avg))
        i += 1
    _epoch_handshake(segmenter)
