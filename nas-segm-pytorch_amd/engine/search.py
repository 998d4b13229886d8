"""Candidate evaluation for the NAS outer loop (BASELINE config 4): the controller stays
on the host (rank 0) and samples genotypes; every sampled decoder is built, trained for a
few steps and validated on ONE GPU, several candidates in parallel one per rank, and the
rewards are gathered back to rank 0 where ``train_agent`` consumes them
(src/main_search.py:490-513,543-660; src/rl/agent.py:73-77).  No gradient exchange is
needed in this mode - candidates are independent.
"""
import time

import torch
import torch.distributed as dist

from ..helpers.utils import compute_params
from ..nn.encoders import create_encoder
from ..nn.micro_decoders import MicroDecoder, TemplateDecoder
from .graphed import GraphedSegmenterStep
from .inference import validate
from .segmenter import RankParallel, Segmenter
from .trainer import train_segmenter


class _Crit(object):
    ignore_index = 255


def build_candidate(config, ctrl_version="wacv", num_classes=19, agg_size=48, aux_cell=True,
                    repeats=1, device="cuda"):
    """Encoder + decoder for one sampled genotype (create_segmenter, main_search.py:490-513).
    The encoder's ``out_sizes`` list is copied: MicroDecoder overwrites its argument."""
    encoder = create_encoder(pretrained=False, ctrl_version=ctrl_version)
    Decoder = MicroDecoder if ctrl_version == "cvpr" else TemplateDecoder
    decoder = Decoder(inp_sizes=list(encoder.out_sizes), num_classes=num_classes, config=config,
                      agg_size=agg_size, aux_cell=aux_cell, repeats=repeats)
    return RankParallel(Segmenter(encoder, decoder).to(device), independent=True)


def evaluate_candidate(config, train_batches, val_batches, ctrl_version="wacv", num_classes=19,
                       agg_size=48, aux_cell=True, repeats=1, epochs=1, aux_weight=0.15,
                       omit_classes=(0,), device="cuda", stats=None, graphed=False):
    """Train the candidate on ``train_batches`` (lists of {"image", "mask"}) for ``epochs``
    passes and return its validation reward; failures score 0 like in the reference."""
    try:
        segmenter = build_candidate(config, ctrl_version, num_classes, agg_size, aux_cell, repeats,
                                    device)
    except RuntimeError:
        return 0.0
    model = segmenter.module
    optim_enc = torch.optim.SGD(model.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    optim_dec = torch.optim.Adam(model.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
    aux = aux_weight if ctrl_version == "cvpr" else -1
    if graphed:
        # one capture per candidate, one replay per step (engine/graphed.py): same results, no host
        # launch cost - what bounds a candidate at 321x321 ... 713x713
        try:
            first = train_batches[0]
            image = first["image"].to(device=device, dtype=torch.float32).contiguous(
                memory_format=torch.channels_last)
            stepper = GraphedSegmenterStep(segmenter, image, first["mask"].to(device).long(), optim_enc,
                                           optim_dec, 255, 3.0, 3.0, aux)
            for epoch in range(epochs):
                for sample in train_batches:
                    stepper.step(sample["image"].to(device=device, dtype=torch.float32).contiguous(
                        memory_format=torch.channels_last), sample["mask"].to(device).long())
        except RuntimeError:  # scored 0, as the reference's try_except does
            return 0.0
    else:
        for epoch in range(epochs):
            ret = train_segmenter(segmenter, train_batches, optim_enc, optim_dec, epoch, _Crit(), False,
                                  3.0, 3.0, False, print_every=10 ** 9, aux_weight=aux)
            if ret == 0:  # try_except: RuntimeError inside the step
                return 0.0
    reward = validate(segmenter, val_batches, 0, 0, num_classes=num_classes, print_every=10 ** 9,
                      omit_classes=list(omit_classes))
    if stats is not None:
        stats["params"] = compute_params(segmenter)[1]
    return float(reward)


def evaluate_candidates(configs, make_batches, **kwargs):
    """Evaluate ``configs[rank::world]`` on this rank and return the full reward list on rank 0
    (None elsewhere).  ``make_batches(rank)`` -> (train_batches, val_batches)."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    train_batches, val_batches = make_batches(rank)
    mine = {i: evaluate_candidate(configs[i], train_batches, val_batches, **kwargs)
            for i in range(rank, len(configs), world)}
    if world == 1:
        return [mine[i] for i in range(len(configs))]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    return [merged[i] for i in range(len(configs))]


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def search_loop(sample_fn, train_agent_fn, evaluate_fn, n_iterations, arch_writer=None, logger=None,
                first_epoch=0):
    """Outer loop of the search with one candidate per GPU (BASELINE config 4).

    The reference evaluates ONE sampled architecture per outer epoch on all GPUs
    (src/main_search.py:543-674); here every outer iteration evaluates ``world`` of them, one
    per rank, and the controller takes ``world`` policy-gradient steps:

      rank 0   : ``sample_fn()`` x world -> (config, entropy, log_prob) each (the reference's
                 ``agent.controller.sample()``, micro_controllers.py:136-139) - the controller
                 and its RNG live on rank 0 only;
      all ranks: receive their config (broadcast of python lists), ``evaluate_fn(config)`` ->
                 reward or (reward, n_params) (``evaluate_candidate``: RuntimeError => 0);
      rank 0   : gathers the rewards and calls ``train_agent_fn((config, reward, entropy,
                 log_prob))`` once per candidate, in sampling order (rl/agent.py:73-77: REINFORCE
                 step or PPO RolloutStorage insert), and appends the reference's genotype log
                 line per candidate (main_search.py:664-674; the format helpers/num_uq.py parses).

    Returns, on rank 0, the list of (config, reward) in evaluation order; None elsewhere.
    """
    world, rank = _world()
    history = []
    for it in range(n_iterations):
        t0 = time.time()
        samples = [sample_fn() for _ in range(world)] if rank == 0 else None
        if world > 1:
            box = [[s[0] for s in samples]] if rank == 0 else [None]
            dist.broadcast_object_list(box, src=0)
            config = box[0][rank]
        else:
            config = samples[0][0]
        result = evaluate_fn(config)
        reward, params = result if isinstance(result, tuple) else (result, -1)
        mine = (float(reward), int(params))
        if world > 1:
            gathered = [None] * world if rank == 0 else None
            dist.gather_object(mine, gathered, dst=0)
        else:
            gathered = [mine]
        if rank != 0:
            continue
        per_arch = (time.time() - t0)
        for k, ((cfg, entropy, log_prob), (rew, par)) in enumerate(zip(samples, gathered)):
            train_agent_fn((cfg, rew, entropy, log_prob))
            epoch = first_epoch + it * world + k
            if logger is not None:
                logger.info(" Decoder: {}".format(cfg))
            if arch_writer is not None:
                arch_writer.write(
                    "reward: {:.4f}, epoch: {}, params: {}, epoch_time: {:.4f}, genotype: {}\n".format(
                        rew, epoch, par, per_arch, cfg))
                arch_writer.flush()
            history.append((cfg, rew))
    return history if rank == 0 else None
