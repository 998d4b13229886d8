"""Candidate evaluation for the NAS outer loop (BASELINE config 4): the controller stays
on the host (rank 0) and samples genotypes; every sampled decoder is built, trained for a
few steps and validated on ONE GPU, several candidates in parallel one per rank, and the
rewards are gathered back to rank 0 where ``train_agent`` consumes them
(src/main_search.py:490-513,543-660; src/rl/agent.py:73-77).  No gradient exchange is
needed in this mode - candidates are independent.
"""
import torch
import torch.distributed as dist

from ..nn.encoders import create_encoder
from ..nn.micro_decoders import MicroDecoder, TemplateDecoder
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
    return RankParallel(Segmenter(encoder, decoder).to(device), broadcast=False)


def evaluate_candidate(config, train_batches, val_batches, ctrl_version="wacv", num_classes=19,
                       agg_size=48, aux_cell=True, repeats=1, epochs=1, aux_weight=0.15,
                       omit_classes=(0,), device="cuda"):
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
    for epoch in range(epochs):
        ret = train_segmenter(segmenter, train_batches, optim_enc, optim_dec, epoch, _Crit(), False,
                              3.0, 3.0, False, print_every=10 ** 9,
                              aux_weight=aux_weight if ctrl_version == "cvpr" else -1)
        if ret == 0:  # try_except: RuntimeError inside the step
            return 0.0
    reward = validate(segmenter, val_batches, 0, 0, num_classes=num_classes, print_every=10 ** 9,
                      omit_classes=list(omit_classes))
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
