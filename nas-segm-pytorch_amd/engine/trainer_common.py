"""Helpers shared by the eager steps (trainer.py) and the hipGraph steppers (graphed.py)."""
from torch import nn


def inner(segmenter):
    return segmenter.module if hasattr(segmenter, "module") else segmenter


def clip_and_step(groups):
    """groups: [(parameters, max_norm, optimiser)] - per-sub-module gradient-norm clipping, then
    the optimiser steps (src/engine/trainer.py:163-166,258-268)"""
    for params, max_norm, _ in groups:
        if max_norm > 0:
            nn.utils.clip_grad_norm_(params, max_norm)
    for _, _, optim in groups:
        if optim is not None:
            optim.step()
