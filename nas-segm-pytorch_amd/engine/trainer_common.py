"""Helpers shared by the eager steps (trainer.py) and the hipGraph steppers (graphed.py)."""
from torch import nn


def inner(segmenter):
    return segmenter.module if hasattr(segmenter, "module") else segmenter


def clip_and_step(groups, native=None):
    """groups: [(parameters, max_norm, optimiser)] - per-sub-module gradient-norm clipping, then
    the optimiser steps (src/engine/trainer.py:163-166,258-268).  Plain torch.optim.SGD / Adam objects on a HIP
    device are stepped by nasseg_optim_step (engine/optim_native.py: two launches, the optimisers' own state);
    anything else by torch.  ``native``: the NativeStep to use (a hipGraph stepper's own) instead of the one
    cached on the optimisers."""
    if native is not None:
        native.step()
        return
    # (a caller may hand generators - model.parameters(): both the native path and torch's walk them)
    groups = [(p if isinstance(p, (list, tuple)) else list(p), m, o) for p, m, o in groups]
    from .optim_native import native_clip_and_step

    if native_clip_and_step(groups):
        return
    for params, max_norm, _ in groups:
        if max_norm > 0:
            nn.utils.clip_grad_norm_(params, max_norm)
    for _, _, optim in groups:
        if optim is not None:
            optim.step()
