"""Validation -> reward (mirrors src/engine/inference.py:18-97).

The reference copies full-resolution logits to the host, arg-maxes with numpy
and loops over pixels in Cython.  Here the bilinear up-sampling to label size,
the argmax (lowest index wins ties), the ``gt < num_classes`` filter and the
confusion-matrix update are one HIP kernel; only the (n, n) int64 matrix leaves
the device.  The metric arithmetic afterwards is unchanged.
"""
import logging

import numpy as np
import torch

from .. import functional as F
from ..helpers.miou_utils import compute_iu, compute_ius_accs
from ..helpers.utils import try_except

logger = logging.getLogger(__name__)


def reward_from_cm(cm, omit_classes=(0,)):
    """(reward, miou, macc, mfwiou): geometric mean of mean-IoU, mean accuracy and
    frequency-weighted IoU over classes that are present (IoU <= 1, the 2.0
    sentinel marks absent ones) and not omitted (inference.py:78-91)."""
    ious, n_pixels, accs = compute_ius_accs(cm)
    present = np.array([i for i, iu in enumerate(ious) if iu <= 1.0])
    present = np.setdiff1d(present, list(omit_classes))
    p_ious, p_pix, p_accs = ious[present], n_pixels[present], accs[present]
    miou = np.mean(p_ious)
    macc = np.mean(p_accs)
    mfwiou = np.sum(p_ious * p_pix) / np.sum(p_pix)
    metrics = [miou, macc, mfwiou]
    reward = np.prod(metrics) ** (1.0 / len(metrics))
    return reward, miou, macc, mfwiou


@try_except
def validate(segmenter, val_loader, epoch, epoch2, num_classes=-1, print_every=10,
             omit_classes=[0]):
    """Evaluate the candidate; returns the scalar reward."""
    ds = getattr(val_loader, "dataset", None)
    if ds is not None:
        try:
            ds.set_stage("val")
        except AttributeError:
            sub = getattr(ds, "dataset", None)
            if sub is not None and hasattr(sub, "set_stage"):
                sub.set_stage("val")
    segmenter.eval()
    model = segmenter.module if hasattr(segmenter, "module") else segmenter
    device = next(model.parameters()).device
    cm = torch.zeros((num_classes, num_classes), device=device, dtype=torch.int64)
    try:
        with torch.no_grad():
            for i, sample in enumerate(val_loader):
                image = sample["image"].to(device=device, dtype=torch.float32).contiguous(
                    memory_format=torch.channels_last)
                gt = sample["mask"].to(device).to(torch.uint8)  # astype(np.uint8) in the reference
                output = segmenter(image)
                if isinstance(output, tuple):
                    output, _ = output
                F.argmax_confusion(output, gt, num_classes, cm=cm)
                if i % print_every == 0:
                    logger.info(" Val epoch: {} [{}/{}]\tMean IoU: {:.3f}".format(
                        epoch, i, len(val_loader),
                        np.mean([iu for iu in compute_iu(cm) if iu <= 1.0])))
    except Exception:  # (not only RuntimeError: a loader error on one rank must not strand its peers)
        # data parallel: the peers will wait in the confusion-matrix all-reduce - take part in
        # it with the failure flag set so that every rank scores this candidate 0
        if hasattr(segmenter, "reduce_confusion"):
            segmenter.reduce_confusion(cm, failed=True)
        raise
    if hasattr(segmenter, "reduce_confusion"):
        segmenter.reduce_confusion(cm)
    cm_host = cm.cpu().numpy()
    ious, _, accs = compute_ius_accs(cm_host)
    logger.info(" IoUs: {}, accs: {}".format(ious, accs))
    reward, miou, macc, mfwiou = reward_from_cm(cm_host, omit_classes)
    logger.info((" Val epoch: {}/{}\tMean IoU: {:.3f}\tMean FW-IoU: {:.3f}\t"
                 "Mean Acc: {:.3f}\tReward: {:.3f}").format(epoch, epoch2, miou, mfwiou, macc, reward))
    return reward
