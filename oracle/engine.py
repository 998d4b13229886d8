"""Restatement of the engine semantics (reference src/engine/trainer.py,
src/engine/inference.py) on CPU: loss, one optimisation step, validation reward."""
import numpy as np
import torch
import torch.nn.functional as F

from . import miou


def segm_loss(logits, target, ignore_index=255):
    """nn.NLLLoss2d(ignore_index)(nn.LogSoftmax()(logits), target) - trainer.py:239-241."""
    return F.nll_loss(F.log_softmax(logits, dim=1), target, ignore_index=ignore_index)


def nearest_labels(target, size):
    """trainer.py:236-238: float -> nearest interpolate -> long."""
    t = F.interpolate(target[:, None].float(), size=tuple(size), mode="nearest")
    return t.long()[:, 0]


def train_loss(output, target, aux_weight=-1, ignore_index=255):
    """Loss of train_segmenter (trainer.py:233-250) given the network output."""
    aux_outs = []
    if isinstance(output, tuple):
        output, aux_outs = output
    tgt = nearest_labels(target, output.shape[2:])
    loss = segm_loss(output, tgt, ignore_index)
    if aux_weight > 0:
        for a in aux_outs:
            a = F.interpolate(a, size=tuple(tgt.shape[1:]), mode="bilinear", align_corners=False)
            loss = loss + segm_loss(a, tgt, ignore_index) * aux_weight
    return loss


def predictions(logits, size):
    """inference.py:58-62: bilinear up-sample, argmax over classes, uint8."""
    up = F.interpolate(logits, size=tuple(size), mode="bilinear", align_corners=False)
    return up.detach().numpy().argmax(axis=1).astype(np.uint8)


def confusion(logits, gt, num_classes):
    """inference.py:58-66 for one batch -> (n, n) int64."""
    pred = predictions(logits, gt.shape[1:])
    gt = np.asarray(gt).astype(np.uint8)
    keep = gt < num_classes
    return miou.fast_cm(pred[keep], gt[keep], num_classes)


def reward_from_cm(cm, omit_classes=(0,)):
    """inference.py:78-91 -> (reward, miou, macc, mfwiou)."""
    ious, n_pixels, accs = miou.compute_ius_accs(cm)
    present = np.array([i for i, iu in enumerate(ious) if iu <= 1.0])
    present = np.setdiff1d(present, list(omit_classes))
    pi, pp, pa = ious[present], n_pixels[present], accs[present]
    m = [np.mean(pi), np.mean(pa), np.sum(pi * pp) / np.sum(pp)]
    return np.prod(m) ** (1.0 / len(m)), m[0], m[1], m[2]
