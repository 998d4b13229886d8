"""berHu (reverse Huber) loss for the depth head of BASELINE config 5.

PARITY UNPINNED: the reference contains no berHu / Huber code (its depth
networks are inference only - tests/test_inference.py:116,186-187).  Restated
from Laina et al., "Deeper Depth Prediction with Fully Convolutional Residual
Networks" (3DV 2016), eq. (2):
    B(x) = |x|                 if |x| <= c
           (x^2 + c^2) / (2c)  otherwise,      c = 0.2 * max_i |x_i| over the batch
averaged over all elements.
"""
import torch


def berhu(pred, target):
    d = (pred - target).abs()
    c = 0.2 * d.max().detach()
    quad = (d * d + c * c) / (2.0 * c)
    return torch.where(d <= c, d, quad).mean()
