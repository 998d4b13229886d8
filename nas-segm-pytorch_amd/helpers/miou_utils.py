"""Confusion matrix / IoU utilities - drop-in for the reference's only native
module, the Cython ``helpers.miou_utils`` (src/helpers/miou_utils.pyx).

``fast_cm`` runs the histogram on the GPU (LDS-privatised bins + int64 atomics:
exact, order independent); ``compute_iu`` / ``compute_ius_accs`` are the C
restatement inside libnasseg_hip.so, including the 32-bit ``unsigned int``
intermediates and the 2.0 sentinel for absent classes.
"""
import ctypes

import numpy as np
import torch

from .._lib import NassegError, current_stream, lib, ptr


def _device():
    if not torch.cuda.is_available():
        raise NassegError("fast_cm needs a HIP device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _as_u8_device(a, name):
    if isinstance(a, torch.Tensor):
        if a.dtype != torch.uint8:
            raise ValueError("{} must be uint8".format(name))
        t = a.reshape(-1)
        return t.contiguous() if t.is_cuda else t.contiguous().to(_device())
    arr = np.ascontiguousarray(a)
    if arr.dtype != np.uint8 or arr.ndim != 1:
        # the Cython signature is `unsigned char[::1]`: wrong dtype / ndim is a ValueError there
        raise ValueError("Buffer dtype mismatch or wrong number of dimensions for {}".format(name))
    return torch.from_numpy(arr).to(_device())


def fast_cm(preds, gt, n_classes):
    """cm[gt[i], preds[i]] += 1 -> (n_classes, n_classes) int64.

    numpy in -> numpy out (reference contract, miou_utils.pyx:7-30); device
    tensors in -> device tensor out (no host round trip).
    """
    numpy_io = not (isinstance(preds, torch.Tensor) and preds.is_cuda)
    p = _as_u8_device(preds, "preds")
    g = _as_u8_device(gt, "gt")
    if p.numel() < g.numel():
        raise IndexError("preds shorter than gt")
    cm = torch.zeros((n_classes, n_classes), device=p.device, dtype=torch.int64)
    lib.call("nasseg_fast_cm", ptr(p), ptr(g), g.numel(), int(n_classes), ptr(cm), current_stream())
    return cm.cpu().numpy() if numpy_io else cm


def _host_cm(cm):
    if isinstance(cm, torch.Tensor):
        cm = cm.detach().cpu().numpy()
    cm = np.ascontiguousarray(cm, dtype=np.int64)
    if cm.ndim != 2 or cm.shape[0] != cm.shape[1]:
        raise ValueError("cm must be a square matrix")
    return cm


def compute_ius_accs(cm):
    """(IU float64[n], n_pixels int64[n], accs float64[n]) - miou_utils.pyx:59-90."""
    cm = _host_cm(cm)
    n = cm.shape[0]
    iu = np.empty(n, dtype=np.float64)
    acc = np.empty(n, dtype=np.float64)
    npx = np.empty(n, dtype=np.int64)
    try:
        lib.call("nasseg_compute_ius_accs", cm.ctypes.data_as(ctypes.c_void_p), n,
                 iu.ctypes.data_as(ctypes.c_void_p), npx.ctypes.data_as(ctypes.c_void_p),
                 acc.ctypes.data_as(ctypes.c_void_p))
    except NassegError as e:
        # Cython raises OverflowError when a count does not fit `unsigned int`
        raise OverflowError(str(e))
    return iu, npx, acc


def compute_iu(cm):
    """IU float64[n], 2.0 for classes absent from both gt and predictions (miou_utils.pyx:32-57)."""
    cm = _host_cm(cm)
    n = cm.shape[0]
    iu = np.empty(n, dtype=np.float64)
    try:
        lib.call("nasseg_compute_ius_accs", cm.ctypes.data_as(ctypes.c_void_p), n,
                 iu.ctypes.data_as(ctypes.c_void_p), None, None)
    except NassegError as e:
        raise OverflowError(str(e))
    return iu
