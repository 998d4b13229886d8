"""ctypes front-end of oracle/miou.c (same call signatures as the reference's
helpers.miou_utils: numpy in, numpy out)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmiou_oracle.so")
_lib = None


def build():
    src = os.path.join(_HERE, "miou.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmiou_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_compute_ius_accs.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fast_cm(preds, gt, n_classes):
    preds = np.ascontiguousarray(preds, dtype=np.uint8).reshape(-1)
    gt = np.ascontiguousarray(gt, dtype=np.uint8).reshape(-1)
    cm = np.zeros((n_classes, n_classes), dtype=np.int64)
    _load().oracle_fast_cm(_p(preds), _p(gt), ctypes.c_int64(gt.shape[0]), ctypes.c_int(n_classes),
                           _p(cm))
    return cm


def compute_ius_accs(cm):
    cm = np.ascontiguousarray(cm, dtype=np.int64)
    n = cm.shape[0]
    iu, acc = np.empty(n, np.float64), np.empty(n, np.float64)
    npx = np.empty(n, np.int64)
    if _load().oracle_compute_ius_accs(_p(cm), ctypes.c_int(n), _p(iu), _p(npx), _p(acc)) != 0:
        raise OverflowError("value too large to convert to unsigned int")
    return iu, npx, acc


def compute_iu(cm):
    return compute_ius_accs(cm)[0]
