"""ctypes loader for oracle/libssd_oracle.so (plain-C oracle; test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libssd_oracle.so")
_lib = None
F = ctypes.POINTER(ctypes.c_float)
I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    src = os.path.join(_HERE, "ssd_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libssd_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    return a.ctypes.data_as(F)


def _i(a):
    return a.ctypes.data_as(I)


def decode(priors, deltas, variances):
    priors = np.ascontiguousarray(priors, np.float32)
    deltas = np.ascontiguousarray(deltas, np.float32)
    var = np.ascontiguousarray(variances, np.float32)
    B, N, _ = deltas.shape
    out = np.empty_like(deltas)
    lib().oracle_decode(_f(priors), _f(deltas), _f(var), B, N, _f(out))
    return out


def decode_nms(deltas, probs, priors, variances, max_per_class=200, max_total=200,
               iou_thr=0.5, score_thr=0.5):
    deltas = np.ascontiguousarray(deltas, np.float32)
    probs = np.ascontiguousarray(probs, np.float32)
    priors = np.ascontiguousarray(priors, np.float32)
    var = np.ascontiguousarray(variances, np.float32)
    B, N, L = probs.shape
    boxes = np.empty((B, max_total, 4), np.float32)
    labels = np.empty((B, max_total), np.float32)
    scores = np.empty((B, max_total), np.float32)
    valid = np.empty((B,), np.int32)
    kept = np.empty((B, max_total), np.int32)
    rc = lib().oracle_decode_nms(_f(deltas), _f(probs), _f(priors), _f(var), B, N, L,
                                 int(max_per_class), int(max_total),
                                 ctypes.c_float(iou_thr), ctypes.c_float(score_thr),
                                 _f(boxes), _f(labels), _f(scores), _i(valid), _i(kept))
    assert rc == 0
    return boxes, labels, scores, valid, kept


def iou_map(priors, gt):
    priors = np.ascontiguousarray(priors, np.float32)
    gt = np.ascontiguousarray(gt, np.float32)
    B, G, _ = gt.shape
    N = priors.shape[0]
    out = np.empty((B, N, G), np.float32)
    lib().oracle_iou_map(_f(priors), _f(gt), B, N, G, _f(out))
    return out


def match_encode(priors, gt, gt_labels, variances, iou_thr=0.5):
    priors = np.ascontiguousarray(priors, np.float32)
    gt = np.ascontiguousarray(gt, np.float32)
    gl = np.ascontiguousarray(gt_labels, np.int32)
    var = np.ascontiguousarray(variances, np.float32)
    B, G, _ = gt.shape
    N = priors.shape[0]
    deltas = np.empty((B, N, 4), np.float32)
    lab = np.empty((B, N), np.int32)
    mi = np.empty((B, N), np.int32)
    lib().oracle_match_encode(_f(priors), _f(gt), _i(gl), _f(var), ctypes.c_float(iou_thr),
                              B, N, G, _f(deltas), _i(lab), _i(mi))
    return deltas, lab, mi
