"""Drop-in for the reference's ``utils/bbox_utils.py`` (same names, argument meaning and
error behaviour) with every array op executed by HIP kernels through ``ssd_hip``.

Inputs may be NumPy arrays, lists or torch tensors (any device); outputs are torch
tensors on the GPU (the analogue of the ``tf.Tensor`` the reference returns).
"""
import ctypes

import numpy as np
import torch

import ssd_hip as _h


def non_max_suppression(pred_bboxes, pred_labels, **kwargs):
    """reference utils/bbox_utils.py:3-25 -> ``tf.image.combined_non_max_suppression``.

    pred_bboxes [B,N,1,4], pred_labels [B,N,C]; kwargs as TF's: max_output_size_per_class,
    max_total_size, iou_threshold=0.5, score_threshold=-inf, pad_per_class=False,
    clip_boxes=True.  Returns (nmsed_boxes, nmsed_scores, nmsed_classes, valid_detections).
    """
    try:
        max_per_class = int(kwargs.pop("max_output_size_per_class"))
        max_total = int(kwargs.pop("max_total_size"))
    except KeyError as e:
        raise TypeError("combined_non_max_suppression() missing required argument %s" % e)
    iou_thr = float(kwargs.pop("iou_threshold", 0.5))
    score_thr = float(kwargs.pop("score_threshold", float("-inf")))
    pad_per_class = bool(kwargs.pop("pad_per_class", False))
    clip_boxes = bool(kwargs.pop("clip_boxes", True))
    kwargs.pop("name", None)
    if kwargs:
        raise TypeError("unexpected keyword arguments: %s" % sorted(kwargs))
    if pad_per_class:
        raise NotImplementedError("pad_per_class=True is never used by the reference")
    boxes = _h.to_dev(pred_bboxes)
    scores = _h.to_dev(pred_labels)
    if boxes.dim() != 4 or scores.dim() != 3:
        raise ValueError("boxes must be [B,N,q,4] and scores [B,N,C]")
    B, N, q, four = boxes.shape
    if four != 4 or scores.shape[0] != B or scores.shape[1] != N:
        raise ValueError("incompatible shapes %s / %s" % (tuple(boxes.shape), tuple(scores.shape)))
    C = scores.shape[2]
    if q != 1:
        raise NotImplementedError("per-class boxes (q == C) are never used by the reference")
    dev = boxes.device
    ob = torch.empty((B, max_total, 4), dtype=torch.float32, device=dev)
    osc = torch.empty((B, max_total), dtype=torch.float32, device=dev)
    oc = torch.empty((B, max_total), dtype=torch.float32, device=dev)
    valid = torch.empty((B,), dtype=torch.int32, device=dev)
    nbytes = _h.lib().ssd_decode_nms_workspace_bytes(B, N, C, max_per_class)
    ws = _h.workspace(nbytes)
    _h.check(_h.lib().ssd_combined_nms(
        _h.ptr(boxes), _h.ptr(scores), B, N, C, max_per_class, max_total, iou_thr, score_thr,
        int(clip_boxes), _h.ptr(ob), _h.ptr(osc), _h.ptr(oc), _h.ptr(valid), _h.vp(0),
        _h.ptr(ws), ws.numel(), _h.stream()), "non_max_suppression")
    return ob, osc, oc, valid


def generate_iou_map(bboxes, gt_boxes, transpose_perm=[0, 2, 1]):
    """reference utils/bbox_utils.py:27-59.  bboxes [N,4] or [B,N,4]; gt_boxes [B,G,4]
    (or [G,4] with 2-d bboxes) -> iou_map [..., N, G]; no epsilon (0/0 -> NaN)."""
    b = _h.to_dev(bboxes)
    g = _h.to_dev(gt_boxes)
    squeeze = False
    if g.dim() == 2:
        if b.dim() != 2:
            raise ValueError("2-d gt_boxes need 2-d bboxes")
        g = g.unsqueeze(0)
        squeeze = True
    if g.dim() != 3 or g.shape[-1] != 4 or b.shape[-1] != 4 or b.dim() not in (2, 3):
        raise ValueError("bad shapes %s / %s" % (tuple(b.shape), tuple(g.shape)))
    B, G = g.shape[0], g.shape[1]
    batched = int(b.dim() == 3)
    if batched and b.shape[0] != B:
        raise ValueError("batch mismatch")
    N = b.shape[-2]
    out = torch.empty((B, N, G), dtype=torch.float32, device=g.device)
    _h.check(_h.lib().ssd_iou_map(_h.ptr(b), batched, _h.ptr(g), B, N, G, _h.ptr(out), _h.stream()),
             "generate_iou_map")
    return out[0] if squeeze else out


def get_bboxes_from_deltas(prior_boxes, deltas):
    """reference utils/bbox_utils.py:61-85: priors [N,4], deltas [B,N,4] -> [B,N,4]."""
    p = _h.to_dev(prior_boxes)
    d = _h.to_dev(deltas)
    squeeze = d.dim() == 2
    if squeeze:
        d = d.unsqueeze(0)
    if p.dim() != 2 or d.dim() != 3 or p.shape[0] != d.shape[1] or p.shape[1] != 4 or d.shape[2] != 4:
        raise ValueError("bad shapes %s / %s" % (tuple(p.shape), tuple(d.shape)))
    out = torch.empty_like(d)
    _h.check(_h.lib().ssd_decode_boxes(_h.ptr(p), _h.ptr(d), None, d.shape[0], d.shape[1],
                                       _h.ptr(out), _h.stream()), "get_bboxes_from_deltas")
    return out[0] if squeeze else out


def get_deltas_from_bboxes(bboxes, gt_boxes):
    """reference utils/bbox_utils.py:87-113: bboxes [N,4], gt_boxes [B,N,4] -> [B,N,4]."""
    b = _h.to_dev(bboxes)
    g = _h.to_dev(gt_boxes)
    squeeze = g.dim() == 2
    if squeeze:
        g = g.unsqueeze(0)
    if b.dim() != 2 or g.dim() != 3 or b.shape[0] != g.shape[1] or b.shape[1] != 4 or g.shape[2] != 4:
        raise ValueError("bad shapes %s / %s" % (tuple(b.shape), tuple(g.shape)))
    out = torch.empty_like(g)
    _h.check(_h.lib().ssd_encode_deltas(_h.ptr(b), _h.ptr(g), g.shape[0], g.shape[1], _h.ptr(out),
                                        _h.stream()), "get_deltas_from_bboxes")
    return out[0] if squeeze else out


def get_scale_for_nth_feature_map(k, m=6, scale_min=0.2, scale_max=0.9):
    """reference utils/bbox_utils.py:115-124 (host float64 arithmetic)."""
    return scale_min + ((scale_max - scale_min) / (m - 1)) * (k - 1)


def generate_base_prior_boxes(aspect_ratios, feature_map_index, total_feature_map):
    """reference utils/bbox_utils.py:126-147.  Host-side helper kept for API parity: the
    hot path (``generate_prior_boxes``) computes the base boxes inside the HIP kernel."""
    f32 = np.float32
    cur = get_scale_for_nth_feature_map(feature_map_index, m=total_feature_map)
    nxt = get_scale_for_nth_feature_map(feature_map_index + 1, m=total_feature_map)
    rows = []
    for ar in aspect_ratios:
        s = np.sqrt(f32(ar))
        h, w = f32(cur) / s, f32(cur) * s
        rows.append([-h / f32(2), -w / f32(2), h / f32(2), w / f32(2)])
    h = w = np.sqrt(f32(cur * nxt))
    rows.append([-h / f32(2), -w / f32(2), h / f32(2), w / f32(2)])
    return torch.as_tensor(np.asarray(rows, dtype=f32))


def generate_prior_boxes(feature_map_shapes, aspect_ratios):
    """reference utils/bbox_utils.py:149-176 -> prior_boxes [N,4] in [0,1] (GPU tensor)."""
    levels = len(feature_map_shapes)
    if len(aspect_ratios) != levels:
        raise ValueError("feature_map_shapes and aspect_ratios differ in length")
    get_scale_for_nth_feature_map(1, m=levels)   # raises ZeroDivisionError for m == 1 like the reference
    fm = (ctypes.c_int * levels)(*[int(f) for f in feature_map_shapes])
    na = (ctypes.c_int * levels)(*[len(a) for a in aspect_ratios])
    rows = [(ctypes.c_float * max(len(a), 1))(*[float(x) for x in a]) for a in aspect_ratios]
    arp = (_h.c_float_p * levels)(*[ctypes.cast(r, _h.c_float_p) for r in rows])
    n = _h.lib().ssd_priors_count(fm, na, levels)
    if n < 0:
        raise ValueError("bad prior-box configuration")
    out = torch.empty((n, 4), dtype=torch.float32, device=_h.device())
    _h.check(_h.lib().ssd_priors(fm, arp, na, levels, _h.ptr(out), _h.stream()), "generate_prior_boxes")
    return out


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))


def renormalize_bboxes_with_min_max(bboxes, min_max):
    """reference utils/bbox_utils.py:178-188 (API-surface helper; elementwise torch ops)."""
    b, mm = _t(bboxes).float(), _t(min_max).float().to(_t(bboxes).device)
    y_min, x_min, y_max, x_max = torch.split(mm, 1)
    r = b - torch.cat([y_min, x_min, y_min, x_min], -1)
    r = r / torch.cat([y_max - y_min, x_max - x_min, y_max - y_min, x_max - x_min], -1)
    return torch.clamp(r, 0, 1)


def normalize_bboxes(bboxes, height, width):
    """reference utils/bbox_utils.py:190-205."""
    b = _t(bboxes).float()
    return torch.stack([b[..., 0] / height, b[..., 1] / width, b[..., 2] / height, b[..., 3] / width], -1)


def denormalize_bboxes(bboxes, height, width):
    """reference utils/bbox_utils.py:207-222 (tf.round == round-half-to-even == torch.round)."""
    b = _t(bboxes).float()
    return torch.round(torch.stack([b[..., 0] * height, b[..., 1] * width,
                                    b[..., 2] * height, b[..., 3] * width], -1))
