"""Drop-in for the reference's ``utils/train_utils.py``: hyper-parameter table, LR
schedule, step size, the fit generator and the IoU-matching target assignment (one fused
HIP kernel instead of ~25 eager TF ops)."""
import math

import torch

import ssd_hip as _h
from utils import bbox_utils

# Configuration values of the reference (utils/train_utils.py:5-26): SSD300 with six feature
# maps; levels 2-4 use five aspect ratios (4 + 2 = 6 anchors), the others three (2 + 2 = 4).
_R3 = (1.0, 2.0, 1.0 / 2.0)
_R5 = _R3 + (3.0, 1.0 / 3.0)
_LEVEL_RATIOS = (_R3, _R5, _R5, _R5, _R3, _R3)
_FEATURE_MAPS = {"vgg16": (38, 19, 10, 5, 3, 1), "mobilenet_v2": (19, 10, 5, 3, 2, 1)}
SSD = {name: {"img_size": 300, "feature_map_shapes": list(fm), "aspect_ratios": [list(r) for r in _LEVEL_RATIOS]}
       for name, fm in _FEATURE_MAPS.items()}

_FIXED = (("iou_threshold", 0.5), ("neg_pos_ratio", 3), ("loc_loss_alpha", 1), ("variances", [0.1, 0.1, 0.2, 0.2]))
_LR_STEPS = ((100, 1e-3), (125, 1e-4))      # (first epoch NOT covered, learning rate)


def get_hyper_params(backbone, **kwargs):
    """The table entry of ``backbone`` plus the fixed training/inference constants
    (utils/train_utils.py:28-45).  As in the reference the returned dict IS the module-level
    entry (later mutations are visible to every holder), and a keyword only overrides a key
    that already exists, and only with a truthy value."""
    hp = SSD[backbone]
    hp.update((k, list(v) if isinstance(v, list) else v) for k, v in _FIXED)
    hp.update({k: v for k, v in kwargs.items() if k in hp and v})
    return hp


def scheduler(epoch):
    """1e-3 below epoch 100, 1e-4 below 125, 1e-5 afterwards (utils/train_utils.py:47-60)."""
    for end, lr in _LR_STEPS:
        if epoch < end:
            return lr
    return 1e-5


def get_step_size(total_items, batch_size):
    """ceil(total_items / batch_size) (utils/train_utils.py:62-71)."""
    return math.ceil(total_items / batch_size)


def generator(dataset, prior_boxes, hyper_params):
    """Endless ``(img, (bbox_deltas, bbox_labels))`` stream for ``model.fit``: every pass over
    ``dataset`` re-encodes the ground truth against the priors (utils/train_utils.py:73-88)."""
    while True:
        for img, gt_boxes, gt_labels in dataset:
            yield img, calculate_actual_outputs(prior_boxes, gt_boxes, gt_labels, hyper_params)


def calculate_actual_outputs(prior_boxes, gt_boxes, gt_labels, hyper_params, return_indices=False):
    """reference utils/train_utils.py:90-127 -> (bbox_deltas [B,N,4], bbox_labels [B,N,L]).

    With ``return_indices`` also returns the int32 label index and matched-GT index per
    prior (bit-exact quantities of the parity contract)."""
    total_labels = int(hyper_params["total_labels"])
    iou_threshold = float(hyper_params["iou_threshold"])
    var_p, _keep = _h.host4(hyper_params["variances"])
    p = _h.to_dev(prior_boxes)
    g = _h.to_dev(gt_boxes)
    gl = _h.to_dev(gt_labels, torch.int32)
    if p.dim() != 2 or p.shape[1] != 4 or g.dim() != 3 or g.shape[2] != 4 or gl.shape != g.shape[:2]:
        raise ValueError("bad shapes %s / %s / %s" % (tuple(p.shape), tuple(g.shape), tuple(gl.shape)))
    B, G = g.shape[0], g.shape[1]
    N = p.shape[0]
    dev = p.device
    deltas = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    labels = torch.empty((B, N, total_labels), dtype=torch.float32, device=dev)
    lab_idx = torch.empty((B, N), dtype=torch.int32, device=dev)
    match_idx = torch.empty((B, N), dtype=torch.int32, device=dev)
    _h.check(_h.lib().ssd_match_encode(_h.ptr(p), _h.ptr(g), _h.ptr(gl), var_p, iou_threshold,
                                       B, N, G, total_labels, _h.ptr(deltas), _h.ptr(lab_idx),
                                       _h.ptr(match_idx), _h.ptr(labels), _h.stream()),
             "calculate_actual_outputs")
    if return_indices:
        return deltas, labels, lab_idx, match_idx
    return deltas, labels
