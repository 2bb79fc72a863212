"""Drop-in for the reference's ``utils/train_utils.py``: hyper-parameter table, LR
schedule, step size, the fit generator and the IoU-matching target assignment (one fused
HIP kernel instead of ~25 eager TF ops)."""
import math

import torch

import ssd_hip as _h
from utils import bbox_utils

# reference utils/train_utils.py:5-26 (verbatim configuration values)
SSD = {
    "vgg16": {
        "img_size": 300,
        "feature_map_shapes": [38, 19, 10, 5, 3, 1],
        "aspect_ratios": [[1., 2., 1./2.],
                         [1., 2., 1./2., 3., 1./3.],
                         [1., 2., 1./2., 3., 1./3.],
                         [1., 2., 1./2., 3., 1./3.],
                         [1., 2., 1./2.],
                         [1., 2., 1./2.]],
    },
    "mobilenet_v2": {
        "img_size": 300,
        "feature_map_shapes": [19, 10, 5, 3, 2, 1],
        "aspect_ratios": [[1., 2., 1./2.],
                         [1., 2., 1./2., 3., 1./3.],
                         [1., 2., 1./2., 3., 1./3.],
                         [1., 2., 1./2., 3., 1./3.],
                         [1., 2., 1./2.],
                         [1., 2., 1./2.]],
    }
}


def get_hyper_params(backbone, **kwargs):
    """reference utils/train_utils.py:28-45: returns (and mutates) the global table entry;
    kwargs override only keys that already exist and only with truthy values."""
    hyper_params = SSD[backbone]
    hyper_params["iou_threshold"] = 0.5
    hyper_params["neg_pos_ratio"] = 3
    hyper_params["loc_loss_alpha"] = 1
    hyper_params["variances"] = [0.1, 0.1, 0.2, 0.2]
    for key, value in kwargs.items():
        if key in hyper_params and value:
            hyper_params[key] = value
    return hyper_params


def scheduler(epoch):
    """reference utils/train_utils.py:47-60."""
    if epoch < 100:
        return 1e-3
    elif epoch < 125:
        return 1e-4
    else:
        return 1e-5


def get_step_size(total_items, batch_size):
    """reference utils/train_utils.py:62-71."""
    return math.ceil(total_items / batch_size)


def generator(dataset, prior_boxes, hyper_params):
    """reference utils/train_utils.py:73-88: endless (img, (deltas, labels)) generator."""
    while True:
        for image_data in dataset:
            img, gt_boxes, gt_labels = image_data
            actual_deltas, actual_labels = calculate_actual_outputs(prior_boxes, gt_boxes, gt_labels, hyper_params)
            yield img, (actual_deltas, actual_labels)


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
