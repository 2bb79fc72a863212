"""Seeded synthetic inputs shared by the tests, smoke() and bench.py."""
import numpy as np

ASPECT_RATIOS = [[1., 2., 1. / 2.]] + [[1., 2., 1. / 2., 3., 1. / 3.]] * 3 + [[1., 2., 1. / 2.]] * 2
FMAPS = {"mobilenet_v2": [19, 10, 5, 3, 2, 1], "vgg16": [38, 19, 10, 5, 3, 1]}
VARIANCES = [0.1, 0.1, 0.2, 0.2]


def hyper_params(backbone="mobilenet_v2", total_labels=21):
    return {"img_size": 300, "feature_map_shapes": list(FMAPS[backbone]),
            "aspect_ratios": [list(a) for a in ASPECT_RATIOS], "iou_threshold": 0.5,
            "neg_pos_ratio": 3, "loc_loss_alpha": 1, "variances": list(VARIANCES),
            "total_labels": total_labels}


def decoder_inputs(B, N, L=21, seed=2, boost_frac=0.10, boost=6.0):
    """deltas ~ N(0,1); logits ~ N(0,1) with +boost on one random non-bg class for a random
    boost_frac of the anchors; probs = softmax(logits)."""
    rng = np.random.default_rng(seed)
    deltas = rng.standard_normal((B, N, 4)).astype(np.float32)
    logits = rng.standard_normal((B, N, L)).astype(np.float32)
    m = rng.random((B, N)) < boost_frac
    cls = rng.integers(1, L, (B, N))
    bi, ni = np.nonzero(m)
    logits[bi, ni, cls[bi, ni]] += np.float32(boost)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    probs = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return deltas, probs


def gt_inputs(B, G=16, L=21, seed=3):
    """G padded GT boxes per image: centres U(.1,.9), sizes U(.05,.5), labels U{1..L-1};
    padding rows are 0 / -1 like the reference's padded_batch (utils/data_utils.py:117-122)."""
    rng = np.random.default_rng(seed)
    gt = np.zeros((B, G, 4), np.float32)
    gl = -np.ones((B, G), np.int32)
    for b in range(B):
        g = int(rng.integers(1, G + 1))
        c = rng.uniform(0.1, 0.9, (g, 2))
        s = rng.uniform(0.05, 0.5, (g, 2))
        gt[b, :g] = np.clip(np.concatenate([c - s / 2, c + s / 2], -1), 0, 1).astype(np.float32)
        gl[b, :g] = rng.integers(1, L, g)
    return gt, gl


def images(B, S=300, seed=0):
    return np.random.default_rng(seed).random((B, S, S, 3), dtype=np.float32)
