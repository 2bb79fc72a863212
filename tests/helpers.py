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


_WCACHE = {}


def synthetic_weights(backbone, hp=None, seed=1, target_frac=0.05):
    """Seeded synthetic weights: He-normal conv kernels in Keras HWIO layout,
    BN gamma 1+-0.1, beta +-0.1, mean +-0.1, var in [0.5,1.5]; the background bias of the
    label heads is shifted (bisection on the oracle's logits of one image) so that about
    ``target_frac`` of the anchors have a non-background probability > 0.5 -- otherwise NMS
    would be a no-op with random weights.  Returns dict name -> float32 array."""
    from oracle import net_oracle as no
    hp = hp or hyper_params(backbone)
    key = (backbone, seed, target_frac, hp["total_labels"], hp["img_size"])
    if key in _WCACHE:
        return _WCACHE[key]
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in no.param_specs(backbone, hp):
        var = name.rsplit("/", 1)[1]
        if var == "kernel":
            fan_in = shape[0] * shape[1] * shape[2]
            w[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif var == "depthwise_kernel":
            w[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / 9.0)).astype(np.float32)
        elif var == "gamma":
            w[name] = rng.uniform(0.9, 1.1, shape).astype(np.float32)
        elif var in ("beta", "moving_mean"):
            w[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif var == "moving_variance":
            w[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif var == "scale":
            w[name] = np.full(shape, 20.0, np.float32)
        else:  # bias
            w[name] = rng.uniform(-0.05, 0.05, shape).astype(np.float32)
    # tame the head logits, then calibrate the background bias
    L = hp["total_labels"]
    for i in range(1, 7):
        w["%d_conv_label_output/kernel" % i] *= np.float32(0.5)
        w["%d_conv_boxes_output/kernel" % i] *= np.float32(0.25)
    acts = {}
    no.forward(backbone, hp, w, images(1, hp["img_size"], seed=0), acts)
    logits = acts["labels_head"][0].astype(np.float64)            # [N, L]

    def frac(t):
        lg = logits.copy()
        lg[:, 0] += t
        e = np.exp(lg - lg.max(-1, keepdims=True))
        p = e / e.sum(-1, keepdims=True)
        return float(((p.argmax(-1) != 0) & (p.max(-1) > 0.5)).mean())
    lo, hi = -50.0, 50.0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if frac(mid) > target_frac:
            lo = mid
        else:
            hi = mid
    t = np.float32(0.5 * (lo + hi))
    for i in range(1, 7):
        w["%d_conv_label_output/bias" % i][0::L] += t
    _WCACHE[key] = w
    return w
