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
    _calibrate_heads(w, backbone, hp, target_frac)
    _WCACHE[key] = w
    return w


RESIDUAL_BLOCKS = (2, 4, 5, 7, 8, 9, 11, 12, 14, 15)       # MobileNetV2 blocks with `x + project(...)` (stride 1, Cin == Cout)


def trained_like_weights(backbone, hp=None, seed=11, target_frac=0.05, calib_images=4):
    """A second seeded weight set with the CONDITIONING of a trained net.  ``synthetic_weights`` draws
    He-normal kernels and BatchNorm statistics that have nothing to do with the activations they normalise: every
    residual branch then has unit gain, a relative perturbation grows ~1.3x per block (x33 through MobileNetV2's 16
    blocks) and the bf16 mode can only be held to a multiple of that amplification.  Here (a) every BatchNorm's moving
    mean / variance ARE the statistics of its input over ``calib_images`` seeded images (one training-mode pass of the
    oracle's torch-CPU graph: what training leaves behind), so activations stay O(1) through the depth, and (b) the
    project BatchNorm of every residual block has gamma in [0.1, 0.3] (trained residual branches contribute small
    updates: gain < 1), and (c) the BatchNorms in front of a ReLU6 put most units into the activation's linear range
    (beta in [0.8, 1.6], gamma in [0.4, 0.7]): a random BN + ReLU stack is CHAOTIC -- every rectification followed by a
    mean removal grows a relative perturbation ~1.45x (measured on the calibrated net with beta ~ 0: 4e-3 behind Conv1 ->
    8e-2 behind block 16) -- which no trained detector is.  Heads are tamed / calibrated exactly as in ``synthetic_weights``.  VGG16 has no BatchNorm and no
    residuals: only the seed differs."""
    import torch
    from oracle import net_oracle as no
    from oracle import torch_cpu_graph as tg
    hp = hp or hyper_params(backbone)
    key = ("trained", backbone, seed, target_frac, hp["total_labels"], hp["img_size"], calib_images)
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
            blk = name.split("_")[1] if name.startswith("block_") and "_project_BN" in name else None
            if blk is not None and int(blk) in RESIDUAL_BLOCKS:
                w[name] = rng.uniform(0.1, 0.3, shape).astype(np.float32)
            elif "project_BN" in name:
                w[name] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
            else:       # BatchNorm in front of a ReLU6: most units in the linear range (see the docstring)
                w[name] = rng.uniform(0.4, 0.7, shape).astype(np.float32)
        elif var == "beta":
            if "project_BN" in name:
                w[name] = rng.uniform(-0.2, 0.2, shape).astype(np.float32)
            else:
                w[name] = rng.uniform(0.8, 1.6, shape).astype(np.float32)
        elif var == "moving_mean":
            w[name] = np.zeros(shape, np.float32)
        elif var == "moving_variance":
            w[name] = np.ones(shape, np.float32)
        elif var == "scale":
            w[name] = np.full(shape, 20.0, np.float32)
        else:  # bias
            w[name] = rng.uniform(-0.05, 0.05, shape).astype(np.float32)
    if any(n.endswith("moving_mean") for n in w):
        T = {n: torch.from_numpy(v) for n, v in w.items()}
        ids = {id(t): n for n, t in T.items()}
        seen = {}

        class CalibOps(tg.TorchOps):
            @staticmethod
            def batch_norm(x, gamma, beta, mean, var, eps=no.BN_EPS):
                mu = x.mean(dim=(0, 2, 3))
                va = ((x - mu.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))
                seen[ids[id(mean)]] = mu.numpy().copy()
                seen[ids[id(var)]] = np.maximum(va.numpy(), 1e-4).copy()
                xh = (x - mu.view(1, -1, 1, 1)) * torch.rsqrt(va + eps).view(1, -1, 1, 1)
                return xh * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)

        with torch.no_grad():
            no.forward(backbone, hp, T, images(calib_images, hp["img_size"], seed=1000 + seed), ops=CalibOps)
        for n, v in seen.items():
            w[n] = v.astype(np.float32)
    _calibrate_heads(w, backbone, hp, target_frac, label_gain=0.35)
    _WCACHE[key] = w
    return w


def _calibrate_heads(w, backbone, hp, target_frac, label_gain=0.5):
    """Tame the head logits, then shift the background bias of the label heads (bisection on the oracle's logits of one
    image) so that about ``target_frac`` of the anchors carry a non-background probability > 0.5."""
    from oracle import net_oracle as no
    L = hp["total_labels"]
    for i in range(1, 7):
        w["%d_conv_label_output/kernel" % i] *= np.float32(label_gain)
        w["%d_conv_boxes_output/kernel" % i] *= np.float32(0.25)
    acts = {}
    no.forward(backbone, hp, w, images(1, hp["img_size"], seed=0), acts)
    logits = acts["labels_head"][0].astype(np.float64)            # [N, L]

    def frac(t, g=1.0):
        lg = logits * g
        lg[:, 0] += t
        e = np.exp(lg - lg.max(-1, keepdims=True))
        p = e / e.sum(-1, keepdims=True)
        return float(((p.argmax(-1) != 0) & (p.max(-1) > 0.5)).mean())
    # the class spread of the random label heads must allow a single class to exceed 0.5 at all
    # (VGG16: logit std 0.55 -> never, whatever the background bias): scale label kernels AND biases
    # by g (the conv is linear: logits scale by g exactly) until, background removed, at least
    # 3 x target_frac of the anchors qualify.  MobileNetV2 keeps g = 1.
    g = 1.0
    while frac(-1e4, g) < 3 * target_frac and g < 64:
        g *= 2.0
    if g != 1.0:
        for i in range(1, 7):
            w["%d_conv_label_output/kernel" % i] *= np.float32(g)
            w["%d_conv_label_output/bias" % i] *= np.float32(g)
    lo, hi = -50.0 * g, 50.0 * g
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if frac(mid, g) > target_frac:
            lo = mid
        else:
            hi = mid
    t = np.float32(0.5 * (lo + hi))
    for i in range(1, 7):
        w["%d_conv_label_output/bias" % i][0::L] += t


# --------------------------------------------------------------------------------------------
# External known answers ([3P], written down FROM MEMORY of TensorFlow's own unit tests -- TF is
# not installable here, so they could not be re-executed; they do not come from this repo's
# oracle): tensorflow/core/kernels/non_max_suppression_op_test.cc (NonMaxSuppressionOpTest /
# CombinedNonMaxSuppressionOpTest) and python/ops/image_ops_test.py (NonMaxSuppressionTest).
# Each case: boxes [N,4], scores [N,C], max_per_class, max_total, iou_thr, score_thr, clip ->
# expected kept anchor indices (in output order), classes, valid.
_TF_THREE_CLUSTERS = [[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9],
                      [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]]
_TF_THREE_CLUSTERS_FLIPPED = [[1, 1, 0, 0], [0, 0.1, 1, 1.1], [0, .9, 1, -0.1],
                              [0, 10, 1, 11], [1, 10.1, 0, 11.1], [1, 101, 0, 100]]
_TF_SCORES6 = [.9, .75, .6, .95, .5, .3]
_TF_COMBINED_BOXES = [[0, 0, 0.1, 0.1], [0, 0.01, 0.1, 0.11], [0, -0.01, 0.1, 0.09],
                      [0, 0.11, 0.1, 0.2], [0, 0.12, 0.1, 0.21], [0, 0.3, 1, 0.4]]


def tf_nms_known_answers():
    f = np.float32
    col = lambda s: np.asarray(s, f).reshape(-1, 1)
    cases = [
        # NonMaxSuppressionOpTest.TestSelectFromThreeClusters -> [3, 0, 5]
        dict(name="three_clusters", boxes=_TF_THREE_CLUSTERS, scores=col(_TF_SCORES6), mpc=3, mt=3,
             iou=0.5, thr=float("-inf"), clip=False, idx=[3, 0, 5], cls=[0, 0, 0]),
        # ...FlippedCoordinates -> [3, 0, 5] (corners are re-ordered inside the IoU)
        dict(name="three_clusters_flipped", boxes=_TF_THREE_CLUSTERS_FLIPPED, scores=col(_TF_SCORES6), mpc=3,
             mt=3, iou=0.5, thr=float("-inf"), clip=False, idx=[3, 0, 5], cls=[0, 0, 0]),
        # TestSelectAtMostTwoBoxesFromThreeClusters -> [3, 0]
        dict(name="at_most_two", boxes=_TF_THREE_CLUSTERS, scores=col(_TF_SCORES6), mpc=2, mt=2, iou=0.5,
             thr=float("-inf"), clip=False, idx=[3, 0], cls=[0, 0]),
        # TestSelectWithNegativeScores (scores - 10) -> [3, 0, 5]
        dict(name="negative_scores", boxes=_TF_THREE_CLUSTERS, scores=col(_TF_SCORES6) - f(10), mpc=6, mt=6,
             iou=0.5, thr=float("-inf"), clip=False, idx=[3, 0, 5], cls=[0, 0, 0]),
        # TestSelectFromTenIdenticalBoxes -> [0]
        dict(name="ten_identical", boxes=[[0, 0, 1, 1]] * 10, scores=col([.9] * 10), mpc=3, mt=3, iou=0.5,
             thr=float("-inf"), clip=False, idx=[0], cls=[0]),
        # TestSelectSingleBox -> [0]
        dict(name="single_box", boxes=[[0, 0, 1, 1]], scores=col([.9]), mpc=3, mt=3, iou=0.5,
             thr=float("-inf"), clip=False, idx=[0], cls=[0]),
        # CombinedNonMaxSuppressionOpTest.TestSelectFromThreeClusters: boxes {0,.11,.1,.2},{0,0,.1,.1},
        # {0,.3,1,.4}; scores {.95,.9,.3}; valid 3
        dict(name="combined_three_clusters", boxes=_TF_COMBINED_BOXES, scores=col(_TF_SCORES6), mpc=3, mt=3,
             iou=0.5, thr=0.0, clip=True, idx=[3, 0, 5], cls=[0, 0, 0]),
        # ...WithScoreThreshold (0.4): third row zero padded, valid 2
        dict(name="combined_score_threshold", boxes=_TF_COMBINED_BOXES, scores=col(_TF_SCORES6), mpc=3, mt=3,
             iou=0.5, thr=0.4, clip=True, idx=[3, 0], cls=[0, 0]),
        # ...WithTwoClasses: boxes {0,.11,.1,.2},{0,0,.1,.1},{0,.01,.1,.11}; classes {0,1,0}
        dict(name="combined_two_classes", boxes=_TF_COMBINED_BOXES,
             scores=np.asarray([[.1, .9], [.75, .8], [.6, .3], [.95, .1], [.5, .5], [.3, .1]], f), mpc=3, mt=3,
             iou=0.5, thr=0.0, clip=True, idx=[3, 0, 1], cls=[0, 1, 0]),
    ]
    for c in cases:
        c["boxes"] = np.asarray(c["boxes"], f)
        c["scores"] = np.asarray(c["scores"], f)
    return cases
