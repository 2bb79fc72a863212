#!/usr/bin/env python
"""Regenerate parity fixtures FROM THE REAL REFERENCE (FurkanOM/tf-ssd executed on TensorFlow).

The build container and the GPU boxes have no TensorFlow, so everything that runs inside TF -- conv / BN
numerics, box decode, `tf.image.combined_non_max_suppression`, target assignment, the loss -- is pinned in
this repository only against its own CPU restatement (oracle/).  This script is the other route: run it ONCE
on any machine that has TensorFlow 2.x and a checkout of the reference, commit the files it writes, and
tests/test_tf_golden.py then holds the oracle AND the HIP path to the reference's own outputs (those tests
SKIP -- not pass -- while the files are absent).

    python tools/make_tf_golden.py --reference /path/to/tf-ssd [--out tests/golden] [--skip-nets]

It imports the reference's modules (nothing of them is copied), feeds them the seeded inputs the repo's tests
already use (tests/golden/decode_nms.npz / match.npz inputs, tests/helpers.py generators) and writes

    tf_priors.npz       generate_prior_boxes for both backbones            (utils/bbox_utils.py:149-176)
    tf_decode_nms.npz   SSDDecoder.call on the four decoder cases           (models/decoder.py:36-55)
    tf_match.npz        calculate_actual_outputs + generate_iou_map         (utils/train_utils.py:90-127)
    tf_loss.npz         CustomLoss.loc_loss_fn / conf_loss_fn               (ssd_loss.py:8-65)
    tf_published.npz    Keras docs' Huber / CategoricalCrossentropy examples, tf.image.resize vectors
    tf_net_<backbone>.npz   get_model(...) forward on 2 seeded images with the seeded synthetic weights
                            (models/ssd_mobilenet_v2.py:7-35, models/ssd_vgg16.py:33-97) -- `--skip-nets` omits

Only data (inputs + outputs) is written; every file records the TensorFlow version that produced it.
"""
import argparse
import functools
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of FurkanOM/tf-ssd")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--skip-nets", action="store_true")
    args = ap.parse_args()

    import tensorflow as tf
    # the repo's seeded generators (numpy only) -- NOT tf-ssd_amd/ (its `utils` / `models` packages would shadow
    # the reference's)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import helpers
    sys.path.insert(0, os.path.abspath(args.reference))
    from utils import bbox_utils, train_utils          # the REFERENCE's modules
    from models.decoder import SSDDecoder
    import ssd_loss
    assert os.path.abspath(bbox_utils.__file__).startswith(os.path.abspath(args.reference)), "reference not first on sys.path"
    ver = np.array(tf.__version__)
    os.makedirs(args.out, exist_ok=True)
    f32 = lambda t: np.asarray(t.numpy() if hasattr(t, "numpy") else t, np.float32)

    # ---- prior boxes
    pri = {}
    for bb in ("mobilenet_v2", "vgg16"):
        hp = dict(train_utils.get_hyper_params(bb))
        pri[bb] = f32(bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"]))
    np.savez_compressed(os.path.join(args.out, "tf_priors.npz"), tf_version=ver, **pri)

    # ---- SSDDecoder on the repo's decoder cases (inputs read from the committed fixture)
    z = np.load(os.path.join(REPO, "tests", "golden", "decode_nms.npz"))
    dec = SSDDecoder(tf.constant(pri["mobilenet_v2"]), helpers.VARIANCES)
    out = {}
    for name in ("rand", "none", "ties", "degenerate"):
        b, l, s = dec([tf.constant(z[name + "_deltas"]), tf.constant(z[name + "_probs"])])
        out.update({name + "_boxes": f32(b), name + "_labels": f32(l), name + "_scores": f32(s)})
    np.savez_compressed(os.path.join(args.out, "tf_decode_nms.npz"), tf_version=ver, **out)

    # ---- target assignment
    m = np.load(os.path.join(REPO, "tests", "golden", "match.npz"))
    hp = dict(train_utils.get_hyper_params("mobilenet_v2"))
    hp["total_labels"] = 21
    deltas, labels = train_utils.calculate_actual_outputs(tf.constant(pri["mobilenet_v2"]), tf.constant(m["gt"]),
                                                          tf.constant(m["gl"]), hp)
    iou = bbox_utils.generate_iou_map(tf.constant(pri["mobilenet_v2"]), tf.constant(m["gt"]))
    np.savez_compressed(os.path.join(args.out, "tf_match.npz"), tf_version=ver, deltas=f32(deltas), labels_one_hot=f32(labels),
                        iou=f32(iou))

    # ---- loss (the generator of tests/test_loss.py::_case, restated here with the same seeds)
    def loss_case(B, N, L, seed, pos_frac=0.02, sharp=3.0):
        rng = np.random.default_rng(seed)
        yl = np.zeros((B, N, L), np.float32)
        yd = np.zeros((B, N, 4), np.float32)
        for b in range(B):
            npos = int(rng.integers(0, max(2, int(pos_frac * N) * 2)))
            if b == 0:
                npos = 0
            idx = rng.choice(N, npos, replace=False)
            yl[b, :, 0] = 1
            yl[b, idx, 0] = 0
            yl[b, idx, rng.integers(1, L, npos)] = 1
            yd[b, idx] = rng.standard_normal((npos, 4)).astype(np.float32) * 2
        pd = (rng.standard_normal((B, N, 4)) * 1.5).astype(np.float32)
        zz = (rng.standard_normal((B, N, L)) * sharp).astype(np.float32)
        zz[..., 0] += 2.0
        e = np.exp(zz - zz.max(-1, keepdims=True))
        return yd, yl, pd, (e / e.sum(-1, keepdims=True)).astype(np.float32)
    cl = ssd_loss.CustomLoss(3, 1)
    out = {}
    for B, N, L, seed in ((4, 2268, 21, 11), (2, 8732, 21, 12), (2, 50, 3, 14)):
        yd, yl, pd, pp = loss_case(B, N, L, seed)
        key = "b%d_n%d_l%d_s%d" % (B, N, L, seed)
        out[key + "_loc"] = f32(cl.loc_loss_fn(tf.constant(yd), tf.constant(pd)))
        out[key + "_conf"] = f32(cl.conf_loss_fn(tf.constant(yl), tf.constant(pp)))
    np.savez_compressed(os.path.join(args.out, "tf_loss.npz"), tf_version=ver, **out)

    # ---- published examples re-executed (Keras API docs) + resize vectors
    h = tf.keras.losses.Huber()
    cce = tf.keras.losses.CategoricalCrossentropy()
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    rs = tf.image.resize(tf.image.convert_image_dtype(tf.constant(img), tf.float32), (30, 30))
    np.savez_compressed(os.path.join(args.out, "tf_published.npz"), tf_version=ver,
                        huber=f32(h([[0, 1], [0, 0]], [[0.6, 0.4], [0.4, 0.6]])),
                        cce=f32(cce([[0, 1, 0], [0, 0, 1]], [[0.05, 0.95, 0], [0.1, 0.8, 0.1]])),
                        resize_in=img, resize_out=f32(rs))

    # ---- whole networks with the repo's seeded synthetic weights
    if not args.skip_nets:
        import tensorflow.keras.applications as apps
        import tensorflow.keras.applications.mobilenet_v2 as mbv2_mod
        # the reference builds MobileNetV2 with the default weights="imagenet" (a download): random init instead,
        # every variable is overwritten below anyway
        mbv2_mod.MobileNetV2 = functools.partial(mbv2_mod.MobileNetV2, weights=None)
        apps.MobileNetV2 = mbv2_mod.MobileNetV2
        for bb in ("mobilenet_v2", "vgg16"):
            mod = __import__("models.ssd_" + bb, fromlist=["get_model"])
            if bb == "mobilenet_v2":
                mod.MobileNetV2 = mbv2_mod.MobileNetV2
            hp = dict(train_utils.get_hyper_params(bb))
            hp["total_labels"] = 21
            model = mod.get_model(hp)
            w = helpers.synthetic_weights(bb, helpers.hyper_params(bb))
            seen = set()
            for v in model.weights:
                name = v.name.split(":")[0]
                layer, var = name.rsplit("/", 1)
                key = layer + "/" + ("scale" if var.startswith("Variable") else var)
                assert key in w, "no synthetic value for " + name
                assert tuple(v.shape) == w[key].shape, (name, v.shape, w[key].shape)
                v.assign(w[key])
                seen.add(key)
            assert seen == set(w), sorted(set(w) - seen)[:5]
            x = helpers.images(2, 300, seed=0)
            d, p = model(tf.constant(x), training=False)
            np.savez_compressed(os.path.join(args.out, "tf_net_%s.npz" % bb), tf_version=ver, deltas=f32(d), probs=f32(p))
    print("wrote tf_*.npz (TensorFlow %s) to %s" % (tf.__version__, args.out))


if __name__ == "__main__":
    main()
