"""GPU parity tests for the box family: HIP kernels (through the C ABI / ctypes surface)
vs the CPU oracle on the same seeded inputs and vs the committed golden fixtures.
Bar: bit-exact indices (kept anchors, match indices, labels, valid counts); priors
bit-exact; boxes/scores/deltas within 1e-4 abs (tolerance of BASELINE.json north_star;
observed ~1e-7, the only non-exact ops being expf/logf last-ulp)."""
import os

import numpy as np
import pytest
import torch

import helpers
from oracle import bbox_oracle as bo
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def bbox_utils():
    from utils import bbox_utils as m
    return m


@pytest.mark.parametrize("backbone", ["mobilenet_v2", "vgg16"])
def test_priors_bit_exact(bbox_utils, backbone):
    p = _np(bbox_utils.generate_prior_boxes(helpers.FMAPS[backbone], helpers.ASPECT_RATIOS))
    gold = np.load(os.path.join(GOLD, "priors.npz"))[backbone]
    np.testing.assert_array_equal(p, gold)


def test_priors_other_configs(bbox_utils):
    for fm in ([32, 16, 8, 4, 2, 1], [64, 32, 16, 8, 6, 4], [3, 1]):
        ars = helpers.ASPECT_RATIOS[:len(fm)]
        np.testing.assert_array_equal(_np(bbox_utils.generate_prior_boxes(fm, ars)),
                                      bo.generate_prior_boxes(fm, ars))
    with pytest.raises(ZeroDivisionError):      # same failure as the reference's A1 with m == 1
        bbox_utils.generate_prior_boxes([7], helpers.ASPECT_RATIOS[:1])
    b = bbox_utils.generate_base_prior_boxes([1., 2., 0.5], 1, 6)
    np.testing.assert_array_equal(_np(b), bo.generate_base_prior_boxes([1., 2., 0.5], 1, 6))


def test_decode_boxes(bbox_utils):
    p = np.load(os.path.join(GOLD, "priors.npz"))["vgg16"]
    d, _ = helpers.decoder_inputs(3, p.shape[0], seed=4)
    out = _np(bbox_utils.get_bboxes_from_deltas(p, d))
    ref = co.decode(p, d, [1, 1, 1, 1])
    # exp() of N(0,1) deltas: relative tolerance on the (unclipped) boxes
    np.testing.assert_allclose(out, ref, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["rand", "none", "ties", "degenerate"])
def test_decoder_golden(name):
    from models.decoder import SSDDecoder
    z = np.load(os.path.join(GOLD, "decode_nms.npz"))
    p = np.load(os.path.join(GOLD, "priors.npz"))["mobilenet_v2"]
    dec = SSDDecoder(p, helpers.VARIANCES)
    b, l, s = dec.call([z[name + "_deltas"], z[name + "_probs"]], return_indices=True)
    np.testing.assert_array_equal(_np(dec.last_kept_indices), z[name + "_idx"])
    np.testing.assert_array_equal(_np(dec.last_valid_detections), z[name + "_valid"])
    np.testing.assert_array_equal(_np(l), z[name + "_labels"])
    np.testing.assert_array_equal(_np(s), z[name + "_scores"])
    np.testing.assert_allclose(_np(b), z[name + "_boxes"], atol=TOL, rtol=0)
    assert np.abs(_np(b) - z[name + "_boxes"]).max() < 1e-6


@pytest.mark.parametrize("B,N,L,seed,frac", [(4, 2268, 21, 31, 0.10), (2, 8732, 21, 32, 0.10),
                                             (2, 24564, 21, 33, 0.10), (3, 2268, 21, 34, 0.9),
                                             (1, 100, 3, 35, 0.5), (2, 1, 21, 36, 1.0), (5, 333, 91, 37, 0.2)])
def test_decoder_vs_c_oracle(B, N, L, seed, frac):
    """Sizes up to the 24 564-anchor stress (BASELINE config 5) against the plain-C oracle."""
    from models.decoder import SSDDecoder
    rng = np.random.default_rng(seed)
    c = rng.uniform(0.05, 0.95, (N, 2)); sz = rng.uniform(0.02, 0.4, (N, 2))
    p = np.clip(np.concatenate([c - sz / 2, c + sz / 2], -1), 0, 1).astype(np.float32)
    d, pr = helpers.decoder_inputs(B, N, L, seed=seed, boost_frac=frac)
    dec = SSDDecoder(p, helpers.VARIANCES)
    b, l, s = dec.call([d, pr], return_indices=True)
    rb, rl, rs, rv, ri = co.decode_nms(d, pr, p, helpers.VARIANCES)
    np.testing.assert_array_equal(_np(dec.last_valid_detections), rv)
    np.testing.assert_array_equal(_np(dec.last_kept_indices), ri)
    np.testing.assert_array_equal(_np(l), rl)
    np.testing.assert_array_equal(_np(s), rs)
    np.testing.assert_allclose(_np(b), rb, atol=TOL, rtol=0)


def test_decoder_many_candidates_per_class():
    """> 4096 candidates in one class exercises the in-HBM sort fallback; > 256 exercises
    multi-chunk suppression; max_total smaller than the survivors exercises truncation."""
    from models.decoder import SSDDecoder
    N, L = 9000, 4
    rng = np.random.default_rng(41)
    c = rng.uniform(0.0, 1.0, (N, 2)); sz = rng.uniform(0.01, 0.05, (N, 2))
    p = np.clip(np.concatenate([c - sz / 2, c + sz / 2], -1), 0, 1).astype(np.float32)
    d = (rng.standard_normal((1, N, 4)) * 0.1).astype(np.float32)
    pr = np.zeros((1, N, L), np.float32)
    pr[0, :, 2] = rng.uniform(0.5001, 0.999, N).astype(np.float32)
    pr[0, :, 0] = 1 - pr[0, :, 2]
    pr[0, ::2, 1] = 0.0
    for T in (200, 50, 1000):
        dec = SSDDecoder(p, helpers.VARIANCES, max_total_size=T)
        b, l, s = dec.call([d, pr], return_indices=True)
        rb, rl, rs, rv, ri = co.decode_nms(d, pr, p, helpers.VARIANCES, max_per_class=T, max_total=T)
        np.testing.assert_array_equal(_np(dec.last_kept_indices), ri)
        np.testing.assert_array_equal(_np(dec.last_valid_detections), rv)
        np.testing.assert_array_equal(_np(s), rs)
        np.testing.assert_allclose(_np(b), rb, atol=TOL, rtol=0)


def test_decoder_empty_and_errors():
    from models.decoder import SSDDecoder
    p = np.load(os.path.join(GOLD, "priors.npz"))["mobilenet_v2"]
    dec = SSDDecoder(p, helpers.VARIANCES)
    b, l, s = dec([np.zeros((0, 2268, 4), np.float32), np.zeros((0, 2268, 21), np.float32)])
    assert b.shape == (0, 200, 4) and l.shape == (0, 200) and s.shape == (0, 200)
    with pytest.raises(ValueError):
        dec([np.zeros((1, 10, 4), np.float32), np.zeros((1, 10, 21), np.float32)])
    cfg = dec.get_config()
    assert cfg["max_total_size"] == 200 and cfg["score_threshold"] == 0.5 and cfg["prior_boxes"].shape == (2268, 4)


def test_combined_nms_raw(bbox_utils):
    """bbox_utils.non_max_suppression on raw boxes incl. inverted / zero-area boxes, negative
    threshold and column 0 as an ordinary class."""
    rng = np.random.default_rng(51)
    B, N, C = 3, 500, 5
    c = rng.uniform(0.0, 1.0, (B, N, 2)); sz = rng.uniform(-0.1, 0.4, (B, N, 2))
    boxes = np.concatenate([c - sz / 2, c + sz / 2], -1).astype(np.float32)   # some inverted
    boxes[:, ::17, 2:] = boxes[:, ::17, :2]                                     # zero area
    scores = rng.uniform(0, 1, (B, N, C)).astype(np.float32)
    for thr, mpc, mt in ((0.5, 200, 200), (-1.0, 30, 40), (0.9, 5, 200)):
        ob, osc, oc, v = bbox_utils.non_max_suppression(boxes[:, :, None, :], scores,
                                                        max_output_size_per_class=mpc, max_total_size=mt,
                                                        score_threshold=thr)
        rb, rs, rc, rv = bo.combined_non_max_suppression(boxes, scores, mpc, mt, 0.5, thr)
        np.testing.assert_array_equal(_np(v), rv)
        np.testing.assert_array_equal(_np(oc), rc)
        np.testing.assert_array_equal(_np(osc), rs)
        np.testing.assert_array_equal(_np(ob), rb)
    with pytest.raises(TypeError):
        bbox_utils.non_max_suppression(boxes[:, :, None, :], scores)


def test_tf_published_nms_known_answers_gpu(bbox_utils):
    """TensorFlow's own NMS unit-test vectors ([3P], from memory: helpers.tf_nms_known_answers)
    through the product's ssd_combined_nms -- known answers that do not come from this repo."""
    import ssd_hip as h
    lib = h.lib()
    for c in helpers.tf_nms_known_answers():
        N, C = c["scores"].shape
        T = c["mt"]
        bd, sd = h.to_dev(c["boxes"][None]), h.to_dev(c["scores"][None])
        ob = torch.empty((1, T, 4), dtype=torch.float32, device=bd.device)
        os_ = torch.empty((1, T), dtype=torch.float32, device=bd.device)
        oc = torch.empty((1, T), dtype=torch.float32, device=bd.device)
        ov = torch.empty((1,), dtype=torch.int32, device=bd.device)
        oi = torch.empty((1, T), dtype=torch.int32, device=bd.device)
        ws = h.workspace(lib.ssd_decode_nms_workspace_bytes(1, N, C, c["mpc"]))
        thr = c["thr"] if np.isfinite(c["thr"]) else -3.0e38
        h.check(lib.ssd_combined_nms(h.ptr(bd), h.ptr(sd), 1, N, C, c["mpc"], T, c["iou"], thr, int(c["clip"]),
                                     h.ptr(ob), h.ptr(os_), h.ptr(oc), h.ptr(ov), h.ptr(oi), h.ptr(ws), ws.numel(),
                                     h.stream()), c["name"])
        n = len(c["idx"])
        assert int(ov[0]) == n, c["name"]
        assert _np(oi)[0, :n].tolist() == c["idx"], c["name"]
        assert _np(oi)[0, n:].tolist() == [-1] * (T - n), c["name"]
        assert _np(oc)[0, :n].tolist() == c["cls"], c["name"]
        exp_b = c["boxes"][c["idx"]]
        if c["clip"]:
            exp_b = np.clip(exp_b, 0, 1)
        np.testing.assert_array_equal(_np(ob)[0, :n], exp_b)
        np.testing.assert_array_equal(_np(os_)[0, :n], c["scores"][c["idx"], c["cls"]])
        assert not _np(ob)[0, n:].any() and not _np(os_)[0, n:].any()


def test_iou_map_and_match(bbox_utils):
    from utils import train_utils
    z = np.load(os.path.join(GOLD, "match.npz"))
    p = np.load(os.path.join(GOLD, "priors.npz"))["mobilenet_v2"]
    np.testing.assert_array_equal(_np(bbox_utils.generate_iou_map(p, z["gt"])), z["iou"])
    hp = helpers.hyper_params()
    d, oh, lab, mi = train_utils.calculate_actual_outputs(p, z["gt"], z["gl"], hp, return_indices=True)
    np.testing.assert_array_equal(_np(lab), z["label_idx"])
    np.testing.assert_array_equal(_np(mi), z["match_idx"])
    np.testing.assert_allclose(_np(d), z["deltas"], atol=TOL, rtol=0)
    assert np.abs(_np(d) - z["deltas"]).max() < 1e-5
    oh = _np(oh)
    assert (oh.argmax(-1) == z["label_idx"]).all() and (oh.sum(-1) == 1).all()
    # batched boxes x gt (eval_utils usage) and 2-d x 2-d
    bb = np.broadcast_to(p[None, :200], (4, 200, 4)).copy()
    np.testing.assert_array_equal(_np(bbox_utils.generate_iou_map(bb, z["gt"])), bo.generate_iou_map(bb, z["gt"]))
    np.testing.assert_array_equal(_np(bbox_utils.generate_iou_map(p[:50], z["gt"][0])),
                                  bo.generate_iou_map(p[:50], z["gt"][0]))
    # NaN for 0/0 like the reference
    assert torch.isnan(bbox_utils.generate_iou_map(np.zeros((1, 4), np.float32), np.zeros((1, 1, 4), np.float32))).all()


def test_match_full_size_vs_c(bbox_utils):
    """BASELINE config 4 shape: 32 images/GPU, VGG priors, G=16."""
    from utils import train_utils
    p = np.load(os.path.join(GOLD, "priors.npz"))["vgg16"]
    gt, gl = helpers.gt_inputs(32, seed=61)
    hp = helpers.hyper_params("vgg16")
    d, oh, lab, mi = train_utils.calculate_actual_outputs(p, gt, gl, hp, return_indices=True)
    rd, rl, rm = co.match_encode(p, gt, gl, helpers.VARIANCES)
    np.testing.assert_array_equal(_np(lab), rl)
    np.testing.assert_array_equal(_np(mi), rm)
    np.testing.assert_allclose(_np(d), rd, atol=TOL, rtol=0)


def test_encode_decode_roundtrip_gpu(bbox_utils):
    p = np.load(os.path.join(GOLD, "priors.npz"))["mobilenet_v2"]
    d, _ = helpers.decoder_inputs(2, p.shape[0], seed=71)
    d *= 0.3
    boxes = bbox_utils.get_bboxes_from_deltas(p, d)
    back = _np(bbox_utils.get_deltas_from_bboxes(p, boxes))
    np.testing.assert_allclose(back, d, atol=2e-4)
    np.testing.assert_allclose(back, bo.get_deltas_from_bboxes(p, _np(boxes)), atol=1e-5)


def test_eval_update_stats_vs_oracle():
    """N2: product utils.eval_utils.update_stats (GPU IoU + host bookkeeping) vs the loop-for-loop
    oracle restatement of reference utils/eval_utils.py:19-54 on synthetic detections: jittered
    copies of the ground truth (several per box: double matches), wrong labels, background
    boxes, label-0 zero padding rows, -1-padded ground truth; then the full mAP."""
    from utils import eval_utils as eu
    from oracle import eval_oracle as eo
    rng = np.random.default_rng(17)
    B, G, T, L = 12, 8, 40, 6
    gt, gl = helpers.gt_inputs(B, G=G, L=L, seed=5)
    pb = np.zeros((B, T, 4), np.float32); pl = np.zeros((B, T), np.float32); ps = np.zeros((B, T), np.float32)
    for b in range(B):
        g = int((gl[b] > 0).sum())
        n = int(rng.integers(T // 2, T))
        for t in range(n):
            if rng.random() < 0.7:
                j = int(rng.integers(0, g))
                pb[b, t] = np.clip(gt[b, j] + rng.normal(0, 0.03, 4), 0, 1)
                pl[b, t] = gl[b, j] if rng.random() < 0.8 else rng.integers(1, L)
            else:
                c = rng.uniform(0.1, 0.9, 2); s = rng.uniform(0.05, 0.3, 2)
                pb[b, t] = np.clip(np.concatenate([c - s / 2, c + s / 2]), 0, 1)
                pl[b, t] = rng.integers(1, L)
            ps[b, t] = rng.uniform(0.5, 1.0)
        order = np.argsort(-ps[b, :n])
        pb[b, :n], pl[b, :n], ps[b, :n] = pb[b, order], pl[b, order], ps[b, order]
    labels = ["bg"] + ["c%d" % i for i in range(1, L)]
    got = eu.update_stats(pb, pl, ps, gt, gl, eu.init_stats(labels))
    ref = eo.update_stats(pb, pl, ps, gt, gl, eo.init_stats(labels))
    for cid in ref:
        assert got[cid]["total"] == ref[cid]["total"]
        assert got[cid]["tp"] == ref[cid]["tp"] and got[cid]["fp"] == ref[cid]["fp"], cid
        np.testing.assert_array_equal(np.asarray(got[cid]["scores"], np.float32), np.asarray(ref[cid]["scores"], np.float32))
    assert sum(sum(r["tp"]) for r in ref.values()) > 20 and sum(sum(r["fp"]) for r in ref.values()) > 20
    got, gm = eu.calculate_mAP(got)
    ref, rm = eo.calculate_mAP(ref)
    assert float(gm) == float(rm) and 0.05 < float(gm) < 1.0
    for cid in ref:
        assert got[cid]["AP"] == ref[cid]["AP"]


@pytest.mark.parametrize("H,W,S", [(375, 500, 300), (500, 333, 300), (300, 300, 300), (120, 97, 512), (1, 1, 8)])
def test_preprocess_gpu_vs_oracle(H, W, S):
    """N4: ssd_preprocess (uint8 -> float32 [0,1], bilinear resize, TF2 half-pixel centres) bit-exact
    against the NumPy restatement, single images (VOC-like sizes) and a batch."""
    from utils import data_utils
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    out, gb, gl = data_utils.preprocessing({"image": img, "objects": {"bbox": np.zeros((2, 4), np.float32),
                                                                       "label": np.array([0, 19]),
                                                                       "is_difficult": np.array([False, True])}},
                                           S, S, evaluate=True)
    np.testing.assert_array_equal(_np(out), bo.preprocess_image(img, S, S))
    assert gl.tolist() == [1] and gb.shape == (1, 4)
    batch = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    np.testing.assert_array_equal(_np(data_utils.preprocess_batch(batch, S, S)), bo.preprocess_image(batch, S, S))
    with pytest.raises(ValueError):
        data_utils.preprocess_batch(batch.astype(np.float32), S, S)
