"""Parity at BASELINE.json's full sizes (configs[1]: SSD300-MobileNetV2 batch 64; configs[2]:
SSD300-VGG16 batch 32; configs[4] per-GPU shard: the MobileNetV2 graph at 512x512, batch 16, and
the 24 564-anchor decoder at batch 16).  The numpy oracle cannot run 64 images in seconds, so the full batch is
checked through (a) the oracle on a subset of its images -- every image of a batch is computed
by the same kernels/tiles, and an image's result does not depend on its batch neighbours --
and (b) size-independent properties of the outputs: bitwise determinism, batch-composition
independence, softmax rows, and the CombinedNMS contract (sorted scores, clipped boxes,
labels in range, zero padding, idempotence: NMS of the survivors keeps all of them)."""
import os

import numpy as np
import pytest
import torch

import helpers
from oracle import c_oracle as co
from oracle import net_oracle as no
from oracle import torch_cpu_graph as tg

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _max_same_class_iou(boxes, labels):
    worst = 0.0
    for c in np.unique(labels):
        bb = boxes[labels == c].astype(np.float64)
        if len(bb) < 2:
            continue
        y1, x1, y2, x2 = bb[:, 0], bb[:, 1], bb[:, 2], bb[:, 3]
        area = (y2 - y1) * (x2 - x1)
        iy = np.clip(np.minimum(y2[:, None], y2[None]) - np.maximum(y1[:, None], y1[None]), 0, None)
        ix = np.clip(np.minimum(x2[:, None], x2[None]) - np.maximum(x1[:, None], x1[None]), 0, None)
        inter = iy * ix
        iou = inter / np.maximum(area[:, None] + area[None] - inter, 1e-30)
        np.fill_diagonal(iou, 0)
        worst = max(worst, float(iou.max()))
    return worst


def _assert_deltas(d, rd, variances):
    """pred_deltas are unbounded regression outputs (|d| up to ~4 with the seeded weights) that
    enter the boxes only as d * variances (models/decoder.py:41, variances 0.1 / 0.2): the
    contract's 1e-4 ABSOLUTE bar is asserted on that product, and on the raw values a bound
    relative to the tensor's range (fp32 accumulation order over ~50 layers; the two CPU
    restatements differ from each other by as much)."""
    err = np.abs(d - rd)
    assert (err * np.asarray(variances, np.float32)).max() <= 1e-4, "variance-scaled deltas: %.3e" % (
        (err * np.asarray(variances, np.float32)).max())
    assert err.max() <= 5e-5 * max(2.0, float(np.abs(rd).max())), "deltas: max abs err %.3e (max |ref| %.3g)" % (
        err.max(), np.abs(rd).max())


def _iou1(a, bs):
    ay1, ax1, ay2, ax2 = a
    iy = np.clip(np.minimum(ay2, bs[:, 2]) - np.maximum(ay1, bs[:, 0]), 0, None)
    ix = np.clip(np.minimum(ax2, bs[:, 3]) - np.maximum(ax1, bs[:, 1]), 0, None)
    inter = iy * ix
    return inter / np.maximum((ay2 - ay1) * (ax2 - ax1) + (bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1]) - inter, 1e-30)


def _assert_same_detections(b, l, s, rb, rl, rs, v, what):
    """Row-for-row equality (labels identical, scores / boxes within the contract's 1e-4 abs).
    The selection itself is a chain of hard decisions on ~50-layer fp32 sums, so two correct
    implementations may differ where a decision is BORDERLINE: two scores within fp32 noise of
    each other (rank swap), a score within noise of the 0.5 threshold, or an IoU within noise of
    the 0.5 suppression threshold.  On a row mismatch every oracle row must therefore have its own
    partner among the product's rows (same label, score and box within 1e-4) or be explained by
    such a borderline decision -- and vice versa.  Returns the number of detections that needed the borderline
    explanation (0 = the image matched row for row); the caller bounds and reports how many IMAGES did."""
    vp = int((s > 0).sum())
    if vp == v and np.array_equal(l, rl) and np.abs(s - rs).max() <= 1e-4 and np.abs(b - rb).max() <= 1e-4:
        return 0
    used = np.zeros(vp, bool)
    unmatched_ref = []
    for j in range(v):
        ok = (~used) & (l[:vp] == rl[j]) & (np.abs(s[:vp] - rs[j]) <= 1e-4) & (np.abs(b[:vp] - rb[j]).max(-1) <= 1e-4)
        k = np.nonzero(ok)[0]
        if k.size == 0:
            unmatched_ref.append(j)
            continue
        used[k[np.argmin(np.abs(k - j))]] = True
    extra = [(b[k], l[k], s[k], rb[:v], rl[:v]) for k in np.nonzero(~used)[0]]
    extra += [(rb[j], rl[j], rs[j], b[:vp], l[:vp]) for j in unmatched_ref]
    assert len(extra) <= 3, "%s: %d unmatched detections" % (what, len(extra))
    for box, lab, sc, other_b, other_l in extra:
        near_thr = abs(float(sc) - 0.5) <= 2e-4
        same = other_b[other_l == lab]
        near_iou = same.size > 0 and bool((np.abs(_iou1(box, same) - 0.5) <= 2e-3).any())
        cut_off = vp == 200 or v == 200                 # the top-200 truncation moved by one rank
        assert near_thr or near_iou or cut_off, "%s: unexplained detection (label %g score %.6f)" % (what, lab, sc)
    return len(extra)


def _check_nms_contract(boxes, labels, scores, L, max_total=200, score_thr=0.5):
    B = boxes.shape[0]
    assert boxes.shape == (B, max_total, 4) and labels.shape == (B, max_total) and scores.shape == (B, max_total)
    valid = (scores > 0).sum(1)
    for b in range(B):
        v = int(valid[b])
        s = scores[b]
        assert (s[:v] > score_thr).all() and (s[v:] == 0).all()          # strict threshold, zero padding
        assert (np.diff(s[:v]) <= 0).all()                                # sorted by score
        assert (boxes[b, v:] == 0).all() and (labels[b, v:] == 0).all()
        assert boxes[b].min() >= 0.0 and boxes[b].max() <= 1.0            # clip_boxes=True
        lab = labels[b, :v]
        assert ((lab >= 1) & (lab <= L - 1)).all() and (lab == np.round(lab)).all()
    return valid


def _hp(backbone, S):
    hp = helpers.hyper_params(backbone)
    if S != 300:
        hp["img_size"] = S
        hp["feature_map_shapes"] = [32, 16, 8, 4, 2, 1]      # MobileNetV2 graph at 512^2: N = 6132
    return hp


@pytest.mark.parametrize("backbone,B,S,subset,subset8", [
    ("mobilenet_v2", 64, 300, (0, 31, 63), (0, 9, 18, 27, 36, 45, 54, 63)),     # C2
    ("vgg16", 32, 300, (0,), (0, 4, 8, 12, 16, 20, 24, 31)),                     # C3
    ("mobilenet_v2", 16, 512, (0, 15), (0, 2, 4, 6, 8, 10, 12, 15)),             # C5 per-GPU shard (fp32)
])
def test_full_batch_forward_and_decode(backbone, B, S, subset, subset8):
    from models.decoder import get_decoder_model
    from utils import bbox_utils
    if backbone == "mobilenet_v2":
        from models.ssd_mobilenet_v2 import get_model
    else:
        from models.ssd_vgg16 import get_model
    hp = _hp(backbone, S)
    w = helpers.synthetic_weights(backbone, hp)
    m = get_model(hp, max_batch=B)
    m.set_weights(w)
    x = helpers.images(B, S, seed=0)          # image 0 is the one the synthetic head bias was calibrated on
    d, p = m(x)
    d, p = _np(d), _np(p)
    N, L = m.num_priors, hp["total_labels"]
    assert d.shape == (B, N, 4) and p.shape == (B, N, L)
    assert np.isfinite(d).all() and np.isfinite(p).all()
    np.testing.assert_allclose(p.sum(-1), 1.0, atol=1e-5)
    # (a) NumPy oracle on a subset of the batch: the contract's 1e-4 ABSOLUTE bar on both outputs
    xs = x[list(subset)]
    rd, rp = no.forward(backbone, hp, w, xs)
    assert np.abs(p[list(subset)] - rp).max() <= 1e-4
    _assert_deltas(d[list(subset)], rd, hp["variances"])
    # (a') END TO END on 8 images spread over the batch: the independent torch-CPU restatement of
    # the graph + the plain-C decode/NMS oracle vs the product's one-call predict
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    dm = get_decoder_model(m, priors, hp)
    boxes, labels, scores = [_np(t) for t in dm(x)]
    import torch as _t
    _t.set_num_threads(min(16, _t.get_num_threads()))
    sel = list(subset8)
    td, tp = tg.forward(backbone, hp, w, x[sel])
    assert np.abs(p[sel] - tp).max() <= 1e-4
    _assert_deltas(d[sel], td, hp["variances"])
    tb, tl, ts, tv, ti = co.decode_nms(td, tp, _np(priors), hp["variances"])
    assert tv.min() > 0, "synthetic calibration must leave NMS something to do on every image"
    borderline = {}
    for j, b in enumerate(sel):
        v = int(tv[j])
        n = _assert_same_detections(boxes[b], labels[b], scores[b], tb[j], tl[j], ts[j], v, "image %d" % b)
        if n:
            borderline[b] = n
    # how often the borderline branch was needed is part of the record (pytest -rA / -s shows it): with the shipped
    # tables every configuration matched ROW FOR ROW on all 8 images when this bound was written, so a regression
    # from 0 is visible in the log and more than 2 of the 8 images needing it fails
    print("end-to-end detections %s B=%d S=%d: %d of %d oracle images needed the borderline branch %s" % (
        backbone, B, S, len(borderline), len(sel), borderline))
    assert len(borderline) <= 2, "borderline explanations on %d of %d images: %s" % (len(borderline), len(sel), borderline)
    # (b) determinism and batch-composition independence (same tiles: same bits)
    d2, p2 = m(x)
    np.testing.assert_array_equal(_np(d2), d)
    np.testing.assert_array_equal(_np(p2), p)
    perm = np.random.default_rng(0).permutation(B)
    d3, p3 = m(x[perm])
    np.testing.assert_array_equal(_np(d3), d[perm])
    np.testing.assert_array_equal(_np(p3), p[perm])
    # decode + CombinedNMS of the whole batch: contract + the plain-C oracle on the same head outputs
    valid = _check_nms_contract(boxes, labels, scores, L)
    assert valid.min() > 0 and valid.mean() > 1
    rb, rl, rs, rv, ri = co.decode_nms(d, p, _np(priors), hp["variances"])
    np.testing.assert_array_equal(valid, rv)
    np.testing.assert_array_equal(labels, rl)          # bit-exact selection on identical inputs
    np.testing.assert_array_equal(scores, rs)
    np.testing.assert_allclose(boxes, rb, atol=1e-6, rtol=0)     # expf: device vs host libm, 1 ulp
    # greedy-NMS invariants on the UNCLIPPED survivors of the busiest image (the decoder clips its
    # outputs after the selection, so they are checked through the raw op with clip_boxes=False):
    # survivors of one class overlap by at most the IoU threshold, and feeding the survivors back
    # keeps all of them in the same order (idempotence)
    bsel = int(np.argmax(valid))
    dec = _np(bbox_utils.get_bboxes_from_deltas(priors, torch.as_tensor(d[bsel:bsel + 1]) * torch.tensor(hp["variances"])))
    thr = 0.5
    kw = dict(max_output_size_per_class=200, max_total_size=200, iou_threshold=0.5, score_threshold=thr, clip_boxes=False)
    ub, us, ul, uv = [_np(t) for t in bbox_utils.non_max_suppression(dec.reshape(1, N, 1, 4), p[bsel:bsel + 1], **kw)]
    # column 0 (background) takes part in the raw op; the decoder masks it (models/decoder.py:43-50)
    v = int(uv[0])
    assert v > 0
    assert _max_same_class_iou(ub[0, :v], ul[0, :v]) <= 0.5 + 1e-6
    pb = np.zeros((1, v, L), np.float32)
    pb[0, np.arange(v), ul[0, :v].astype(int)] = us[0, :v]
    ob, os_, ol, ov = [_np(t) for t in bbox_utils.non_max_suppression(ub[0, :v].reshape(1, v, 1, 4), pb, **kw)]
    assert int(ov[0]) == v
    np.testing.assert_array_equal(os_[0, :v], us[0, :v])
    np.testing.assert_array_equal(ol[0, :v], ul[0, :v])
    np.testing.assert_array_equal(ob[0, :v], ub[0, :v])


def test_decoder_c5_shard_24564_anchors():
    """BASELINE configs[4] per-GPU shard of the decoder stress: batch 16, N = 24 564 anchors, all
    16 images against the plain-C oracle (indices / labels / scores bit-exact)."""
    from models.decoder import SSDDecoder
    B, N, L = 16, 24564, 21
    rng = np.random.default_rng(77)
    c = rng.uniform(0.05, 0.95, (N, 2)); sz = rng.uniform(0.02, 0.4, (N, 2))
    pri = np.clip(np.concatenate([c - sz / 2, c + sz / 2], -1), 0, 1).astype(np.float32)
    d, pr = helpers.decoder_inputs(B, N, L, seed=78, boost_frac=0.10)
    dec = SSDDecoder(pri, helpers.VARIANCES)
    b, l, s = dec.call([d, pr], return_indices=True)
    rb, rl, rs, rv, ri = co.decode_nms(d, pr, pri, helpers.VARIANCES)
    assert rv.min() == 200
    np.testing.assert_array_equal(_np(dec.last_valid_detections), rv)
    np.testing.assert_array_equal(_np(dec.last_kept_indices), ri)
    np.testing.assert_array_equal(_np(l), rl)
    np.testing.assert_array_equal(_np(s), rs)
    np.testing.assert_allclose(_np(b), rb, atol=1e-4, rtol=0)
    _check_nms_contract(_np(b), _np(l), _np(s), L)


@pytest.mark.parametrize("backbone", ["mobilenet_v2", "vgg16"])
def test_stream_plans_agree(backbone):
    """forward_impl's per-layer stream plan: heads overlapped on side streams (default), the swapped
    roles (tail_on_side), no overlap, graph replay on and off -- all the same kernels in another
    order, so the outputs are bitwise identical."""
    from models.decoder import get_decoder_model
    from utils import bbox_utils
    if backbone == "mobilenet_v2":
        from models.ssd_mobilenet_v2 import get_model
    else:
        from models.ssd_vgg16 import get_model
    hp = helpers.hyper_params(backbone)
    w = helpers.synthetic_weights(backbone, hp)
    m = get_model(hp, max_batch=6)
    m.set_weights(w)
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    dm = get_decoder_model(m, priors, hp)
    x = helpers.images(6, 300, seed=4)
    ref = None
    for opts in ({"overlap_heads": 1, "tail_on_side": 0, "use_graph": 1}, {"overlap_heads": 1, "tail_on_side": 1, "use_graph": 1},
                 {"overlap_heads": 1, "tail_on_side": 1, "use_graph": 0}, {"overlap_heads": 0, "tail_on_side": 0, "use_graph": 0},
                 {"overlap_heads": 1, "tail_on_side": 0, "use_graph": 0}):
        for k, v in opts.items():
            m.set_option(k, v)
        outs = []
        for rep in range(3):                      # eager first call, capture, replay
            outs.append([r.copy() for r in dm.predict_on_batch(x)])
        if ref is None:
            ref = outs[0]
            assert (ref[2] > 0).sum() > 0
        for o in outs:
            for a, b in zip(o, ref):
                np.testing.assert_array_equal(a, b, err_msg=str(opts))


def test_two_lanes_match_one_lane():
    """DecoderModel with two batches in flight (``submit`` / ``predict`` on two replicas of the net and
    two streams): every step's detections are bitwise those of the one-step-at-a-time path, in order,
    also after the base model's weights change (the replicas follow)."""
    from models.decoder import get_decoder_model
    from models.ssd_mobilenet_v2 import get_model
    from utils import bbox_utils
    hp = helpers.hyper_params("mobilenet_v2")
    w = helpers.synthetic_weights("mobilenet_v2", hp)
    m = get_model(hp, max_batch=6)
    m.set_weights(w)
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    dm1 = get_decoder_model(m, priors, hp, lanes=1)
    dm2 = get_decoder_model(m, priors, hp, lanes=2)
    batches = [helpers.images(6, 300, seed=40 + i) for i in range(5)]
    ref = [tuple(t.cpu().numpy() for t in dm1(b)) for b in batches]
    assert any((r[2] > 0).sum() > 0 for r in ref)
    outs = [dm2.submit(b) for b in batches]
    dm2.wait()
    torch.cuda.synchronize()
    assert "two_lanes_used" in dm2.lane_calibration
    for o, r in zip(outs, ref):
        for a, b in zip(o, r):
            np.testing.assert_array_equal(a.cpu().numpy(), b)
    dm3 = get_decoder_model(m, priors, hp, lanes=3)            # three replicas: still every step's own result, in order
    outs3 = [dm3.submit(b) for b in batches]
    dm3.wait()
    torch.cuda.synchronize()
    for o, r in zip(outs3, ref):
        for a, b in zip(o, r):
            np.testing.assert_array_equal(a.cpu().numpy(), b)
    del dm3
    stacked = np.concatenate(batches)
    p1 = dm1.predict(stacked, batch_size=6)
    p2 = dm2.predict(stacked, batch_size=6)
    for a, b in zip(p1, p2):
        np.testing.assert_array_equal(a, b)
    # ragged last batch (20 = 3 x 6 + 2) and a `steps` limit
    for a, b in zip(dm1.predict(stacked[:20], batch_size=6), dm2.predict(stacked[:20], batch_size=6)):
        assert a.shape[0] == 20
        np.testing.assert_array_equal(a, b)
    for a, b in zip(dm1.predict(stacked, batch_size=6, steps=2), dm2.predict(stacked, batch_size=6, steps=2)):
        assert a.shape[0] == 12
        np.testing.assert_array_equal(a, b)
    # new weights: the replicas are rebuilt from the base model
    w2 = {k: (v * np.float32(1.01) if k.endswith("conv_label_output/bias") else v) for k, v in w.items()}
    m.set_weights(w2)
    ref2 = [tuple(t.cpu().numpy() for t in dm1(b)) for b in batches[:3]]
    outs2 = [dm2.submit(b) for b in batches[:3]]
    dm2.wait()
    torch.cuda.synchronize()
    for o, r in zip(outs2, ref2):
        for a, b in zip(o, r):
            np.testing.assert_array_equal(a.cpu().numpy(), b)


def test_lane_check_retries_on_fresh_streams(monkeypatch):
    """Where the lanes do not pay on the streams they got (which hardware queue a stream lands on depends on what the
    process created before), ``DecoderModel`` repeats its check on fresh streams before it falls back to one lane: a
    first check forced to fail must be followed by a second one on OTHER streams, and the results stay those of the
    one-at-a-time path bit for bit."""
    from models import decoder as dec
    from models.ssd_mobilenet_v2 import get_model
    from utils import bbox_utils
    hp = helpers.hyper_params("mobilenet_v2")
    m = get_model(hp, max_batch=4)
    m.set_weights(helpers.synthetic_weights("mobilenet_v2", hp))
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    dm1 = dec.get_decoder_model(m, priors, hp, lanes=1)
    dm3 = dec.get_decoder_model(m, priors, hp, lanes=3)
    seen = []
    real = dec.DecoderModel._check_lanes_pay_once

    def once(self, x):
        real(self, x)
        seen.append([st.cuda_stream for st in self._lane_streams])
        if len(seen) == 1:
            self._lanes_active = False              # pretend the first set of streams shared a hardware queue

    monkeypatch.setattr(dec.DecoderModel, "_check_lanes_pay_once", once)
    batches = [helpers.images(4, 300, seed=70 + i) for i in range(4)]
    ref = [tuple(t.cpu().numpy() for t in dm1(b)) for b in batches]
    outs = [dm3.submit(b) for b in batches]
    dm3.wait()
    torch.cuda.synchronize()
    assert len(seen) >= 2 and not set(seen[0]) & set(seen[1])
    assert len(dm3.lane_calibration["attempts_ms_per_step"]) == len(seen)
    for o, r in zip(outs, ref):
        for a, b in zip(o, r):
            np.testing.assert_array_equal(a.cpu().numpy(), b)


def test_uint8_pinned_batches_are_converted_on_the_lane():
    """``DecoderModel.submit`` of a PINNED UINT8 batch [B,H,W,3] (the reference's images before ``preprocessing``,
    utils/data_utils.py:17-22): the lane copies the bytes (a quarter of the float batch over PCIe), runs
    ``ssd_preprocess`` (x 1/255 + TF2 bilinear resize to the net's input) and the step on its own stream.  Bitwise the
    detections of the one-at-a-time path on ``data_utils.preprocess_batch`` of the same bytes -- at the net's own size
    and through a resize (280 x 320 sources) -- for pageable uint8 arrays and the one-lane model as well."""
    import ssd_hip
    from models.decoder import get_decoder_model
    from models.ssd_mobilenet_v2 import get_model
    from utils import bbox_utils, data_utils
    hp = helpers.hyper_params("mobilenet_v2")
    m = get_model(hp, max_batch=4)
    m.set_weights(helpers.synthetic_weights("mobilenet_v2", hp))
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    dm1 = get_decoder_model(m, priors, hp, lanes=1)
    dm3 = get_decoder_model(m, priors, hp, lanes=3)
    rng = np.random.RandomState(5)
    for (H, W) in ((300, 300), (280, 320)):
        raw = []
        for i in range(5):
            u = (helpers.images(4, 300, seed=60 + i) * 255.0 + 0.5).astype(np.uint8)
            u = np.pad(u, ((0, 0), (0, 0), (0, max(W - 300, 0)), (0, 0)), mode="edge")[:, :H, :W]
            raw.append(np.ascontiguousarray(u ^ rng.randint(0, 2, u.shape).astype(np.uint8)))
        ref = [tuple(t.cpu().numpy() for t in dm1(data_utils.preprocess_batch(r, 300, 300))) for r in raw]
        assert any((r[2] > 0).sum() > 0 for r in ref)
        hosts = []
        for r in raw:
            hb = ssd_hip.pinned_empty(r.shape, dtype=torch.uint8)
            hb.numpy()[...] = r
            hosts.append(hb)
        outs = [dm3.submit(hb) for hb in hosts]
        dm3.wait()
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            for a, b in zip(o, r):
                np.testing.assert_array_equal(a.cpu().numpy(), b)
        for a, b in zip(dm1(raw[0]), ref[0]):                                  # one lane, pageable uint8 array
            np.testing.assert_array_equal(a.cpu().numpy(), b)
        o = dm3.submit(raw[1])                                                 # lanes, pageable uint8 array
        dm3.wait()
        torch.cuda.synchronize()
        for a, b in zip(o, ref[1]):
            np.testing.assert_array_equal(a.cpu().numpy(), b)


def test_predict_fused_softmax_matches_the_layered_decoder_across_batch_sizes():
    """``ssd_net_predict`` folds the softmax into the decoder's compaction kernel and keeps its candidate counters zero
    between calls without a memset (csrc/ssd_bbox.hip): bitwise the detections of forward (softmax layer) +
    ``SSDDecoder`` for a sequence of DIFFERENT batch sizes on one net (the workspace is carved once, for max_batch --
    a per-call layout once let a small batch's kept counts land on a larger batch's candidate counters), and with the
    option off."""
    from models.decoder import SSDDecoder
    from models.ssd_mobilenet_v2 import get_model
    from utils import bbox_utils
    hp = helpers.hyper_params("mobilenet_v2")
    w = helpers.synthetic_weights("mobilenet_v2", hp)
    m = get_model(hp, max_batch=6)
    m.set_weights(w)
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    dec = SSDDecoder(priors, hp["variances"])
    x = helpers.images(6, 300, seed=11)
    for fuse in (1, 0, 1):
        m.set_option("fuse_softmax", fuse)
        for B in (6, 2, 5, 1, 6, 3):
            b, l, s, v = [_np(t) for t in m.predict_on_device(x[:B], priors, hp["variances"])]
            d, p = m(x[:B])
            rb, rl, rs = [_np(t) for t in dec([d, p])]
            assert (s > 0).sum() > 0
            np.testing.assert_array_equal(s, rs)
            np.testing.assert_array_equal(l, rl)
            np.testing.assert_array_equal(b, rb)
            np.testing.assert_array_equal(v, _np(dec.last_valid_detections))


def test_lanes_hint_changes_only_the_summation_grouping():
    """With N lanes in flight the whole-image blocks split an image's expanded channels over fewer workgroups
    (option lanes_hint: B x groups x lanes fills the CUs).  The lanes' results are bitwise those of a single net
    with the same hint, and equal to the unhinted net's up to the fp32 regrouping of the partial sums."""
    from models.decoder import get_decoder_model
    from models.ssd_mobilenet_v2 import get_model
    from utils import bbox_utils
    hp = helpers.hyper_params("mobilenet_v2")
    m = get_model(hp, max_batch=24)              # 24 images: 12 channel groups alone, 4 with three lanes in flight
    m.set_weights(helpers.synthetic_weights("mobilenet_v2", hp))
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    x = helpers.images(24, 300, seed=77)
    d1, p1 = (t.cpu().numpy() for t in m(x))
    dm3 = get_decoder_model(m, priors, hp, lanes=3)
    outs = [dm3.submit(x) for _ in range(3)]
    dm3.wait()
    torch.cuda.synchronize()
    outs = [[t.cpu().numpy() for t in o] for o in outs]
    m.set_option("lanes_hint", 3)
    d3, p3 = (t.cpu().numpy() for t in m(x))
    ref = [t.cpu().numpy() for t in get_decoder_model(m, priors, hp, lanes=1)(x)]
    m.set_option("lanes_hint", 1)
    assert np.abs(d3 - d1).max() <= 2e-5 and np.abs(p3 - p1).max() <= 2e-5
    assert (ref[2] > 0).sum() > 0
    for o in outs:
        for a, b in zip(o, ref):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("extra", [[], ["--lanes", "1"], ["--train"], ["--other-configs", "on"]])
def test_bench_line_contract(extra, tmp_path):
    """`python bench.py --steps K --warmup W` prints ONE JSON line with the driver's keys; value, ms_per_step
    and the batch agree; `roofline` / step figures are self-consistent (small batch: this checks the
    contract, not the performance)."""
    import json, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSD_HIP_TUNE_CACHE=str(tmp_path))
    env.pop("GPU_MAX_HW_QUEUES", None)          # (the suite pins the runtime's default in conftest; bench.py chooses its own: one queue per lane)
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "8", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["unit"] == "images/sec" and r["n_gpus"] == 1 and r["steps"] == 4 and r["warmup"] == 2
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "f32"
    assert "workload" in r["config"] and "model" not in r["config"] and "synthetic" in r["data"]
    assert abs(r["value"] - 8 / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]
    if "--other-configs" in extra:
        # the driver's one invocation also times the other single-GPU BASELINE configs, each in a fresh process at ITS size
        # (VGG16 B=32, the 512x512 B=16 shard fp32 + bf16, the training step B=32 fp32 + bf16, ...): labelled sub-records,
        # `value` stays the headline's
        oc = r["other_configs"]
        assert len(oc) >= 5 and not [c for c in oc if "error" in c], [c.get("error") for c in oc]
        for c in oc:
            assert c["ms_per_step"] > 0 and c["images_per_sec"] > 0 and c["steps"] >= 10 and c["timing_spread"]["repeats"] == 3
            assert 0 < c["roofline"]["frac"] < 1 and c["dtype"] in ("f32", "bf16") and c["workload"]
        want = {"--backbone vgg16 --batch 32": "f32", "--img-size 512 --batch 16": "f32", "--img-size 512 --batch 16 --dtype bf16": "bf16",
                "--train --batch 32": "f32", "--train --batch 32 --dtype bf16": "bf16"}
        got = {c["args"]: c["dtype"] for c in oc}
        assert all(got.get(k) == v for k, v in want.items()), got
        extra = []
    else:
        assert "other_configs" not in r            # auto: only beside the default workload (B=64)
    if "--train" not in extra:
        ro = r["roofline"]
        assert ro["bound"] == "mfma" and ro["unit"] == "TFLOP/s" and ro["peak_detail"]["fp32_mfma"] == 157.3
        assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and 0 < ro["frac"] < 1       # executed FLOPs: a real fraction
        assert ro["achieved_algorithmic"] >= ro["achieved"] * (1 - 1e-9) and "executed" in ro["frac_kind"]
        assert abs(ro["peak_detail"]["split_bf16_fp32_equivalent"] - 16 * 157.3 / 6) < 1e-9     # dense bf16 peak / six products
        assert ro["kernel_ms_per_step"] > 0 and 0 < ro["frac_step_vs_fp32_mfma_peak"] < 1 and ro["peak_detail"]["fp32_mfma"] <= ro["peak"] <= ro["peak_detail"]["split_bf16_fp32_equivalent"] + 1e-9
        dk = ro["dominant_kernel"]
        assert dk["ms_per_launch"] > 0 and 0 < dk["frac"] < 1 and dk["ms_per_launch"] < r["ms_per_step"]
        # default: two batches in flight (one in-order stream / hardware queue per lane) are the headline, one step at
        # a time is reported beside it (and vice versa)
        assert r["config"]["inputs_rotated"] == 4 and "device-resident" in r["config"]["results"]
        assert r["config"]["batches_in_flight_per_gpu"] == (1 if extra else 3)
        assert r["other_mode"]["ms_per_step"] > 0 and ("three batches" in r["other_mode"]["mode"]) == bool(extra)
        if not extra:
            lc = r["config"]["lane_calibration"]
            assert lc["hw_queues"] == "3" and lc["pair"] is None          # one hardware queue per lane, no stream-pair search
        assert r["config"]["nms_active"] is True
        kt = r["config"]["kernel_table"]
        assert kt["source"] in ("shipped", "cache", "memo", "autotune") and len(kt["table_sha16"]) == 16
    else:
        assert 0 < r["roofline"]["frac"] < 1


def test_predict_ascending_batch_sizes():
    """ssd_net_predict's head-output scratch must follow a re-finalize with a larger max_batch
    (B=1 then B=8 on the same model used to overflow the B=1-sized buffers), and the captured
    graphs must not survive the reallocation."""
    from models.decoder import get_decoder_model
    from models.ssd_mobilenet_v2 import get_model
    from utils import bbox_utils
    hp = helpers.hyper_params("mobilenet_v2")
    w = helpers.synthetic_weights("mobilenet_v2", hp)
    m = get_model(hp)
    m.set_weights(w)
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    dm = get_decoder_model(m, priors, hp)
    x = helpers.images(8, 300, seed=0)
    one = [r.copy() for r in dm.predict_on_batch(x[:1])]
    one_again = dm.predict_on_batch(x[:1])                      # graph replay at B=1
    for a, b in zip(one, one_again):
        np.testing.assert_array_equal(a, b)
    eight = dm.predict_on_batch(x)                               # re-finalize at 8: scratch regrown
    eight_again = dm.predict_on_batch(x)
    for a, b in zip(eight, eight_again):
        np.testing.assert_array_equal(a, b)
    back = dm.predict_on_batch(x[:1])
    ref = get_model(hp, max_batch=8)
    ref.set_weights(w)
    rdm = get_decoder_model(ref, priors, hp)
    for a, b in zip(eight, rdm.predict_on_batch(x)):      # same shapes -> same kernel table (shipped / memo) -> same bits
        np.testing.assert_array_equal(a, b)
    assert ref.get_tuning() == m.get_tuning()
    assert (eight[2] > 0).sum() > 0
    for a, b in zip(back, [r[:1] for r in eight]):              # same tiles at any batch? at least close
        assert np.abs(a - b).max() <= 1e-4


def test_pinned_host_batches_through_the_lanes():
    """``DecoderModel.submit`` of a pinned host batch (``ssd_hip.pinned_empty``): the H2D copy runs on the lane's own
    stream in front of its step; results are bitwise those of the resident-input path, also when a pinned buffer is
    reused for later steps and when the batch is smaller than the lane's buffer."""
    import ssd_hip as h
    from models.decoder import get_decoder_model
    from models.ssd_mobilenet_v2 import get_model
    from utils import bbox_utils
    hp = helpers.hyper_params("mobilenet_v2")
    w = helpers.synthetic_weights("mobilenet_v2", hp)
    m = get_model(hp, max_batch=6)
    m.set_weights(w)
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    dm1 = get_decoder_model(m, priors, hp, lanes=1)
    dm3 = get_decoder_model(m, priors, hp, lanes=3)
    batches = [helpers.images(6 if i != 3 else 4, 300, seed=70 + i) for i in range(7)]
    ref = [tuple(t.cpu().numpy() for t in dm1(b)) for b in batches]
    hosts = []
    for b in batches:
        hb = h.pinned_empty(b.shape)
        hb.numpy()[...] = b
        hosts.append(hb)
    outs = [dm3.submit(hb) for hb in hosts]
    dm3.wait()
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        for a, b in zip(o, r):
            np.testing.assert_array_equal(a.cpu().numpy(), b)
    # predict() over an iterable of pinned batches
    p = dm3.predict(hosts)
    for k in range(3):
        np.testing.assert_array_equal(p[k], np.concatenate([r[k] for r in ref], 0))


def test_full_size_c2_under_the_serving_queue_setup():
    """The suite runs on the runtime's default hardware queues (conftest); the bench and ``predictor.py`` run under
    GPU_MAX_HW_QUEUES=3 (``ssd_hip.configure_serving()``).  One more pass of the C2 full-size parity test (B=64 forward +
    decode/NMS against the NumPy / torch-CPU / C oracles) and of the lanes' bitwise test in a fresh process under THAT setup."""
    import subprocess
    import sys
    if os.environ.get("SSD_TEST_NESTED") == "1":
        pytest.skip("already the nested pass")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="3", SSD_TEST_NESTED="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(repo, "tests", "test_fullsize_gpu.py"), "-k",
           "test_full_batch_forward_and_decode and mobilenet_v2-64-300 or test_two_lanes_match_one_lane"]
    out = subprocess.run(cmd, env=env, text=True, capture_output=True, timeout=1200, cwd=repo)
    assert out.returncode == 0 and " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_default_serving_setup_gives_the_lanes_to_a_plain_predict(tmp_path):
    """What the serving entry points get (VERDICT r4 #6a, opt-in since ADVICE r5): in a FRESH process,
    ``ssd_hip.configure_serving()`` -- what ``predictor.py`` / ``bench.py`` call first thing --
    limits the HIP runtime to three hardware queues, ``get_decoder_model(model, priors, hp)`` is a three-lane model in
    auto mode, a ``predict`` over enough batches runs them three in flight (the lane check ran) and returns exactly the
    one-step-at-a-time detections; a two-batch ``predict`` stays on the classic path (no replicas built)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path[:0] = [%r, %r, %r]
import numpy as np
import ssd_hip
assert "GPU_MAX_HW_QUEUES" not in os.environ and ssd_hip.configure_serving()
assert os.environ["GPU_MAX_HW_QUEUES"] == "3"
import helpers
from models.decoder import get_decoder_model
from models.ssd_mobilenet_v2 import get_model
from utils import bbox_utils
hp = helpers.hyper_params("mobilenet_v2")
m = get_model(hp, max_batch=4)
m.set_weights(helpers.synthetic_weights("mobilenet_v2", hp))
priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
dm = get_decoder_model(m, priors, hp)
assert dm.lanes == 3 and dm.auto_lanes
x = np.concatenate([helpers.images(4, 300, seed=70 + i) for i in range(8)])
short = dm.predict(x[:8], batch_size=4)
assert not dm._lane_models, "a two-batch predict must not build lane replicas"
full = dm.predict(x, batch_size=4)
assert len(dm._lane_models) == 3 and "two_lanes_used" in dm.lane_calibration
ref = get_decoder_model(m, priors, hp, lanes=1).predict(x, batch_size=4)
assert (ref[2] > 0).sum() > 0
for a, b in zip(full, ref):
    np.testing.assert_array_equal(a, b)
for a, b in zip(short, ref):
    np.testing.assert_array_equal(a, b[:8])
print("default-serving-ok", dm.lane_calibration)
''' % (repo, os.path.join(repo, "tf-ssd_amd"), os.path.join(repo, "tests"))
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "SSD_HIP_HW_QUEUES", "SSD_HIP_LANES")}
    out = subprocess.run([sys.executable, "-c", code], env=env, text=True, capture_output=True, timeout=900)
    assert out.returncode == 0 and "default-serving-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
