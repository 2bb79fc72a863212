"""CPU oracle (TEST INFRASTRUCTURE, not product code) for the SSD box math.

NumPy restatement, op for op, of the reference's closed-form box functions.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``tf-ssd_amd/``) never does.

PARITY STATUS: **parity unpinned**.  The reference ships no tests / golden vectors and
TensorFlow cannot be imported in the build container, so this
restatement is pinned only by the reference *source text* plus IEEE-754 and by the
known-answer values SURVEY.md 8c lists for the prior boxes (``tests/test_oracle_bbox.py``).
Behaviour of third-party TF kernels (``tf.image.combined_non_max_suppression`` tie
order, Eigen ``exp``/``log`` last-ulp) is restated from the published TF 2.0-2.2 CPU
kernel (``tensorflow/core/kernels/non_max_suppression_op.cc``, ``CombinedNonMaxSuppression``)
and could not be executed here.

All citations are relative to /root/reference.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- A1
def get_scale_for_nth_feature_map(k, m=6, scale_min=0.2, scale_max=0.9):
    """utils/bbox_utils.py:115-124 -- Python float64 arithmetic."""
    return scale_min + ((scale_max - scale_min) / (m - 1)) * (k - 1)


# --------------------------------------------------------------------------- A2
def generate_base_prior_boxes(aspect_ratios, feature_map_index, total_feature_map):
    """utils/bbox_utils.py:126-147.

    ``tf.sqrt(python_float)`` is an fp32 sqrt; ``python_float / fp32_tensor`` casts the
    scalar to fp32 first (:141-142).  The extra box multiplies the two scales in float64,
    casts, then takes the fp32 sqrt (:145).  Given ratios first, extra square last.
    """
    current_scale = get_scale_for_nth_feature_map(feature_map_index, m=total_feature_map)
    next_scale = get_scale_for_nth_feature_map(feature_map_index + 1, m=total_feature_map)
    base = []
    two = F32(2.0)
    for aspect_ratio in aspect_ratios:
        s = np.sqrt(F32(aspect_ratio))
        height = F32(current_scale) / s
        width = F32(current_scale) * s
        base.append([-height / two, -width / two, height / two, width / two])
    height = width = np.sqrt(F32(current_scale * next_scale))
    base.append([-height / two, -width / two, height / two, width / two])
    return np.asarray(base, dtype=F32)


# --------------------------------------------------------------------------- A3
def generate_prior_boxes(feature_map_shapes, aspect_ratios):
    """utils/bbox_utils.py:149-176.

    ``tf.range(f) / f`` is an int32 true-divide => float64; ``+ stride/2`` in float64;
    *then* cast to fp32 (:165).  ``tf.meshgrid`` default 'xy' indexing + row-major reshape
    => prior index = (y*f + x)*A + a (:166-172).  Final clip to [0,1] (:176).
    """
    out = []
    for i, f in enumerate(feature_map_shapes):
        base = generate_base_prior_boxes(aspect_ratios[i], i + 1, len(feature_map_shapes))
        stride = 1 / f
        grid = (np.arange(0, f, dtype=np.int32) / f + stride / 2).astype(F32)
        grid_x, grid_y = np.meshgrid(grid, grid)
        fx, fy = grid_x.reshape(-1), grid_y.reshape(-1)
        grid_map = np.stack([fy, fx, fy, fx], -1)
        pb = base.reshape(1, -1, 4) + grid_map.reshape(-1, 1, 4)
        out.append(pb.reshape(-1, 4))
    return np.clip(np.concatenate(out, 0), F32(0), F32(1)).astype(F32)


# --------------------------------------------------------------------------- D1
def get_bboxes_from_deltas(prior_boxes, deltas):
    """utils/bbox_utils.py:61-85 (no clipping here)."""
    p = np.asarray(prior_boxes, F32)
    d = np.asarray(deltas, F32)
    half = F32(0.5)
    pw = p[..., 3] - p[..., 1]
    ph = p[..., 2] - p[..., 0]
    pcx = p[..., 1] + half * pw
    pcy = p[..., 0] + half * ph
    w = np.exp(d[..., 3]) * pw
    h = np.exp(d[..., 2]) * ph
    cx = (d[..., 1] * pw) + pcx
    cy = (d[..., 0] * ph) + pcy
    y1 = cy - (half * h)
    x1 = cx - (half * w)
    y2 = h + y1
    x2 = w + x1
    return np.stack([y1, x1, y2, x2], axis=-1).astype(F32)


# --------------------------------------------------------------------------- M3
def get_deltas_from_bboxes(bboxes, gt_boxes):
    """utils/bbox_utils.py:87-113."""
    b = np.asarray(bboxes, F32)
    g = np.asarray(gt_boxes, F32)
    half = F32(0.5)
    bw = b[..., 3] - b[..., 1]
    bh = b[..., 2] - b[..., 0]
    bcx = b[..., 1] + half * bw
    bcy = b[..., 0] + half * bh
    gw = g[..., 3] - g[..., 1]
    gh = g[..., 2] - g[..., 0]
    gcx = g[..., 1] + half * gw
    gcy = g[..., 0] + half * gh
    bw = np.where(bw == 0, F32(1e-3), bw)
    bh = np.where(bh == 0, F32(1e-3), bh)
    with np.errstate(divide="ignore", invalid="ignore"):
        dx = np.where(gw == 0, F32(0), (gcx - bcx) / bw)
        dy = np.where(gh == 0, F32(0), (gcy - bcy) / bh)
        dw = np.where(gw == 0, F32(0), np.log(gw / bw))
        dh = np.where(gh == 0, F32(0), np.log(gh / bh))
    shape = np.broadcast(dy, dx, dh, dw).shape
    return np.stack([np.broadcast_to(v, shape) for v in (dy, dx, dh, dw)], axis=-1).astype(F32)


# --------------------------------------------------------------------------- M1
def generate_iou_map(bboxes, gt_boxes):
    """utils/bbox_utils.py:27-59 with the default ``transpose_perm`` -- no epsilon.

    bboxes [..., N, 4] (rank 2 or 3), gt_boxes [B, G, 4] (or [G, 4]) -> [..., N, G].
    """
    b = np.asarray(bboxes, F32)
    g = np.asarray(gt_boxes, F32)
    by1, bx1, by2, bx2 = (b[..., i:i + 1] for i in range(4))
    gy1, gx1, gy2, gx2 = (g[..., i:i + 1] for i in range(4))
    gt_area = ((gy2 - gy1) * (gx2 - gx1))[..., 0]
    bbox_area = ((by2 - by1) * (bx2 - bx1))[..., 0]
    T = lambda a: np.swapaxes(a, -1, -2)
    x_top = np.maximum(bx1, T(gx1))
    y_top = np.maximum(by1, T(gy1))
    x_bottom = np.minimum(bx2, T(gx2))
    y_bottom = np.minimum(by2, T(gy2))
    inter = np.maximum(x_bottom - x_top, F32(0)) * np.maximum(y_bottom - y_top, F32(0))
    union = bbox_area[..., None] + gt_area[..., None, :] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / union).astype(F32)


# --------------------------------------------------------------------------- M2
def calculate_actual_outputs(prior_boxes, gt_boxes, gt_labels, hyper_params, return_indices=False):
    """utils/train_utils.py:90-127.

    argmax over G: first max wins (:113); positives are ``max_iou > iou_threshold``
    strictly (:117); no forced best-prior-per-GT match.
    """
    total_labels = hyper_params["total_labels"]
    iou_threshold = F32(hyper_params["iou_threshold"])
    variances = np.asarray(hyper_params["variances"], F32)
    gt_boxes = np.asarray(gt_boxes, F32)
    gt_labels = np.asarray(gt_labels, np.int32)
    iou_map = generate_iou_map(prior_boxes, gt_boxes)                     # [B,N,G]
    # [3P] tf.argmax / tf.reduce_max are Eigen reducers: the accumulator starts at lowest() and is
    # replaced only by a strictly greater value -> first max wins and a NaN IoU (0/0: degenerate
    # prior against a padded ground-truth box) is never selected (NumPy's argmax would pick it)
    safe = np.where(np.isnan(iou_map), F32(-np.inf), iou_map)
    max_idx = np.argmax(safe, axis=2).astype(np.int32)                    # first max wins (:113)
    merged = np.max(safe, axis=2)                                         # (:115)
    pos = merged > iou_threshold
    gt_map = np.take_along_axis(gt_boxes, max_idx[..., None], axis=1)     # [B,N,4]
    exp_gt = np.where(pos[..., None], gt_map, F32(0))
    deltas = get_deltas_from_bboxes(prior_boxes, exp_gt) / variances
    lab_map = np.take_along_axis(gt_labels, max_idx, axis=1)
    exp_lab = np.where(pos, lab_map, 0).astype(np.int32)
    onehot = (exp_lab[..., None] == np.arange(total_labels, dtype=np.int32)).astype(F32)
    if return_indices:
        return deltas.astype(F32), onehot, exp_lab, max_idx
    return deltas.astype(F32), onehot


# --------------------------------------------------------------------------- D2
def _nms_iou(a, b):
    """[3P] TF ``IOU`` helper of the (Combined)NonMaxSuppression CPU kernel (Appendix B.3)."""
    ymin_i, ymax_i = min(a[0], a[2]), max(a[0], a[2])
    xmin_i, xmax_i = min(a[1], a[3]), max(a[1], a[3])
    ymin_j, ymax_j = min(b[0], b[2]), max(b[0], b[2])
    xmin_j, xmax_j = min(b[1], b[3]), max(b[1], b[3])
    area_i = F32(ymax_i - ymin_i) * F32(xmax_i - xmin_i)
    area_j = F32(ymax_j - ymin_j) * F32(xmax_j - xmin_j)
    if area_i <= 0 or area_j <= 0:
        return F32(0)
    iy = max(F32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), F32(0))
    ix = max(F32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), F32(0))
    inter = F32(iy * ix)
    return F32(inter / F32(F32(area_i + area_j) - inter))


def combined_non_max_suppression(boxes, scores, max_output_size_per_class, max_total_size,
                                 iou_threshold=0.5, score_threshold=float("-inf"),
                                 clip_boxes=True, return_indices=False):
    """[3P] ``tf.image.combined_non_max_suppression`` (TF 2.0-2.2 CPU kernel), q == 1 case,
    ``pad_per_class=False``.  Called from utils/bbox_utils.py:21-25.

    boxes [B,N,1,4] or [B,N,4]; scores [B,N,C].  Tie order, which TF leaves unspecified
    (score-only comparator on an unstable heap/sort), is fixed to the build's documented
    rule: per class, equal scores -> lower anchor index first; in the per-image merge,
    equal scores -> lower anchor index, then lower class index.
    """
    boxes = np.asarray(boxes, F32)
    if boxes.ndim == 4:
        assert boxes.shape[2] == 1
        boxes = boxes[:, :, 0, :]
    scores = np.asarray(scores, F32)
    B, N, C = scores.shape
    T = int(max_total_size)
    iou_thr, score_thr = F32(iou_threshold), F32(score_threshold)
    out_b = np.zeros((B, T, 4), F32)
    out_s = np.zeros((B, T), F32)
    out_c = np.zeros((B, T), F32)
    out_i = np.full((B, T), -1, np.int32)
    valid = np.zeros((B,), np.int32)
    for b in range(B):
        merged = []  # (score, anchor, class)
        for c in range(C):
            sc = scores[b, :, c]
            cand = np.nonzero(sc > score_thr)[0]
            if cand.size == 0:
                continue
            order = cand[np.lexsort((cand, -sc[cand].astype(np.float64)))]
            selected = []
            for i in order:
                if len(selected) >= min(int(max_output_size_per_class), N):
                    break
                keep = True
                for j in reversed(selected):
                    if _nms_iou(boxes[b, i], boxes[b, j]) > iou_thr:
                        keep = False
                        break
                if keep:
                    selected.append(i)
            merged.extend((float(sc[i]), int(i), c) for i in selected)
        merged.sort(key=lambda t: (-t[0], t[1], t[2]))
        n = min(len(merged), T)
        valid[b] = n
        for r in range(n):
            s, i, c = merged[r]
            bx = boxes[b, i]
            out_b[b, r] = np.clip(bx, F32(0), F32(1)) if clip_boxes else bx
            out_s[b, r] = s
            out_c[b, r] = c
            out_i[b, r] = i
    if return_indices:
        return out_b, out_s, out_c, valid, out_i
    return out_b, out_s, out_c, valid


# --------------------------------------------------------------------------- D3
def ssd_decode(prior_boxes, variances, pred_deltas, pred_label_probs,
               max_total_size=200, score_threshold=0.5, iou_threshold=0.5,
               return_indices=False):
    """models/decoder.py:36-55 ``SSDDecoder.call``: returns (boxes, labels, scores)
    [+ valid, kept anchor indices when ``return_indices``]."""
    d = np.asarray(pred_deltas, F32) * np.asarray(variances, F32)          # :41
    bboxes = get_bboxes_from_deltas(prior_boxes, d)                        # :42
    probs = np.asarray(pred_label_probs, F32)
    amax = np.argmax(probs, -1)[..., None]                                 # :44 first max wins
    labels = np.where(amax != 0, probs, F32(0))                            # :45 all L columns
    res = combined_non_max_suppression(                                    # :49-53
        bboxes[:, :, None, :], labels,
        max_output_size_per_class=max_total_size, max_total_size=max_total_size,
        iou_threshold=iou_threshold, score_threshold=score_threshold,
        return_indices=True)
    fb, fs, fl, valid, idx = res
    if return_indices:
        return fb, fl, fs, valid, idx
    return fb, fl, fs                                                      # :55


# --------------------------------------------------------------------------- U1
def renormalize_bboxes_with_min_max(bboxes, min_max):
    """utils/bbox_utils.py:178-188."""
    b = np.asarray(bboxes, F32)
    y_min, x_min, y_max, x_max = np.split(np.asarray(min_max, F32), 4)
    r = b - np.concatenate([y_min, x_min, y_min, x_min], -1)
    r = r / np.concatenate([y_max - y_min, x_max - x_min, y_max - y_min, x_max - x_min], -1)
    return np.clip(r, F32(0), F32(1)).astype(F32)


def normalize_bboxes(bboxes, height, width):
    """utils/bbox_utils.py:190-205."""
    b = np.asarray(bboxes, F32)
    return np.stack([b[..., 0] / F32(height), b[..., 1] / F32(width),
                     b[..., 2] / F32(height), b[..., 3] / F32(width)], -1).astype(F32)


def denormalize_bboxes(bboxes, height, width):
    """utils/bbox_utils.py:207-222 (tf.round == round-half-to-even == np.round)."""
    b = np.asarray(bboxes, F32)
    return np.round(np.stack([b[..., 0] * F32(height), b[..., 1] * F32(width),
                              b[..., 2] * F32(height), b[..., 3] * F32(width)], -1)).astype(F32)


# --------------------------------------------------------------------------- N4 (input pipeline)
def preprocess_image(img_u8, out_h, out_w):
    """utils/data_utils.py:22-23: ``convert_image_dtype(uint8 -> float32)`` (x * float32(1/255))
    then ``tf.image.resize`` bilinear ([3P] TF 2.x ResizeBilinear CPU kernel: half-pixel centres,
    no antialias, all arithmetic float32).  img_u8 [H,W,C] or [B,H,W,C]."""
    x = np.asarray(img_u8)
    assert x.dtype == np.uint8
    batched = x.ndim == 4
    if not batched:
        x = x[None]
    B, H, W, C = x.shape
    f = x.astype(F32) * F32(1.0 / 255.0)

    def weights(out_size, in_size):
        scale = F32(in_size) / F32(out_size)
        src = (np.arange(out_size, dtype=F32) + F32(0.5)) * scale - F32(0.5)
        fl = np.floor(src)
        lo = np.maximum(fl.astype(np.int64), 0)
        hi = np.minimum(np.ceil(src).astype(np.int64), in_size - 1)
        return lo, hi, (src - fl).astype(F32)
    y0, y1, ly = weights(out_h, H)
    x0, x1, lx = weights(out_w, W)
    lx = lx[None, None, :, None]
    ly = ly[None, :, None, None]
    tl, tr = f[:, y0][:, :, x0], f[:, y0][:, :, x1]
    bl, br = f[:, y1][:, :, x0], f[:, y1][:, :, x1]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    out = (top + (bot - top) * ly).astype(F32)
    return out if batched else out[0]
