"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's VOC07 11-point mAP
(``utils/eval_utils.py:5-97``) as plain Python loops over NumPy arrays.
Only tests/ may import this module; the product (tf-ssd_amd/utils/eval_utils.py) never does.

Pinning: ``init_stats`` / ``calculate_ap`` / ``calculate_mAP`` of the reference are pure
Python/NumPy and WERE EXECUTED in the build container to produce tests/golden/eval_map.npz
(tests/golden/make_eval_golden.py) -- this restatement is checked against those outputs.
``update_stats`` uses TF ops (reduce_max / argmax / argsort / unique_with_counts) and is
restated from the source text; [3P] facts used: ``tf.argsort(direction="DESCENDING")`` orders
equal keys by ascending index (top_k semantics), ``tf.argmax`` returns the first maximum.
"""
import numpy as np

from oracle import bbox_oracle as bo


def init_stats(labels):
    """utils/eval_utils.py:5-17."""
    stats = {}
    for i, label in enumerate(labels):
        if i == 0:
            continue
        stats[i] = {"label": label, "total": 0, "tp": [], "fp": [], "scores": []}
    return stats


def update_stats(pred_bboxes, pred_labels, pred_scores, gt_boxes, gt_labels, stats):
    """utils/eval_utils.py:19-54, loop for loop."""
    pred_bboxes = np.asarray(pred_bboxes, np.float32)
    gt_boxes = np.asarray(gt_boxes, np.float32)
    B, T = pred_bboxes.shape[:2]
    G = gt_boxes.shape[1]
    # :20 generate_iou_map(pred_bboxes [B,T,4], gt_boxes [B,G,4]) -> [B,T,G] (M1, batched boxes)
    iou_map = np.stack([bo.generate_iou_map(pred_bboxes[b], gt_boxes[b:b + 1])[0] for b in range(B)])
    merged = np.empty((B, T), np.float32)                       # :21 reduce_max over G
    max_idx = np.empty((B, T), np.int32)                        # :22 argmax over G (first max)
    for b in range(B):
        for t in range(T):
            best, arg = np.float32(-np.inf), 0      # [3P] Eigen reducers: start at lowest(), strict >: NaN never wins
            for g in range(G):
                if iou_map[b, t, g] > best:
                    best, arg = iou_map[b, t, g], g
            merged[b, t], max_idx[b, t] = best, arg
    # :23 argsort DESCENDING, equal keys by ascending index
    sorted_ids = np.empty((B, T), np.int64)
    for b in range(B):
        keys = [(-float(merged[b, t]), t) for t in range(T)]
        sorted_ids[b] = [k[1] for k in sorted(keys)]
    # :25-30 unique_with_counts over all gt labels; -1 is padding
    flat = np.asarray(gt_labels).reshape(-1)
    for lab in np.unique(flat):
        if lab == -1:
            continue
        stats[int(lab)]["total"] += int((flat == lab).sum())
    # :31-50
    for b in range(B):
        true_labels = []
        for sid in sorted_ids[b]:
            pred_label = pred_labels[b][sid]
            if pred_label == 0:
                continue
            iou = merged[b, sid]
            gt_id = int(max_idx[b, sid])
            gt_label = int(gt_labels[b][gt_id])
            pred_label = int(pred_label)
            score = pred_scores[b][sid]
            stats[pred_label]["scores"].append(score)
            stats[pred_label]["tp"].append(0)
            stats[pred_label]["fp"].append(0)
            if iou >= 0.5 and pred_label == gt_label and gt_id not in true_labels:
                stats[pred_label]["tp"][-1] = 1
                true_labels.append(gt_id)
            else:
                stats[pred_label]["fp"][-1] = 1
    return stats


def calculate_ap(recall, precision):
    """utils/eval_utils.py:56-64 (note ``np.arange(0, 1.1, 0.1)``: thresholds 0.30000000000000004,
    0.6000000000000001, 0.7000000000000001 -- a recall of exactly 3/10 does NOT reach the 4th)."""
    ap = 0
    for r in np.arange(0, 1.1, 0.1):
        prec_rec = precision[recall >= r]
        if len(prec_rec) > 0:
            ap += np.amax(prec_rec)
    ap /= 11
    return ap


def calculate_mAP(stats):
    """utils/eval_utils.py:66-85."""
    aps = []
    for label in stats:
        ls = stats[label]
        tp = np.array(ls["tp"])
        fp = np.array(ls["fp"])
        scores = np.array(ls["scores"])
        ids = np.argsort(-scores)
        total = ls["total"]
        acc_tp = np.cumsum(tp[ids])
        acc_fp = np.cumsum(fp[ids])
        with np.errstate(divide="ignore", invalid="ignore"):
            recall = acc_tp / total
            precision = acc_tp / (acc_fp + acc_tp)
        ap = calculate_ap(recall, precision)
        ls["recall"], ls["precision"], ls["AP"] = recall, precision, ap
        aps.append(ap)
    return stats, np.mean(aps)
