"""Drop-in for the reference's ``utils/eval_utils.py`` (VOC2007 11-point mAP; SURVEY.md 8f
row N2).  The pairwise IoU runs on the GPU (``bbox_utils.generate_iou_map``); the per-image
bookkeeping is host NumPy, with the reference's quirks kept (see docstrings)."""
import numpy as np

from utils import bbox_utils


def init_stats(labels):
    """reference utils/eval_utils.py:5-17."""
    stats = {}
    for i, label in enumerate(labels):
        if i == 0:
            continue
        stats[i] = {"label": label, "total": 0, "tp": [], "fp": [], "scores": []}
    return stats


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def update_stats(pred_bboxes, pred_labels, pred_scores, gt_boxes, gt_labels, stats):
    """reference utils/eval_utils.py:19-54.  Predictions are visited in descending-IoU order
    (not score order, :23,33); label 0 rows are padding (:35-36); a prediction is a TP iff
    IoU >= 0.5, labels match and that GT index was not used before (:46-48)."""
    iou_map = _np(bbox_utils.generate_iou_map(pred_bboxes, gt_boxes))          # [B, T, G]
    pred_labels, pred_scores, gt_labels = _np(pred_labels), _np(pred_scores), _np(gt_labels)
    merged = iou_map.max(-1)
    max_idx = iou_map.argmax(-1)
    sorted_ids = np.argsort(-merged, axis=-1, kind="stable")
    uniq, counts = np.unique(gt_labels.reshape(-1), return_counts=True)
    for lab, cnt in zip(uniq, counts):
        if lab == -1:
            continue
        stats[int(lab)]["total"] += int(cnt)
    for b in range(merged.shape[0]):
        used = []
        for sid in sorted_ids[b]:
            pl = pred_labels[b, sid]
            if pl == 0:
                continue
            iou = merged[b, sid]
            gt_id = int(max_idx[b, sid])
            gl = int(gt_labels[b, gt_id])
            pl = int(pl)
            st = stats[pl]
            st["scores"].append(pred_scores[b, sid])
            st["tp"].append(0)
            st["fp"].append(0)
            if iou >= 0.5 and pl == gl and gt_id not in used:
                st["tp"][-1] = 1
                used.append(gt_id)
            else:
                st["fp"][-1] = 1
    return stats


def calculate_ap(recall, precision):
    """reference utils/eval_utils.py:56-64 (11-point interpolation)."""
    ap = 0
    for r in np.arange(0, 1.1, 0.1):
        prec_rec = precision[recall >= r]
        if len(prec_rec) > 0:
            ap += np.amax(prec_rec)
    ap /= 11
    return ap


def calculate_mAP(stats):
    """reference utils/eval_utils.py:66-85 (a class with no GT/predictions yields NaN, as there)."""
    aps = []
    for label in stats:
        s = stats[label]
        tp, fp, scores = np.array(s["tp"]), np.array(s["fp"]), np.array(s["scores"])
        ids = np.argsort(-scores)
        total = s["total"]
        acc_tp = np.cumsum(tp[ids]) if len(ids) else np.array([])
        acc_fp = np.cumsum(fp[ids]) if len(ids) else np.array([])
        with np.errstate(divide="ignore", invalid="ignore"):
            recall = acc_tp / total
            precision = acc_tp / (acc_fp + acc_tp)
        ap = calculate_ap(recall, precision)
        s["recall"], s["precision"], s["AP"] = recall, precision, ap
        aps.append(ap)
    return stats, np.mean(aps)


def evaluate_predictions(dataset, pred_bboxes, pred_labels, pred_scores, labels, batch_size):
    """reference utils/eval_utils.py:87-97."""
    stats = init_stats(labels)
    for batch_id, image_data in enumerate(dataset):
        imgs, gt_boxes, gt_labels = image_data
        start = batch_id * batch_size
        end = start + batch_size
        stats = update_stats(pred_bboxes[start:end], pred_labels[start:end], pred_scores[start:end],
                             gt_boxes, gt_labels, stats)
    stats, mAP = calculate_mAP(stats)
    print("mAP: {}".format(float(mAP)))
    return stats
