"""Host-side mirror of the reference's ``utils/eval_utils.py`` (VOC2007 11-point mAP; SURVEY.md
8f row N2): same function names, arguments and ``stats`` layout.  The pairwise IoU runs on the
GPU (``bbox_utils.generate_iou_map`` -> ``ssd_iou_map``); the bookkeeping is host NumPy and keeps
the reference's quirks, which the docstrings name.  Cited line numbers are the reference's."""
import numpy as np

from utils import bbox_utils

_IOU_TP = 0.5
# the reference's thresholds are np.arange(0, 1.1, 0.1): 0.30000000000000004, 0.6000000000000001,
# 0.7000000000000001 ... -- a recall of exactly 3/10 does not reach the 4th point (:58)
_RECALL_POINTS = np.arange(0, 1.1, 0.1)


def _host(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def init_stats(labels):
    """One record per foreground class id (index 0, the background, is skipped):
    ``{"label", "total", "tp", "fp", "scores"}`` (utils/eval_utils.py:5-17)."""
    return {cid: {"label": name, "total": 0, "tp": [], "fp": [], "scores": []}
            for cid, name in enumerate(labels) if cid != 0}


def update_stats(pred_bboxes, pred_labels, pred_scores, gt_boxes, gt_labels, stats):
    """utils/eval_utils.py:19-54.  Quirks kept: detections of an image are visited in descending
    best-IoU order, not score order (:23,33); label 0 marks padding and is skipped (:35-36); a
    detection is a true positive iff best IoU >= 0.5, its label equals the label of that
    ground-truth box, and that box was not matched earlier (:46-48); ground-truth label -1
    is padding and not counted (:27-31)."""
    iou = _host(bbox_utils.generate_iou_map(pred_bboxes, gt_boxes))          # [B, T, G]
    det_label, det_score, gt_label = _host(pred_labels), _host(pred_scores), _host(gt_labels)
    for cid, n in zip(*np.unique(gt_label.ravel(), return_counts=True)):
        if cid != -1:
            stats[int(cid)]["total"] += int(n)
    # [3P] Eigen max / arg-max reducers never select a NaN (0/0 IoU of a degenerate box against padding)
    iou = np.where(np.isnan(iou), -np.inf, iou)
    best_iou, best_gt = iou.max(axis=2), iou.argmax(axis=2)
    visit = np.argsort(-best_iou, axis=1, kind="stable")
    for img in range(best_iou.shape[0]):
        taken = set()
        for t in visit[img]:
            cid = int(det_label[img, t])
            if cid == 0:
                continue
            g = int(best_gt[img, t])
            hit = best_iou[img, t] >= _IOU_TP and cid == int(gt_label[img, g]) and g not in taken
            if hit:
                taken.add(g)
            rec = stats[cid]
            rec["scores"].append(det_score[img, t])
            rec["tp"].append(1 if hit else 0)
            rec["fp"].append(0 if hit else 1)
    return stats


def calculate_ap(recall, precision):
    """11-point interpolated AP: mean over r in {0, 0.1, .., 1} of the best precision at
    recall >= r, 0 where no such point exists (utils/eval_utils.py:56-64)."""
    recall, precision = np.asarray(recall), np.asarray(precision)
    total = 0.0
    for r in _RECALL_POINTS:
        reachable = precision[recall >= r]
        if reachable.size:
            total += float(np.amax(reachable))
    return total / len(_RECALL_POINTS)


def calculate_mAP(stats):
    """Per class: sort by score, cumulate TP/FP, recall = TP / total, precision = TP / (TP + FP);
    returns ``(stats, mean AP)`` with ``recall``/``precision``/``AP`` added to every record
    (utils/eval_utils.py:66-85).  Kept as in the reference: the sort is ``np.argsort(-scores)`` on
    the float32 score array (NumPy's default, unstable sort decides the order of tied scores);
    a class without ground truth divides by zero (NaN recall -> AP 0) and still enters the mean."""
    per_class = []
    for rec in stats.values():
        scores = np.array(rec["scores"], dtype=np.float32) if len(rec["scores"]) else np.array(rec["scores"])
        order = np.argsort(-scores)
        tp = np.cumsum(np.array(rec["tp"])[order])
        fp = np.cumsum(np.array(rec["fp"])[order])
        with np.errstate(divide="ignore", invalid="ignore"):
            rec["recall"] = tp / rec["total"]
            rec["precision"] = tp / (fp + tp)
        rec["AP"] = calculate_ap(rec["recall"], rec["precision"])
        per_class.append(rec["AP"])
    return stats, np.mean(per_class)


def evaluate_predictions(dataset, pred_bboxes, pred_labels, pred_scores, labels, batch_size):
    """Walks ``dataset`` (batches of ``(images, gt_boxes, gt_labels)``) alongside the prediction
    arrays, prints ``mAP: <value>`` and returns the stats (utils/eval_utils.py:87-97)."""
    stats = init_stats(labels)
    for i, (_, gt_boxes, gt_labels) in enumerate(dataset):
        rows = slice(i * batch_size, (i + 1) * batch_size)
        update_stats(pred_bboxes[rows], pred_labels[rows], pred_scores[rows], gt_boxes, gt_labels, stats)
    stats, mean_ap = calculate_mAP(stats)
    print("mAP: {}".format(float(mean_ap)))
    return stats
