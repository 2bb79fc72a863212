"""N2: VOC07 11-point mAP.  tests/golden/eval_map.npz holds outputs of the
REFERENCE's own ``calculate_ap`` / ``calculate_mAP`` (pure NumPy, executed in the build container
by tests/golden/make_eval_golden.py); both the oracle restatement and the product's host code must
reproduce them bit for bit -- tied scores, a class without predictions, a class without ground
truth (0-division -> NaN recall -> AP 0) and recalls that hit k/10 exactly included."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_map.npz")


def _stats(z, ci, init):
    stats = init(["bg"] + ["c%d" % i for i in range(1, 7)])
    for cid in range(1, 7):
        tp = z["c%d_k%d_tp" % (ci, cid)]
        stats[cid]["total"] = int(z["c%d_k%d_total" % (ci, cid)])
        stats[cid]["tp"] = [int(v) for v in tp]
        stats[cid]["fp"] = [int(1 - v) for v in tp]
        stats[cid]["scores"] = [np.float32(v) for v in z["c%d_k%d_scores" % (ci, cid)]]
    return stats


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_map_matches_reference_outputs(which):
    if which == "oracle":
        from oracle import eval_oracle as eu
    else:
        from utils import eval_utils as eu
    z = np.load(GOLD)
    assert float(eu.calculate_ap(z["ap_rec"], z["ap_pre"])) == float(z["ap_val"])
    for ci in range(int(z["n_cases"])):
        stats, m = eu.calculate_mAP(_stats(z, ci, eu.init_stats))
        assert float(m) == float(z["c%d_mAP" % ci]), ci
        for cid in range(1, 7):
            assert float(stats[cid]["AP"]) == float(z["c%d_k%d_AP" % (ci, cid)]), (ci, cid)
            np.testing.assert_array_equal(np.asarray(stats[cid]["recall"], np.float64), z["c%d_k%d_recall" % (ci, cid)])
            np.testing.assert_array_equal(np.asarray(stats[cid]["precision"], np.float64),
                                          z["c%d_k%d_precision" % (ci, cid)])


def test_oracle_update_stats_hand_case():
    """Visit order is descending best-IoU (not score); label 0 rows are skipped; a GT box is
    matched once; -1 labels are padding (utils/eval_utils.py:23-48)."""
    from oracle import eval_oracle as eo
    gt = np.array([[[0.1, 0.1, 0.5, 0.5], [0.6, 0.6, 0.9, 0.9], [0, 0, 0, 0]]], np.float32)
    gl = np.array([[1, 2, -1]], np.int32)
    pb = np.array([[[0.1, 0.1, 0.5, 0.5],        # exact hit on gt0, label 1      -> TP (visited first, IoU 1.0)
                    [0.12, 0.1, 0.5, 0.5],       # second hit on gt0, label 1     -> FP (gt0 taken)
                    [0.6, 0.6, 0.9, 0.9],        # exact hit on gt1 but label 1   -> FP (label mismatch)
                    [0.61, 0.6, 0.9, 0.9],       # hit on gt1, label 2            -> TP
                    [0.0, 0.0, 0.05, 0.05],      # no overlap, label 2            -> FP
                    [0, 0, 0, 0]]], np.float32)  # padding row, label 0           -> skipped
    pl = np.array([[1, 1, 1, 2, 2, 0]], np.float32)
    ps = np.array([[0.6, 0.9, 0.8, 0.7, 0.95, 0.0]], np.float32)
    st = eo.update_stats(pb, pl, ps, gt, gl, eo.init_stats(["bg", "a", "b"]))
    assert st[1]["total"] == 1 and st[2]["total"] == 1
    # class 1 in visit order: row 0 (IoU 1), row 2 (IoU 1, after row 0 by index), row 1
    assert st[1]["tp"] == [1, 0, 0] and st[1]["fp"] == [0, 1, 1]
    assert [float(v) for v in st[1]["scores"]] == [np.float32(0.6), np.float32(0.8), np.float32(0.9)]
    assert st[2]["tp"] == [1, 0] and st[2]["fp"] == [0, 1]
    st, m = eo.calculate_mAP(st)
    # class 1 sorted by score: FP(.9) FP(.8) TP(.6) -> precision 1/3 at recall 1 -> AP 1/3; class 2: FP(.95) TP(.7) -> 1/2
    assert abs(st[1]["AP"] - 1 / 3) < 1e-12 and abs(st[2]["AP"] - 0.5) < 1e-12 and abs(m - 5 / 12) < 1e-12
