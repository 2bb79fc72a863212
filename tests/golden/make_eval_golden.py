"""Generates tests/golden/eval_map.npz by EXECUTING the reference's own pure-NumPy functions
``init_stats`` / ``calculate_ap`` / ``calculate_mAP`` (/root/reference/utils/eval_utils.py:5-17,
56-85) in the build container.  The module itself cannot be imported (its first line is
``import tensorflow as tf`` and TensorFlow is not installable here), so the three function
definitions -- which reference nothing but ``np`` -- are taken from the parsed file (``ast``) and
compiled on their own; ``update_stats``, which does use TF ops, is NOT executed: it is restated in
oracle/eval_oracle.py.  Run from the repo root:  python tests/golden/make_eval_golden.py
Only data (inputs + expected outputs) is written; no reference source travels."""
import ast
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
WANTED = ("init_stats", "calculate_ap", "calculate_mAP")


def load_reference_eval_utils():
    path = os.path.join(REF, "utils", "eval_utils.py")
    tree = ast.parse(open(path).read(), path)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(d.name for d in defs) == sorted(WANTED)
    mod = types.ModuleType("ref_eval_utils")
    mod.np = np
    exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), mod.__dict__)
    return mod


def cases():
    rng = np.random.default_rng(123)
    out = []
    labels = ["bg"] + ["c%d" % i for i in range(1, 7)]
    for case in range(6):
        per = {}
        for cid in range(1, 7):
            n = int(rng.integers(0, 40)) if case != 3 else (0 if cid % 2 else 12)
            total = int(rng.integers(1, 25)) if not (case == 4 and cid == 2) else 0     # class without GT: 0-division
            if case == 5:
                total = 10                                                           # recall hits k/10 exactly
            tp = (rng.random(n) < 0.55).astype(np.int64)
            # never more true positives than ground-truth boxes
            over = np.cumsum(tp) > total
            tp[over] = 0
            scores = rng.random(n).astype(np.float32)
            if case == 2 and n > 4:
                scores[: n // 2] = scores[0]                                         # tied scores
            per[cid] = (total, tp, 1 - tp, scores)
        out.append((labels, per))
    return out


def main():
    ref = load_reference_eval_utils()
    blob = {}
    for ci, (labels, per) in enumerate(cases()):
        stats = ref.init_stats(labels)
        assert sorted(stats) == list(range(1, len(labels)))
        for cid, (total, tp, fp, scores) in per.items():
            stats[cid]["total"] = total
            stats[cid]["tp"] = [int(v) for v in tp]
            stats[cid]["fp"] = [int(v) for v in fp]
            stats[cid]["scores"] = [np.float32(v) for v in scores]
            blob["c%d_k%d_total" % (ci, cid)] = np.int64(total)
            blob["c%d_k%d_tp" % (ci, cid)] = tp
            blob["c%d_k%d_scores" % (ci, cid)] = scores
        with np.errstate(divide="ignore", invalid="ignore"):
            stats, m = ref.calculate_mAP(stats)
        blob["c%d_mAP" % ci] = np.float64(m)
        for cid in per:
            blob["c%d_k%d_AP" % (ci, cid)] = np.float64(stats[cid]["AP"])
            blob["c%d_k%d_recall" % (ci, cid)] = np.asarray(stats[cid]["recall"], np.float64)
            blob["c%d_k%d_precision" % (ci, cid)] = np.asarray(stats[cid]["precision"], np.float64)
    # calculate_ap alone on hand-made curves
    rec = np.array([0.1, 0.3, 0.3, 0.6, 0.7, 1.0])
    pre = np.array([1.0, 0.9, 0.5, 0.8, 0.4, 0.2])
    blob["ap_rec"], blob["ap_pre"], blob["ap_val"] = rec, pre, np.float64(ref.calculate_ap(rec, pre))
    blob["n_cases"] = np.int64(6)
    np.savez_compressed(os.path.join(HERE, "eval_map.npz"), **blob)
    print("wrote eval_map.npz: mAPs", [float(blob["c%d_mAP" % i]) for i in range(6)], "ap", float(blob["ap_val"]))


if __name__ == "__main__":
    main()
