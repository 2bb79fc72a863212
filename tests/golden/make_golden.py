"""Generates tests/golden/*.npz from the CPU oracle (oracle/bbox_oracle.py).

The reference cannot be imported here (every module imports tensorflow, which is not
installed: SURVEY.md 8c), so these fixtures are NOT outputs of the reference; they pin the
oracle restatement (and the SURVEY's known-answer values) against regressions and give the
GPU tests fixed inputs/expected outputs.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402
from oracle import bbox_oracle as bo  # noqa: E402


def main():
    pri = {k: bo.generate_prior_boxes(v, helpers.ASPECT_RATIOS) for k, v in helpers.FMAPS.items()}
    np.savez_compressed(os.path.join(HERE, "priors.npz"), **pri)

    # decode + NMS: small random case, tie case, >200 survivors, zero survivors, degenerate boxes
    p = pri["mobilenet_v2"]
    cases = {}
    d, pr = helpers.decoder_inputs(2, 2268, seed=2)
    cases["rand"] = (d, pr)
    d2, pr2 = helpers.decoder_inputs(1, 2268, seed=5, boost_frac=0.0)
    pr2[...] = 0.02                       # zero survivors: bg dominates every row
    pr2[..., 0] = 0.6
    cases["none"] = (d2, pr2)
    # ties: many anchors with exactly equal scores in two classes
    d3, pr3 = helpers.decoder_inputs(1, 2268, seed=7, boost_frac=0.0)
    pr3[0, ::7, :] = 0.0
    pr3[0, ::7, 3] = 0.75
    pr3[0, ::7, 0] = 0.25
    pr3[0, 3::7, :] = 0.0
    pr3[0, 3::7, 5] = 0.75
    pr3[0, 3::7, 1] = 0.25
    cases["ties"] = (d3, pr3)
    # degenerate: huge negative h/w deltas => near-zero-area boxes; inverted via raw NMS elsewhere
    d4, pr4 = helpers.decoder_inputs(1, 2268, seed=9, boost_frac=0.3)
    d4[0, ::3, 2:] = -400.0
    cases["degenerate"] = (d4, pr4)
    out = {}
    for name, (dd, pp) in cases.items():
        b, l, s, v, i = bo.ssd_decode(p, helpers.VARIANCES, dd, pp, return_indices=True)
        out.update({name + "_deltas": dd, name + "_probs": pp, name + "_boxes": b, name + "_labels": l,
                    name + "_scores": s, name + "_valid": v, name + "_idx": i})
    np.savez_compressed(os.path.join(HERE, "decode_nms.npz"), **out)

    gt, gl = helpers.gt_inputs(4, seed=3)
    hp = helpers.hyper_params()
    dl, oh, lab, mi = bo.calculate_actual_outputs(p, gt, gl, hp, return_indices=True)
    np.savez_compressed(os.path.join(HERE, "match.npz"), gt=gt, gl=gl, deltas=dl, label_idx=lab,
                        match_idx=mi, iou=bo.generate_iou_map(p, gt))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
