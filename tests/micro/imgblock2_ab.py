"""Diagnostics (not a test): the whole-image kernel's second form (ssd_imgblock2.hip, option image_v2) against the first:
bitwise comparison of every block output and per-block times.  SSD_IMAGE2_VARIANT=n picks the n-th configuration of a shape.
usage: python tests/micro/imgblock2_ab.py [B] [precision]"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import numpy as np
import ssd_hip as h
from utils import train_utils, data_utils
from models.ssd_mobilenet_v2 import get_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B, precision=prec)
m.set_option("fuse_image", 2)           # the whole-image kernel wherever it applies
data_utils.synthetic_weights(m)
if os.environ.get("SSD_AB_FORCE_SPLIT", "1") == "1" and prec == "fp32":
    # the split-bf16 form of the whole-image kernel for every block it fits (the shipped table keeps the fp32-MFMA form for 11, 12, 16)
    m._ensure(B)
    t = "\n".join((l.rsplit(" ", 1)[0] + " 2") if " image " in l else l for l in m.get_tuning().splitlines()) + "\n"
    m.set_tuning(t)
x = h.to_dev(data_utils.synthetic_images(B))
names = ["block_%d_out" % k for k in range(7, 17)] + ["block_13_expand_relu"]
res = {}
for v in (0, 1):
    m.set_option("image_v2", v)
    m(x)
    outs = {}
    for n in names:
        try:
            outs[n] = m.fetch_activation(n).copy()
        except ValueError:
            pass
    best = {}
    for rep in range(3):
        for rec in m.profile_layers(x, reps=20):
            if rec["ms"] > 0 and rec["kind"] == "fused" and rec["flops"] > 0:
                best[rec["name"]] = min(best.get(rec["name"], 1e9), rec["ms"])
    res[v] = (outs, best)
print("variant", os.environ.get("SSD_IMAGE2_VARIANT", "0"), "B", B, prec)
for n in sorted(res[0][0]):
    a, b = res[0][0][n], res[1][0][n]
    print("  %-22s bitwise %s   max|d| %.3g" % (n, bool((a.view(np.uint32) == b.view(np.uint32)).all()), float(np.abs(a - b).max())))
tot = [0.0, 0.0]
for n in res[0][1]:
    if not n.startswith("block_") or int(n.split("_")[1]) < 7:
        continue
    t0, t1 = res[0][1][n], res[1][1][n]
    tot[0] += t0; tot[1] += t1
    print("  %-18s v1 %.4f ms   v2 %.4f ms   %+.1f %%" % (n, t0, t1, 100 * (t1 - t0) / t0))
print("  blocks 7-16 total  v1 %.4f   v2 %.4f ms" % tuple(tot))
