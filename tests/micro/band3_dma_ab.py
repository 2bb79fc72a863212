"""Diagnostics (round 6): the split row-band kernel with its weight fragments staged by LDS-DMA (option image_v2 = the "second forms")
against the register-prefetch form, per layer for the stem and blocks 1-6; SSD_BAND3_SPLIT12=1|2 additionally moves blocks 1-2 from
the fp32-MFMA band kernel to the split form (only possible with the LDS-staged weights).  One process per setting (the env switch is
read once): python tests/micro/band3_dma_ab.py <image_v2 0|1>"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import numpy as np
import ssd_hip as h
from utils import train_utils, data_utils
from models.ssd_mobilenet_v2 import get_model
v2 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
m.set_option("image_v2", v2)
data_utils.synthetic_weights(m)
x = h.to_dev(data_utils.synthetic_images(B))
d, p = m(x)
np.save("/tmp/band3_ab_v%d_s%s.npy" % (v2, os.environ.get("SSD_BAND3_SPLIT12", "0")), m.fetch_activation("block_6_out"))
best = {}
for rep in range(3):
    for rec in m.profile_layers(x, reps=20):
        if rec["ms"] > 0 and rec["kind"] == "fused" and rec["flops"] > 0:
            best[rec["name"]] = (min(best.get(rec["name"], (1e9,))[0], rec["ms"]), rec["config"])
tot = 0.0
print("image_v2 %d  SSD_BAND3_SPLIT12 %s" % (v2, os.environ.get("SSD_BAND3_SPLIT12", "0")))
for n, (ms, cfg) in best.items():
    k = n.split("_")
    if n.startswith("stem") or (k[0] == "block" and int(k[1]) <= 6):
        print("   %-16s %-10s %.4f ms" % (n, cfg, ms)); tot += ms
print("   stem + blocks 1-6 total %.4f ms" % tot)
ref = "/tmp/band3_ab_v0_s0.npy"
if os.path.exists(ref):
    a, b = np.load(ref), m.fetch_activation("block_6_out")
    print("   block_6_out vs the register-prefetch form: max |d| %.3e (max |ref| %.2f)" % (float(np.abs(a - b).max()), float(np.abs(a).max())))
