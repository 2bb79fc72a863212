"""Diagnostics (not a test): per-layer and end-to-end A/B of the whole-image block kernel (fuse_image)."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import torch
import ssd_hip as h
from utils import train_utils, data_utils, bbox_utils
from models.ssd_mobilenet_v2 import get_model
from models.decoder import get_decoder_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
pri = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
dm = get_decoder_model(m, pri, hp)
x = h.to_dev(data_utils.synthetic_images(B))
for v in (0, 2, 1, 0, 2, 1):
    m.set_option("fuse_image", v)
    for _ in range(10):
        dm(x)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(50):
            dm(x)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 50)
    print("fuse_image=%d  %.4f ms/step  %.0f img/s" % (v, best * 1e3, B / best))
    if v == 1:
        print('   tuned choice:', [l for l in m.get_tuning().splitlines() if ' image ' in l])
        continue
    tot = 0.0
    for rec in m.profile_layers(x, reps=10):
        if rec["ms"] > 0 and rec["name"].startswith("block_") and int(rec["name"].split("_")[1]) >= 7:
            print("   %-24s %-8s %.4f ms  %6.1f TF/s" % (rec["name"], rec["kind"], rec["ms"], rec["flops"] / rec["ms"] / 1e9))
            tot += rec["ms"]
    print("   blocks 7-16 total %.4f ms" % tot)
