"""Diagnostics (not a test): row-band kernel (fuse_band 1) vs the 8x8-tile kernel (0) for MobileNetV2 blocks 1-6:
per-layer times, end-to-end step, and the difference of the network outputs between the two modes."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import numpy as np
import torch
import ssd_hip as h
from utils import train_utils, data_utils, bbox_utils
from models.ssd_mobilenet_v2 import get_model
from models.decoder import get_decoder_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 300        # 512: the BASELINE configs[4] graph
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
if S == 512:
    hp["img_size"] = 512
    hp["feature_map_shapes"] = [32, 16, 8, 4, 2, 1]
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
pri = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
dm = get_decoder_model(m, pri, hp)
x = h.to_dev(data_utils.synthetic_images(B, S) if S != 300 else data_utils.synthetic_images(B))
outs = {}
for v in (0, 1, 2, 1, 2):
    m.set_option("fuse_band", v)
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
    print("fuse_band=%d  %.4f ms/step  %.0f img/s" % (v, best * 1e3, B / best), flush=True)
    d, p = m(x)
    outs[v] = (d.cpu().numpy(), p.cpu().numpy(), m.fetch_activation("block_6_project_BN") if False else None)
    tot = 0.0
    for rec in m.profile_layers(x, reps=10):
        if rec["ms"] > 0 and rec["kind"] == "fused" and (rec["name"].startswith("stem") or (rec["name"].split("_")[0] == "block" and int(rec["name"].split("_")[1]) <= 6)):
            print("   %-24s %.4f ms  %6.1f TF/s" % (rec["name"], rec["ms"], rec["flops"] / rec["ms"] / 1e9))
            tot += rec["ms"]
    print("   stem + blocks 1-6 total %.4f ms" % tot, flush=True)
print("max |d probs| band vs tile: %.3e   max |d deltas|: %.3e" % (np.abs(outs[0][1] - outs[1][1]).max(), np.abs(outs[0][0] - outs[1][0]).max()))
print("max |d probs| split-bf16 band vs fp32 band: %.3e   max |d deltas|: %.3e" % (np.abs(outs[2][1] - outs[1][1]).max(), np.abs(outs[2][0] - outs[1][0]).max()))
