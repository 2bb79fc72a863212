"""Diagnostics (not a test): the SSD tail (extras 2-4, head levels 3-6) after a fresh on-device autotune --
chosen configuration and time per layer, end-to-end step one batch at a time."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
os.environ["SSD_HIP_IGNORE_SHIPPED"] = "1"
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
for _ in range(10):
    dm(x)
torch.cuda.synchronize()
print(m.tuning_info)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50):
        dm(x)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 50)
print("%.4f ms/step  %.0f img/s" % (best * 1e3, B / best), flush=True)
tot = 0.0
for rec in m.profile_layers(x, reps=10):
    n = rec["name"]
    if rec["ms"] > 0 and rec["kind"] == "conv" and (n.startswith("extra") or n[0] in "3456" or n in ("Conv_1",)):
        print("   %-16s %-22s %.4f ms  %6.1f TF/s" % (n, rec["config"], rec["ms"], rec["flops"] / rec["ms"] / 1e9))
        if n != "Conv_1" and not n.startswith("extra1"):
            tot += rec["ms"]
print("   tail (extras 2-4 + heads 3-6) total %.4f ms" % tot, flush=True)
