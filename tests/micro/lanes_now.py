"""Diagnostics: two-lane submit loop vs the calibration-style free-running loop vs one lane."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import torch
import ssd_hip as h
from utils import train_utils, data_utils, bbox_utils
from models.ssd_mobilenet_v2 import get_model
from models.decoder import get_decoder_model
B = 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
pri = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
if os.environ.get("LANES_NO_SIDE") == "1":
    m.set_option("overlap_heads", 0)          # every lane = ONE in-order stream
edit = os.environ.get("LANE_TABLE_EDIT")        # e.g. "1_conv_heads=1,2_conv_heads=1": split-K overrides for the lanes' table
if edit:
    m._ensure(B)
    base_get = m.get_tuning
    def edited():
        t = base_get()
        for kv in edit.split(","):
            name, split = kv.split("=")
            t = "\n".join((" ".join(l.split()[:2] + [split]) if l.split() and l.split()[0] == name else l) for l in t.splitlines()) + "\n"
        return t
    m.get_tuning = edited
NL = int(os.environ.get('LANES', '2'))
dm = get_decoder_model(m, pri, hp, lanes=NL)
x = h.to_dev(data_utils.synthetic_images(B))
for _ in range(3 * NL):
    dm.submit(x, sync_input=False)
dm.wait(); torch.cuda.synchronize()
print(getattr(dm, 'lane_calibration', None))
def t_submit(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): dm.submit(x, sync_input=False)
    dm.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def t_free(n):
    d = dm.decoder
    ms, ss = dm._lane_models, dm._lane_streams
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(ss[i % NL]):
            ms[i % NL].predict_on_device(x, d.prior_boxes, d.variances, max_total=d.max_total_size, iou_threshold=d.iou_threshold, score_threshold=d.score_threshold)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def t_seq(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): dm(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    print("submit %.4f  free %.4f  seq(base model) %.4f  submit %.4f" % (t_submit(40), t_free(40), t_seq(40), t_submit(40)), flush=True)
