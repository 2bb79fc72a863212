"""Diagnostics (not a test): ms per step of two lanes for EVERY pair among N candidate streams."""
import os, sys, time, itertools
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import torch
import ssd_hip as h
from utils import train_utils, data_utils, bbox_utils
from models.ssd_mobilenet_v2 import get_model
from models.decoder import get_decoder_model
B = 64
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
pri = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
x = h.to_dev(data_utils.synthetic_images(B))
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
dm = get_decoder_model(m, pri, hp, lanes=2)
dm.submit(x); dm.wait(); torch.cuda.synchronize()
models = dm._lane_models
d = dm.decoder
streams = [torch.cuda.Stream() for _ in range(N)]
def trial(sa, sb, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream((sa, sb)[i % 2]):
            models[i % 2].predict_on_device(x, d.prior_boxes, d.variances, max_total=d.max_total_size, iou_threshold=d.iou_threshold, score_threshold=d.score_threshold)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
res = []
for a, b in itertools.combinations(range(N), 2):
    trial(streams[a], streams[b], 2)
    res.append((trial(streams[a], streams[b], 16), a, b))
res.sort()
print("best  :", ["%.3f (%d,%d)" % r for r in res[:8]])
print("worst :", ["%.3f (%d,%d)" % r for r in res[-5:]])
import collections
hist = collections.Counter(round(r[0], 1) for r in res)
print("histogram (ms -> pairs):", sorted(hist.items()))
print("calibrated by DecoderModel:", dm.lane_calibration)
