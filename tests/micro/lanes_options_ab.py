"""Diagnostics (not a test): runtime options under two batches in flight."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import torch
import ssd_hip as h
from utils import train_utils, data_utils, bbox_utils
from models.ssd_mobilenet_v2 import get_model
from models.decoder import get_decoder_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
pri = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
x = h.to_dev(data_utils.synthetic_images(B))
base = get_model(hp, max_batch=B)
w = data_utils.synthetic_weights(base)
for opts in ({}, {"overlap_heads": 0}, {"tail_on_side": 1}, {"fuse_image": 0}, {}):
    m = get_model(hp, max_batch=B); m.set_weights(w)
    for k, v in opts.items():
        m.set_option(k, v)
    dm = get_decoder_model(m, pri, hp, lanes=2)
    for _ in range(8):
        dm.submit(x)
    dm.wait(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40):
            dm.submit(x)
        dm.wait(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40)
    print(opts, "%.4f ms/step  %.0f img/s  pair %s" % (best * 1e3, B / best, dm.lane_calibration["pair"]))
