"""Diagnostics script (not a test): A/B of runtime options on the bench workload."""
import os, sys, time, itertools
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/tf-ssd_amd"]
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
dm = get_decoder_model(m, pri, hp)
x = h.to_dev(data_utils.synthetic_images(B))
opts = sys.argv[1:] or ["use_graph", "overlap_heads", "fuse_blocks"]
for combo in itertools.product([0, 1], repeat=len(opts)):
    for o, v in zip(opts, combo):
        m.set_option(o, v)
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
    print(dict(zip(opts, combo)), "%.4f ms/step  %.0f img/s" % (best * 1e3, B / best))
