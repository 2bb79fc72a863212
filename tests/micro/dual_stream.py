"""Experiment: one B=64 predict per step vs two / four concurrent half / quarter batches on
separate streams (each with its own net + captured graph)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "tf-ssd_amd")]
import numpy as np, torch
import ssd_hip
from utils import bbox_utils, train_utils, data_utils
from models.decoder import get_decoder_model
from models.ssd_mobilenet_v2 import get_model

hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
B = 64
x = ssd_hip.to_dev(data_utils.synthetic_images(B, 300, seed=0))
base = get_model(hp, max_batch=B)
w = data_utils.synthetic_weights(base, seed=1)


def run(parts, steps=30):
    n = B // parts
    models, dms, streams, xs = [], [], [], []
    for i in range(parts):
        m = get_model(hp, max_batch=n); m.set_weights(w)
        models.append(m); dms.append(get_decoder_model(m, priors, hp))
        streams.append(torch.cuda.Stream()); xs.append(x[i * n:(i + 1) * n].contiguous())

    def step():
        for dm, s, xi in zip(dms, streams, xs):
            with torch.cuda.stream(s):
                dm(xi)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("parts %d x B=%d: %.3f ms/step  %.0f img/s" % (parts, n, dt * 1e3, B / dt), flush=True)


for parts in (1, 2, 4, 1, 2):
    run(parts)
