"""Experiment: consecutive B=64 steps alternate between two lanes (own net + graph + stream each), so
the latency-bound tail of step n (softmax, decode, NMS, top-k) overlaps the stem of step n+1."""
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


def run(lanes, steps=40):
    dms, streams = [], []
    for i in range(lanes):
        m = get_model(hp, max_batch=B); m.set_weights(w)
        dms.append(get_decoder_model(m, priors, hp)); streams.append(torch.cuda.Stream())

    def step(i):
        with torch.cuda.stream(streams[i % lanes]):
            return dms[i % lanes](x)
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("lanes %d: %.3f ms/step  %.0f img/s" % (lanes, dt * 1e3, B / dt), flush=True)


for lanes in ([int(os.environ['PIPELINE2_LANES'])] if os.environ.get('PIPELINE2_LANES') else (1, 2, 3, 1, 2)):
    run(lanes)
