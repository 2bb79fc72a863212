"""Diagnostics (not a test): 300 pipelined steps on 6 different batches, bitwise against the first pipelined pass over the same batches.
usage: python tests/micro/lanes_stress.py [batch] [lanes]"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import numpy as np, torch
import ssd_hip as h
from utils import train_utils, data_utils, bbox_utils
from models.ssd_mobilenet_v2 import get_model
from models.decoder import get_decoder_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
pri = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
xs = [h.to_dev(data_utils.synthetic_images(B, 300, seed=s)) for s in range(6)]
dm = get_decoder_model(m, pri, hp, lanes=int(sys.argv[2]) if len(sys.argv) > 2 else 3)
# reference = the lanes' own first results (since round 4 a lane replica splits the whole-image blocks over fewer channel groups than
# the base model -- lanes_hint -- so dm(x), which runs the base model, differs in the last bits by design)
first = [dm.submit(x) for x in xs]
dm.wait(); torch.cuda.synchronize()
ref = [[t.clone() for t in o] for o in first]
bad = 0
outs = []
for i in range(300):
    outs.append((i % 6, dm.submit(xs[i % 6])))
    if len(outs) == 30:
        dm.wait(); torch.cuda.synchronize()
        for k, o in outs:
            for a, b in zip(o, ref[k]):
                if not torch.equal(a, b):
                    bad += 1
        outs = []
print("pipelined steps differing from the first pipelined pass:", bad, "pair", dm.lane_calibration)
