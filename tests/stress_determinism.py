"""Diagnostics script (not a test): repeated forwards must be bitwise identical."""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/tf-ssd_amd"]
import torch
import ssd_hip as h
from utils import train_utils, data_utils
from models.ssd_mobilenet_v2 import get_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
x = h.to_dev(data_utils.synthetic_images(B))
for opts in ({"fuse_blocks": 0}, {"fuse_blocks": 1, "use_graph": 0, "overlap_heads": 0}, {"fuse_blocks": 1, "use_graph": 1, "overlap_heads": 1},
             {"fuse_image": 2, "image_ticket": 1, "use_graph": 1}, {"fuse_image": 2, "image_ticket": 0, "tail_on_side": 1}):
    for k, v in opts.items():
        m.set_option(k, v)
    d0, p0 = m(x)
    d0, p0 = d0.clone(), p0.clone()
    bad = 0
    for i in range(40):
        d, p = m(x)
        if not (torch.equal(d, d0) and torch.equal(p, p0)):
            bad += 1
            print("  mismatch iter", i, float((d - d0).abs().max()), float((p - p0).abs().max()))
    print(opts, "mismatches:", bad)
