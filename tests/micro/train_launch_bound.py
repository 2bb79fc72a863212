"""Diagnostics (not a test): is the training step bound by the host's launch rate?  Time to ENQUEUE K steps against the
time until the device has finished them."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, R + "/tf-ssd_amd"]
import torch
import ssd_hip
from utils import bbox_utils, data_utils, train_utils
from ssd_loss import CustomLoss
from models.ssd_mobilenet_v2 import get_model

B = 32
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
model = get_model(hp, max_batch=B)
cl = CustomLoss(hp["neg_pos_ratio"], hp["loc_loss_alpha"])
model.compile(learning_rate=1e-3, loss=[cl.loc_loss_fn, cl.conf_loss_fn])
priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
x = ssd_hip.to_dev(data_utils.synthetic_images(B, hp["img_size"], seed=0))
gt, gl = data_utils.synthetic_gt(B, total_labels=hp["total_labels"], seed=3)
gt, gl = ssd_hip.to_dev(gt), ssd_hip.to_dev(gl, torch.int32)


def step():
    yd, yl = train_utils.calculate_actual_outputs(priors, gt, gl, hp)
    model.plan_gradient_exchange(B)
    loc, conf, g = model.forward_backward(x, yd, yl)
    w = model.exchange_gradients(g)
    model.apply_gradients(g, 1e-3, 1.0 / w)


for _ in range(4):
    step()
torch.cuda.synchronize()
for K in (1, 3, 6):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("K=%d  enqueue %.2f ms/step   finished %.2f ms/step" % (K, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
