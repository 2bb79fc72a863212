import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tf-ssd_amd"]
import torch, ssd_hip
from utils import bbox_utils, train_utils, data_utils
from models.decoder import get_decoder_model
from models.ssd_mobilenet_v2 import get_model
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
B = 64
x = ssd_hip.to_dev(data_utils.synthetic_images(B, 300, seed=0))
base = get_model(hp, max_batch=B)
w = data_utils.synthetic_weights(base, seed=1)
if os.environ.get("LANES_DBG_NO_OVERLAP"):
    base.set_option("overlap_heads", 0)
dm = get_decoder_model(base, priors, hp, lanes=2)
def timeit(fn, n=40):
    for _ in range(6): fn()
    dm.wait(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dm.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("submit (2 lanes)      %.3f ms" % timeit(lambda: dm.submit(x)), getattr(dm, "lane_calibration", None))
print("__call__ (lane 0)     %.3f ms" % timeit(lambda: dm(x)))
# pipeline2-style with the SAME lane models / streams
ms, ss = dm._lane_models, dm._lane_streams
state = {"i": 0}
def p2():
    i = state["i"] % 2; state["i"] += 1
    with torch.cuda.stream(ss[i]):
        d = dm.decoder
        ms[i].predict_on_device(x, d.prior_boxes, d.variances, max_total=d.max_total_size, iou_threshold=d.iou_threshold, score_threshold=d.score_threshold)
print("direct alternate      %.3f ms" % timeit(p2))
print("tuning equal:", ms[0].get_tuning() == ms[1].get_tuning())
# pipeline2-style: two FRESH models (own autotune), own DecoderModels
dms, sts = [], []
for i in range(2):
    m = get_model(hp, max_batch=B); m.set_weights(w)
    if os.environ.get("LANES_DBG_PIN"): m.set_option("use_graph", 0)
    if os.environ.get("LANES_DBG_NO_OVERLAP"): m.set_option("overlap_heads", 0)
    dms.append(get_decoder_model(m, priors, hp)); sts.append(torch.cuda.Stream())
def p3():
    i = state["i"] % 2; state["i"] += 1
    with torch.cuda.stream(sts[i]):
        dms[i](x)
print("fresh models alternate %.3f ms" % timeit(p3))
print("fresh tuning == base tuning:", dms[0].base_model.get_tuning() == ms[0].get_tuning())
def alt(ma, mb, sa, sb):
    st8 = {"i": 0}
    d = dm.decoder
    def f():
        i = st8["i"] % 2; st8["i"] += 1
        with torch.cuda.stream((sa, sb)[i]):
            (ma, mb)[i].predict_on_device(x, d.prior_boxes, d.variances, max_total=d.max_total_size, iou_threshold=d.iou_threshold, score_threshold=d.score_threshold)
    return timeit(f)
f0, f1 = dms[0].base_model, dms[1].base_model
print("base  + fresh1 (ss0, sts1)  %.3f ms" % alt(ms[0], f1, ss[0], sts[1]))
print("clone + fresh1 (ss1, sts1)  %.3f ms" % alt(ms[1], f1, ss[1], sts[1]))
print("fresh0 + fresh1 (ss0, ss1)  %.3f ms" % alt(f0, f1, ss[0], ss[1]))
print("base + clone (sts0, sts1)   %.3f ms" % alt(ms[0], ms[1], sts[0], sts[1]))
c2 = f0.clone(); c2.set_option("use_graph", 0)
print("fresh0 + clone(fresh0)      %.3f ms" % alt(f0, c2, sts[0], sts[1]))
