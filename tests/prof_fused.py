"""Diagnostics script (not a test): per-phase cycles of the fused MobileNetV2 block kernel."""
import ctypes
import sys
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/tf-ssd_amd"]
import ssd_hip as h
from utils import train_utils, data_utils
from models.ssd_mobilenet_v2 import get_model

B = 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
x = h.to_dev(data_utils.synthetic_images(B))
m(x)
names = ["prologue", "barrier", "depthwise", "project", "expand", "epilogue"]   # row-band kernel (fuse_band 1); the 8x8-tile kernel: prologue, expand, depthwise, project, wstage, epilogue
for k in (1, 2, 3, 4, 6):
    out = (ctypes.c_double * 6)()
    h.check(h.lib().ssd_net_profile_fused(m._net, ("block_%d_fused" % k).encode(), B, out), "profile_fused")
    tot = sum(out)
    print("block_%d_fused: total %.0f cycles/wave | " % (k, tot) + "  ".join("%s %.0f (%.0f%%)" % (n, c, 100 * c / tot) for n, c in zip(names, out)))
