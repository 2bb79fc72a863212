"""Diagnostics: shader-clock phase stamps of the whole-image kernel's second form (ssd_net_profile_fused)."""
import ctypes, os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import ssd_hip as h
from utils import train_utils, data_utils
from models.ssd_mobilenet_v2 import get_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
m.set_option("fuse_image", 2)
data_utils.synthetic_weights(m)
x = h.to_dev(data_utils.synthetic_images(B))
m(x)
for k in [int(a) for a in sys.argv[2:]] or (7, 10):
    out = (ctypes.c_double * 6)()
    h.check(h.lib().ssd_net_profile_fused(m._net, ("block_%d_fused" % k).encode(), B, out), "profile_fused")
    print("block_%d: prologue own %.0f | prologue barrier wait %.0f | loop %.0f | epilogue %.0f  (shader clocks per wave, mean)" % (k, out[0], out[1], out[2], out[3]))
