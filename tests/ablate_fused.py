"""Diagnostics script (not a test): fused block kernel time with phases removed (results invalid)."""
import ctypes, os, sys, subprocess
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
out = (ctypes.c_double * 6)()
for name in ("block_1_fused", "block_2_fused", "block_3_fused"):
    res = []
    for ab in (0, 1, 2, 4, 8, 15, 31):
        os.environ["SSD_FUSED_ABLATE"] = str(ab)
        h.check(h.lib().ssd_net_profile_fused(m._net, name.encode(), B, out), "profile")
        res.append("%d: %.1f us" % (ab, out[0]))
    print(name, os.environ.get("SSD_FUSED_8WAVE", "1"), " | ".join(res))
