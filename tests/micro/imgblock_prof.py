"""Diagnostics (not a test): whole-image block kernel -- clock64 phase profile and timings with phases removed."""
import ctypes, os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import ssd_hip as h
from utils import train_utils, data_utils
from models.ssd_mobilenet_v2 import get_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
x = h.to_dev(data_utils.synthetic_images(B))
m(x)
out = (ctypes.c_double * 6)()
for name in ("block_7_fused", "block_11_fused", "block_14_fused", "block_16_fused"):
    os.environ.pop("SSD_FUSED_ABLATE", None)
    h.check(h.lib().ssd_net_profile_fused(m._net, name.encode(), B, out), "profile")
    tot = sum(out)
    print(name, "cycles/wave: prologue %.0f | barrier %.0f | depthwise %.0f | project %.0f | expand+stage %.0f | epilogue %.0f | total %.0f"
          % (out[0], out[1], out[2], out[3], out[4], out[5], tot))
    res = []
    for ab in (0, 1, 2, 4, 7, 8, 16, 23):
        os.environ["SSD_FUSED_ABLATE"] = str(ab) if ab else "32"
        h.check(h.lib().ssd_net_profile_fused(m._net, name.encode(), B, out), "profile")
        res.append("%d: %.1f us" % (ab, out[0]))
    print("   ablate (1 expand MFMA, 2 depthwise, 4 project MFMA, 8 no combine, 16 no slab/ticket):", " | ".join(res))
