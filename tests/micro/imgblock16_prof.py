"""Diagnostics (not a test): clock64 phase profile of the split-bf16 whole-image kernel (mbv2_image16_block_kernel).
Needs the diagnostic build (-DSSD_IMAGE16_PROF on ssd_imgblock.hip, linked as tests/micro/bin/libssd_hip_prof.so):
  SSD_HIP_LIBRARY=tests/micro/bin/libssd_hip_prof.so python tests/micro/imgblock16_prof.py [B] [lanes_hint]"""
import ctypes, os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import ssd_hip as h
from utils import train_utils, data_utils
from models.ssd_mobilenet_v2 import get_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hint = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
if hint > 1: m.set_option("lanes_hint", hint)
data_utils.synthetic_weights(m)
x = h.to_dev(data_utils.synthetic_images(B))
m(x)
out = (ctypes.c_double * 6)()
for k in range(7, 17):
    name = "block_%d_fused" % k
    os.environ.pop("SSD_FUSED_ABLATE", None)
    h.check(h.lib().ssd_net_profile_fused(m._net, name.encode(), B, out), "profile")
    cyc = list(out)
    os.environ["SSD_FUSED_ABLATE"] = "32"
    h.check(h.lib().ssd_net_profile_fused(m._net, name.encode(), B, out), "profile")
    print("%-15s cycles/wave: prologue %6.0f | barrier %6.0f | depthwise %6.0f | project %6.0f | expand+stage %6.0f | epilogue %6.0f | total %6.0f | %.1f us/launch"
          % (name, cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], cyc[5], sum(cyc), out[0]))
