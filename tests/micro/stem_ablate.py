import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd"]
import ssd_hip as h
from utils import train_utils, data_utils
from models.ssd_mobilenet_v2 import get_model
B = 64
hp = dict(train_utils.get_hyper_params("mobilenet_v2")); hp["total_labels"] = 21
m = get_model(hp, max_batch=B)
data_utils.synthetic_weights(m)
x = h.to_dev(data_utils.synthetic_images(B))
for rec in m.profile_layers(x, reps=20):
    if rec["name"] in ("stem_fused", "block_1_fused", "block_2_fused"):
        print("SSD_STEM_ABLATE=%s %s %.4f ms" % (os.environ.get("SSD_STEM_ABLATE", "0"), rec["name"], rec["ms"]))
