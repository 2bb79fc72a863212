"""hipGraph capture / replay of the real step under a limited number of hardware queues (VERDICT r4 #6b): which ingredient
crashes?  Each case runs in its own process (a segfault must not end the survey): queues x {forked side streams, one in-order
stream} x {capture on the caller's stream, on a lane's native stream}.   usage: python tests/micro/graph_queues_net.py"""
import os
import subprocess
import sys

CASE = r'''
import os, sys
sys.path[:0] = [%(repo)r, %(pkg)r, %(tests)r]
import numpy as np, torch
import ssd_hip as h
import helpers
from models.ssd_mobilenet_v2 import get_model
from utils import bbox_utils
hp = helpers.hyper_params("mobilenet_v2")
m = get_model(hp, max_batch=4)
m.set_weights(helpers.synthetic_weights("mobilenet_v2", hp))
m.set_option("overlap_heads", %(overlap)d)
m.set_option("use_graph", 1)
pri = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
x = h.to_dev(helpers.images(4, 300, seed=1))
st = h.new_stream() if %(native)d else torch.cuda.current_stream()
outs = []
with torch.cuda.stream(st):
    for i in range(4):            # eager, capture, replay, replay
        b, l, s, v = m.predict_on_device(x, pri, hp["variances"])
        torch.cuda.synchronize()
        outs.append(s.cpu().numpy())
assert all(np.array_equal(outs[0], o) for o in outs[1:])
print("ok graphs=%%s" %% m.tuning_info.get("launch") if hasattr(m, "tuning_info") else "ok")
'''
repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("2", "3", "4"):
    for overlap in (1, 0):
        for native in (0, 1):
            code = CASE % dict(repo=repo, pkg=os.path.join(repo, "tf-ssd_amd"), tests=os.path.join(repo, "tests"), overlap=overlap, native=native)
            env = dict(os.environ, GPU_MAX_HW_QUEUES=q, SSD_HIP_GRAPH_FORCE="1", SSD_HIP_WARN_STALE_TABLE="0")
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            tail = (r.stdout.strip().splitlines() or [""])[-1][:80] if r.returncode == 0 else (r.stderr.strip().splitlines() or [""])[-1][:160]
            print("queues %s  side streams %d  native stream %d  -> rc %d  %s" % (q, overlap, native, r.returncode, tail), flush=True)
