"""Throughput-mode retuning of single table lines under three lanes (B = 64): the shipped tables are tuned for one step at a
time (isolated latency per layer); with three batches in flight what counts is the CU TIME of a layer, so tiles that leave
CUs idle but do less padded matrix work / no split-K slab pass may win.  Each variant = the shipped table with some lines
replaced, timed as bench.py times its headline (three replicas, three in-order streams).
usage: GPU_MAX_HW_QUEUES=3 python tests/micro/lanes_table_ab.py"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", ".."), os.path.join(HERE, "..", "..", "tf-ssd_amd"), os.path.join(HERE, "..")]
import numpy as np
import torch
import ssd_hip as h
from utils import bbox_utils, train_utils, data_utils
from models.decoder import get_decoder_model
from models.ssd_mobilenet_v2 import get_model

B = 64
hp = train_utils.get_hyper_params("mobilenet_v2")
hp["total_labels"] = 21
priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
x = h.to_dev(data_utils.synthetic_images(B, 300, seed=0))
base_model = get_model(hp, max_batch=B)
weights = data_utils.synthetic_weights(base_model)
base_model.set_weights(weights)
base_model._ensure(B)
base_table = base_model.get_tuning()

VARIANTS = [
    ("shipped", {}),
    ("head1 dma3_2x4_6x2/s2", {"1_conv_heads": "dma3_2x4_6x2 2"}),
    ("head2 dma3_4x5_4x2/s8", {"2_conv_heads": "dma3_4x5_4x2 8"}),
    ("head2 dma3_4x5_4x2/s10", {"2_conv_heads": "dma3_4x5_4x2 10"}),
    ("heads 1 + 2 on dma3", {"1_conv_heads": "dma3_2x4_6x2 2", "2_conv_heads": "dma3_4x5_4x2 8"}),
    ("heads 1 + 2 + extra1_1 / 1_2 on dma3", {"1_conv_heads": "dma3_2x4_6x2 2", "2_conv_heads": "dma3_4x5_4x2 8",
                                              "extra1_1": "dma3_2x4_4x2 2", "extra1_2": "dma3_2x4_4x2 4"}),
    ("shipped again", {}),
]
if os.environ.get("VARIANTS"):          # "name=layer:config split,layer:config split;name=..." replaces the list above
    VARIANTS = [("shipped", {})] + [(v.split("=", 1)[0], dict(kv.split(":", 1) for kv in v.split("=", 1)[1].split(",")))
                                    for v in os.environ["VARIANTS"].split(";")]


def table_with(repl):
    out = []
    for l in base_table.splitlines():
        name = l.split(" ", 1)[0]
        out.append(name + " " + repl[name] if name in repl else l)
    return "\n".join(out) + "\n"


def timed(dm, steps=40):
    for _ in range(9):
        dm.submit(x, sync_input=False)
    dm.wait()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            dm.submit(x, sync_input=False)
        dm.wait()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    return sorted(ts)[2]


for name, repl in VARIANTS:
    m = get_model(hp, max_batch=B)
    m.set_weights(weights)
    m.set_tuning(table_with(repl))
    m._ensure(B)
    got = {l.split(" ", 1)[0]: l for l in m.get_tuning().splitlines()}
    used = "; ".join(got[k] for k in repl) if repl else ""
    dm = get_decoder_model(m, priors, hp, lanes=3)
    t3 = timed(dm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.predict_on_device(x, priors, hp["variances"])
    torch.cuda.synchronize()
    t1 = (time.perf_counter() - t0) / 20
    print("%-40s three in flight %.4f ms (%.0f img/s) | one at a time %.4f ms   %s" % (name, t3 * 1e3, B / t3, t1 * 1e3, used), flush=True)
    dm.close()
    del dm, m
