"""Diagnostics script (not a test): decoder-only microbench: decode + class mask +
per-class NMS + top-200 on synthetic deltas/probs for N in {2268, 8732, 24564} anchors."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/tf-ssd_amd", os.path.dirname(os.path.abspath(__file__))]
import numpy as np
import torch
import helpers
import ssd_hip as h
from models.decoder import SSDDecoder

for B, N in ((64, 2268), (32, 8732), (16, 24564), (128, 24564)):
    rng = np.random.default_rng(2)
    c = rng.uniform(0.05, 0.95, (N, 2)); sz = rng.uniform(0.02, 0.4, (N, 2))
    pri = np.clip(np.concatenate([c - sz / 2, c + sz / 2], -1), 0, 1).astype(np.float32)
    d, p = helpers.decoder_inputs(B, N, 21, seed=2)
    dec = SSDDecoder(pri, helpers.VARIANCES)
    dd, pp = h.to_dev(d), h.to_dev(p)
    for _ in range(3):
        dec([dd, pp])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        dec([dd, pp])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    algo = B * (N * 25 * 4 + 200 * 24) + N * 16
    print("B=%3d N=%5d: %.3f ms/batch  %.0f img/s  algorithmic %.1f MB -> %.0f GB/s  (mean valid %.0f)" % (
        B, N, dt * 1e3, B / dt, algo / 1e6, algo / dt / 1e9, float(dec.last_valid_detections.float().mean())))
