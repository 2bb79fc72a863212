"""Cycle timeline of the split-bf16 conv loop (diagnostic; not a test):
    tools/gpu/build_c3prof.sh && python tests/micro/conv3_prof.py
The profiling build (-DSSD_C3_PROF, tests/micro/bin/libssd_hip_c3prof.so) accumulates, per wave of ONE workgroup, the core
clocks between the loop's phase boundaries: [0] MFMA phase of the early group (waves 0-3), [1] split + LDS stores, [2] issue of
the next tile's global loads, [3] MFMA phase of the late group (waves 4-7), [4] barrier wait; printed per K tile."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SSD_HIP_LIBRARY", os.path.join(REPO, "tests", "micro", "bin", "libssd_hip_c3prof.so"))
sys.path[:0] = [REPO, os.path.join(REPO, "tf-ssd_amd")]
import torch
import ssd_hip as h

SHAPES = [("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_4x4_4x2", 256, 128), ("mbv2 head1 (no split-K)", 64, 19, 576, 100, 3, "mfma3_2x7_8x1", 256, 112),
          ("vgg conv3_2", 32, 75, 256, 256, 3, "mfma3_4x4_4x2", 256, 128), ("vgg fc7", 32, 19, 1024, 1024, 1, "mfma3_4x4_4x2", 256, 128)]
lib = h.lib()
names = [lib.ssd_conv_config_name(c).decode() for c in range(lib.ssd_conv_num_configs())]
for name, B, H, Cin, Cout, k, cfg, BM, BN in SHAPES:
    pad = (k - 1) // 2
    d = h.ConvDesc(B, H, H, Cin, Cout, k, k, 1, 1, pad, pad, pad, pad, 0, 0)
    torch.manual_seed(0)
    x = torch.randn(B, H, H, Cin, device=h.device())
    w = torch.randn(k, k, Cin, Cout, device=x.device) / (k * k * Cin) ** 0.5
    packed = torch.empty(lib.ssd_conv_packed_weight_floats(k, k, Cin, Cout), device=x.device)
    h.check(lib.ssd_conv_pack_weights(h.ptr(w), k, k, Cin, Cout, h.ptr(packed), h.stream()), "pack")
    o = torch.empty(B, H, H, Cout, device=x.device)
    ws = torch.empty(1, device=x.device)
    c = names.index(cfg)
    for _ in range(3):
        rc = lib.ssd_conv2d_ex(ctypes.byref(d), h.ptr(x), h.ptr(packed), None, None, None, h.ptr(o), 0, 0, c, 1, h.ptr(ws), h.stream())
        assert rc == 0, lib.ssd_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.ssd_conv2d_ex(ctypes.byref(d), h.ptr(x), h.ptr(packed), None, None, None, h.ptr(o), 0, 0, c, 1, h.ptr(ws), h.stream())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    M = B * H * H
    nb_n = (Cout + BN - 1) // BN
    grid = ((M + BM - 1) // BM) * nb_n
    bid = grid // 3
    m0, n0 = (bid // nb_n) * BM, (bid % nb_n) * BN
    dump = o.reshape(M, Cout)[m0, n0:n0 + 96].cpu().numpy().reshape(8, 12)
    nkt = dump[0, 9]
    rounds = -(-grid // 256)
    print("%s %s: %d K tiles, workgroup %d of %d; kernel %.1f us under the profiling build = %.3f us per K tile at %d rounds of workgroups; "
          "clock units per K tile below" % (name, cfg, nkt, bid, grid, us, us / rounds / max(nkt, 1), rounds))
    print("  wave   mma(early)  [load wait     split  st issue  st drain]  (rest)  load issue   mma(late)   barrier      sum")
    for wv in range(8):
        r = dump[wv, :9] / max(nkt, 1)
        print("  %4d  %10.0f  %10.0f %9.0f %9.0f %9.0f %8.0f  %10.0f  %10.0f  %8.0f  %8.0f" % (wv, r[0], r[5], r[6], r[7], r[8], r[1], r[2], r[3], r[4], r.sum()))
