"""Winograd kernels, fp32 MFMA ("wino_*") against split-bf16 ("wino3_*"): every config x channel split on the 3x3
stride-1 shapes of the two graphs.  usage: python tests/micro/wino_ab.py [batch]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tf-ssd_amd"))
import ssd_hip as h

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SHAPES = [  # name, batch, H, W, Cin, Cout
    ("mbv2 head1", B, 19, 19, 576, 100), ("mbv2 head2", B, 10, 10, 1280, 150), ("mbv2 head3", B, 5, 5, 512, 150),
    ("vgg conv1_2", B // 2, 300, 300, 64, 64), ("vgg conv2_2", B // 2, 150, 150, 128, 128), ("vgg conv3_2", B // 2, 75, 75, 256, 256),
    ("vgg conv4_2", B // 2, 38, 38, 512, 512), ("vgg conv5_2", B // 2, 19, 19, 512, 512), ("vgg head1", B // 2, 38, 38, 512, 100),
    ("vgg head2", B // 2, 19, 19, 1024, 150),
]
lib = h.lib()
st = h.stream()
for name, b, H, W, Cin, Cout in SHAPES:
    x = torch.randn(b, H, W, Cin, device="cuda")
    w = torch.randn(3, 3, Cin, Cout, device="cuda") / (9 * Cin) ** 0.5
    U = torch.empty(lib.ssd_conv_wino_weight_floats(Cin, Cout), device="cuda")
    h.check(lib.ssd_conv_wino_pack_weights(h.ptr(w), Cin, Cout, h.ptr(U), st), "pack")
    out = torch.empty(b, H, W, Cout, device="cuda")
    d = h.ConvDesc(b, H, W, Cin, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 2, 0)
    ref = None
    best = {}
    for cfg in range(lib.ssd_conv_wino_num_configs()):
        cname = lib.ssd_conv_wino_config_name(cfg).decode()
        kind = cname.split("_")[0]
        for sk in (1, 2, 3, 4, 6, 8):
            if sk > Cin // 32:
                continue
            ws = torch.empty(sk * b * H * W * Cout, device="cuda") if sk > 1 else None
            args = (ctypes.byref(d), h.ptr(x), h.ptr(U), None, None, h.ptr(out), 0, 0, cfg, sk, h.ptr(ws), st)
            if lib.ssd_conv2d_wino(*args) != 0:
                break
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            err = float((out - ref).abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                lib.ssd_conv2d_wino(*args)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / n
            if kind not in best or us < best[kind][0]:
                best[kind] = (us, cname, sk, err)
            if len(sys.argv) > 2:
                print("   %-14s split %d  %8.1f us  err %.1e" % (cname, sk, us, err))
    gf = 2.0 * b * H * W * Cin * Cout * 9 / 1e9
    print("%-12s %4dx%3dx%3d %4d->%4d  " % (name, b, H, W, Cin, Cout) +
          "   ".join("%s: %7.1f us (%s/s%d, %.0f TF direct-equivalent, |d| %.1e)" % (k, v[0], v[1], v[2], gf / v[0] / 1e3, v[3])
                     for k, v in sorted(best.items())))
