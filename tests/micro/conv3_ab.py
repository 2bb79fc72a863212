"""Implicit-GEMM conv families against each other on the dense-conv shapes of the two graphs: fp32-MFMA tiles ("mfma"),
Winograd ("wino"), in-workgroup K split ("skinny"), split-bf16 tiles ("mfma3").  Best config x split per family.
usage: python tests/micro/conv3_ab.py [batch] [verbose]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tf-ssd_amd"))
import ssd_hip as h

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
V = len(sys.argv) > 2
SHAPES = [  # name, batch, H, Cin, Cout, k, stride, pads (t, b, l, r)
    ("mbv2 Conv_1", B, 10, 320, 1280, 1, 1, (0, 0, 0, 0)), ("mbv2 head1", B, 19, 576, 100, 3, 1, (1, 1, 1, 1)),
    ("mbv2 head2", B, 10, 1280, 150, 3, 1, (1, 1, 1, 1)), ("mbv2 head3", B, 5, 512, 150, 3, 1, (1, 1, 1, 1)),
    ("mbv2 head4", B, 3, 256, 150, 3, 1, (1, 1, 1, 1)), ("mbv2 extra1_1", B, 10, 1280, 256, 1, 1, (0, 0, 0, 0)),
    ("mbv2 extra1_2", B, 10, 256, 512, 3, 2, (0, 1, 0, 1)), ("mbv2 b13 expand", B, 19, 96, 576, 1, 1, (0, 0, 0, 0)),
    ("vgg conv1_2", B // 2, 300, 64, 64, 3, 1, (1, 1, 1, 1)), ("vgg conv2_1", B // 2, 150, 64, 128, 3, 1, (1, 1, 1, 1)),
    ("vgg conv2_2", B // 2, 150, 128, 128, 3, 1, (1, 1, 1, 1)), ("vgg conv3_2", B // 2, 75, 256, 256, 3, 1, (1, 1, 1, 1)),
    ("vgg conv4_2", B // 2, 38, 512, 512, 3, 1, (1, 1, 1, 1)), ("vgg conv5_2", B // 2, 19, 512, 512, 3, 1, (1, 1, 1, 1)),
    ("vgg fc7", B // 2, 19, 1024, 1024, 1, 1, (0, 0, 0, 0)), ("vgg head1", B // 2, 38, 512, 100, 3, 1, (1, 1, 1, 1)),
]
lib = h.lib()
st = h.stream()
if os.environ.get("ONLY"):
    SHAPES = [s for s in SHAPES if os.environ["ONLY"] in s[0]]


def family(name):
    return name.split("_")[0]


for name, b, H, Cin, Cout, k, stride, pads in SHAPES:
    x = torch.randn(b, H, H, Cin, device="cuda")
    w = torch.randn(k, k, Cin, Cout, device="cuda") / (k * k * Cin) ** 0.5
    packed = torch.empty(lib.ssd_conv_packed_weight_floats(k, k, Cin, Cout), device="cuda")
    h.check(lib.ssd_conv_pack_weights(h.ptr(w), k, k, Cin, Cout, h.ptr(packed), st), "pack")
    Ho = lib.ssd_conv_out_size(H, k, stride, 1, pads[0], pads[1])
    out = torch.empty(b, Ho, Ho, Cout, device="cuda")
    d = h.ConvDesc(b, H, H, Cin, Cout, k, k, stride, 1, pads[0], pads[2], pads[1], pads[3], 2, 0)
    ref = None
    best = {}
    wino = k == 3 and stride == 1
    if wino:
        U = torch.empty(lib.ssd_conv_wino_weight_floats(Cin, Cout), device="cuda")
        h.check(lib.ssd_conv_wino_pack_weights(h.ptr(w), Cin, Cout, h.ptr(U), st), "wino pack")
    runs = [("ex", c) for c in range(lib.ssd_conv_num_configs() - 1)]
    for kind, cfg in runs:
        cname = lib.ssd_conv_config_name(cfg).decode()
        fam = family(cname)
        if fam == "wino" and not wino:
            continue
        for sk in (1, 2, 3, 4, 6, 8):
            ws = torch.empty(sk * b * Ho * Ho * Cout, device="cuda") if sk > 1 else None
            if fam == "wino":
                wcfg = cfg - [i for i in range(lib.ssd_conv_num_configs()) if family(lib.ssd_conv_config_name(i).decode()) == "wino"][0]
                call = lambda: lib.ssd_conv2d_wino(ctypes.byref(d), h.ptr(x), h.ptr(U), None, None, h.ptr(out), 0, 0, wcfg, sk, h.ptr(ws), st)
            else:
                call = lambda: lib.ssd_conv2d_ex(ctypes.byref(d), h.ptr(x), h.ptr(packed), None, None, None, h.ptr(out), 0, 0, cfg, sk, h.ptr(ws), st)
            if call() != 0:
                break
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            err = float((out - ref).abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / n
            if fam not in best or us < best[fam][0]:
                best[fam] = (us, cname, sk, err)
            if V:
                print("   %-22s split %d  %8.1f us  err %.1e" % (cname, sk, us, err))
    gf = 2.0 * b * Ho * Ho * Cin * Cout * k * k / 1e9
    print("%-16s %3dx%3dx%3d %4d->%4d k%d s%d  %6.1f GF | " % (name, b, H, H, Cin, Cout, k, stride, gf) +
          "  ".join("%s %6.1f us %s/s%d (%.0f TF)" % (f, v[0], v[1][len(f) + 1:], v[2], gf / v[0] * 1e-3 * 1e3 / 1e3 * 1e3) for f, v in sorted(best.items())),
          flush=True)
