"""LDS-DMA tiles (dma3_* / dmab_*: pre-split bf16 activation planes, csrc/ssd_convdma.hip) against the register-staged
families (mfma3_* split-bf16, bf16_* one-product, wino_*) on the dense-conv shapes of the two graphs: best tile x split
per family, plus the cost of writing the planes (split_planes_kernel: what a producer without a plane epilogue pays).
usage: python tests/micro/convdma_ab.py [batch] [verbose]      env ONLY=<substring of a shape name>"""
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
    ("mbv2 extra1_1", B, 10, 1280, 256, 1, 1, (0, 0, 0, 0)), ("mbv2 extra1_2", B, 10, 256, 512, 3, 2, (0, 1, 0, 1)),
    ("mbv2 b13 expand", B, 19, 96, 576, 1, 1, (0, 0, 0, 0)),
    ("vgg conv1_2", B // 2, 300, 64, 64, 3, 1, (1, 1, 1, 1)), ("vgg conv2_2", B // 2, 150, 128, 128, 3, 1, (1, 1, 1, 1)),
    ("vgg conv3_2", B // 2, 75, 256, 256, 3, 1, (1, 1, 1, 1)), ("vgg conv4_2", B // 2, 38, 512, 512, 3, 1, (1, 1, 1, 1)),
    ("vgg conv5_2", B // 2, 19, 512, 512, 3, 1, (1, 1, 1, 1)), ("vgg fc7", B // 2, 19, 1024, 1024, 1, 1, (0, 0, 0, 0)),
    ("vgg head1", B // 2, 38, 512, 100, 3, 1, (1, 1, 1, 1)),
]
lib = h.lib()
st = h.stream()
if os.environ.get("ONLY"):
    SHAPES = [s for s in SHAPES if os.environ["ONLY"] in s[0]]
FAMS = ("mfma3", "bf16", "wino", "dma3", "dmab")


def timed(call, n=6):
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(n):
            call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / n)
    return best


for name, b, H, Cin, Cout, k, stride, pads in SHAPES:
    x = torch.randn(b, H, H, Cin, device="cuda")
    w = torch.randn(k, k, Cin, Cout, device="cuda") / (k * k * Cin) ** 0.5
    packed = torch.empty(lib.ssd_conv_packed_weight_floats(k, k, Cin, Cout), device="cuda")
    h.check(lib.ssd_conv_pack_weights(h.ptr(w), k, k, Cin, Cout, h.ptr(packed), st), "pack")
    Ho = lib.ssd_conv_out_size(H, k, stride, 1, pads[0], pads[1])
    out = torch.empty(b, Ho, Ho, Cout, device="cuda")
    n_in = x.numel()
    stride_e = (n_in + 63) // 64 * 64
    planes = {np_: torch.zeros(np_ * stride_e + 64, dtype=torch.int16, device="cuda") for np_ in (1, 3)}
    split_us = {}
    for np_ in (1, 3):
        split_us[np_] = timed(lambda: lib.ssd_split_planes(h.ptr(x), n_in, Cin, np_, h.ptr(planes[np_]), stride_e, st))
    d = h.ConvDesc(b, H, H, Cin, Cout, k, k, stride, 1, pads[0], pads[2], pads[1], pads[3], 2, 0)
    wino = k == 3 and stride == 1
    if wino:
        U = torch.empty(lib.ssd_conv_wino_weight_floats(Cin, Cout), device="cuda")
        h.check(lib.ssd_conv_wino_pack_weights(h.ptr(w), Cin, Cout, h.ptr(U), st), "wino pack")
    wino0 = [i for i in range(lib.ssd_conv_num_configs()) if lib.ssd_conv_config_name(i).startswith(b"wino_")][0]
    ref = {}
    best = {}
    for cfg in range(lib.ssd_conv_num_configs() - 1):
        cname = lib.ssd_conv_config_name(cfg).decode()
        fam = cname.split("_")[0]
        if fam not in FAMS or (fam == "wino" and not wino):
            continue
        for sk in (1, 2, 3, 4, 6, 8):
            ws = torch.empty(sk * b * Ho * Ho * Cout, device="cuda") if sk > 1 else None
            if fam == "wino":
                call = lambda: lib.ssd_conv2d_wino(ctypes.byref(d), h.ptr(x), h.ptr(U), None, None, h.ptr(out), 0, 0, cfg - wino0, sk, h.ptr(ws), st)
            elif fam in ("dma3", "dmab"):
                np_ = 3 if fam == "dma3" else 1
                call = lambda: lib.ssd_conv2d_planes(ctypes.byref(d), h.ptr(planes[np_]), np_, stride_e, h.ptr(packed), None, None, None,
                                                     h.ptr(out), 0, 0, None, 0, cfg, sk, h.ptr(ws), st)
            else:
                call = lambda: lib.ssd_conv2d_ex(ctypes.byref(d), h.ptr(x), h.ptr(packed), None, None, None, h.ptr(out), 0, 0, cfg, sk, h.ptr(ws), st)
            if call() != 0:
                break
            torch.cuda.synchronize()
            prec = "b" if fam in ("bf16", "dmab") else "f"
            if prec not in ref:
                ref[prec] = out.clone()
            err = float((out - ref[prec]).abs().max())
            us = timed(call)
            if fam not in best or us < best[fam][0]:
                best[fam] = (us, cname, sk, err)
            if V:
                print("   %-22s split %d  %8.1f us  err %.1e" % (cname, sk, us, err))
    gf = 2.0 * b * Ho * Ho * Cin * Cout * k * k / 1e9
    print("%-16s %3dx%3dx%3d %4d->%4d k%d s%d %6.1f GF | split pass np3 %.1f us, np1 %.1f us | " % (name, b, H, H, Cin, Cout, k, stride, gf, split_us[3], split_us[1]) +
          "  ".join("%s %.1f us %s/s%d (%.0f TF, err %.0e)" % (f, best[f][0], best[f][1][len(f) + 1:], best[f][2], gf / best[f][0] * 1e3, best[f][3])
                    for f in FAMS if f in best), flush=True)
