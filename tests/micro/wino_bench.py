"""Times every Winograd configuration (and the best direct MFMA config) on a few layer shapes."""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "tf-ssd_amd")]
import numpy as np, torch
import ssd_hip as h
lib = h.lib()
h.device()
SHAPES = {"vgg_conv4_2": (32, 38, 512, 512, 1), "vgg_conv2_2": (32, 150, 128, 128, 1), "head1": (64, 19, 576, 100, 5),
          "head2": (64, 10, 1280, 150, 5)}
only = sys.argv[1:] or list(SHAPES)
for name in only:
    B, H, Cin, Cout, sk = SHAPES[name]
    x = torch.randn((B, H, H, Cin), device="cuda")
    w = torch.randn((3, 3, Cin, Cout), device="cuda") / (3 * Cin ** 0.5)
    U = torch.empty((lib.ssd_conv_wino_weight_floats(Cin, Cout),), device="cuda")
    h.check(lib.ssd_conv_wino_pack_weights(h.ptr(w), Cin, Cout, h.ptr(U), h.stream()), "pack")
    out = torch.empty((B, H, H, Cout), device="cuda")
    ws = torch.empty((max(1, sk * B * H * H * Cout),), device="cuda")
    d = h.ConvDesc(B, H, H, Cin, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0)
    flops = 2.0 * B * H * H * 9 * Cin * Cout
    res = []
    for cfg in range(lib.ssd_conv_wino_num_configs()):
        for s in sorted({1, sk}):
            def run():
                return lib.ssd_conv2d_wino(ctypes.byref(d), h.ptr(x), h.ptr(U), None, None, h.ptr(out), 0, 0, cfg, s, h.ptr(ws), h.stream())
            if run() != 0:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res.append((ms, cfg, s))
    res.sort()
    print(name, " | ".join("cfg%d/s%d %.3f ms %.0f TF/s" % (c, s, ms, flops / ms / 1e9) for ms, c, s in res[:5]), flush=True)
