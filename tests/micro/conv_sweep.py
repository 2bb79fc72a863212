"""Times every valid (tile config, split-K) of conv_mfma_kernel on a few layer shapes of
SSD300-MobileNetV2 at B=64 (diagnostic; the same measurement ssd_net_finalize's autotune makes).
    python tests/micro/conv_sweep.py"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "tf-ssd_amd")]
import torch  # noqa: E402
import ssd_hip as h  # noqa: E402

SHAPES = [  # name, B, H, Cin, Cout, k
    ("block_7_expand", 64, 19, 64, 384, 1), ("block_11_expand", 64, 19, 96, 576, 1),
    ("block_14_expand", 64, 10, 160, 960, 1), ("Conv_1", 64, 10, 320, 1280, 1), ("extra1_1", 64, 10, 1280, 256, 1),
]
lib = h.lib()
names = [lib.ssd_conv_config_name(c).decode() for c in range(lib.ssd_conv_num_configs())]
for name, B, H, Cin, Cout, k in SHAPES:
    pad = (k - 1) // 2
    d = h.ConvDesc(B, H, H, Cin, Cout, k, k, 1, 1, pad, pad, pad, pad, 2, 0)
    x = torch.randn(B, H, H, Cin, device=h.device())
    packed = torch.randn(lib.ssd_conv_packed_weight_floats(k, k, Cin, Cout), device=x.device)
    sc = torch.ones(Cout, device=x.device)
    o = torch.empty(B, H, H, Cout, device=x.device)
    ws = torch.empty(16 * B * H * H * Cout, device=x.device)
    res = []
    for c, cn in enumerate(names[:-1]):
        for split in (1, 2, 3, 4):
            def run():
                return lib.ssd_conv2d_ex(ctypes.byref(d), h.ptr(x), h.ptr(packed), h.ptr(sc), h.ptr(sc), None, h.ptr(o), 0, 0,
                                         c, split, h.ptr(ws), h.stream())
            if run() != 0:
                continue
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            res.append((best, cn, split))
    res.sort()
    gf = 2.0 * B * H * H * Cin * Cout * k * k / 1e9
    print(name, "GFLOP %.2f:" % gf, "  ".join("%s/s%d %.1fus(%.0fTF)" % (cn, s, t * 1e3, gf / t) for t, cn, s in res[:8]))
