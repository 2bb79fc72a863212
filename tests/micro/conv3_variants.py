"""Experimental variants of the split-bf16 conv tiles against the production kernel (diagnostic, not a test):
    tools/gpu/build_c3var.sh 1 2 ... && python tests/micro/conv3_variants.py 0 1 2 ...
Each variant library (tests/micro/bin/libssd_hip_c3v<N>.so, -DSSD_C3_VARIANT=N in ssd_conv_mfma.h) runs the shapes
below with a fixed (config, split-K); 0 = the production library."""
import ctypes
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W12 = [("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x7_8x1", 5), ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_1x7_12x1", 5),
       ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_1x7_12x1", 3), ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x4_6x2", 3), ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x4_6x2", 2),
       ("mbv2 head2", 64, 10, 1280, 150, 3, "mfma3_4x5_4x2", 10), ("mbv2 head2", 64, 10, 1280, 150, 3, "mfma3_2x5_6x2", 8), ("mbv2 head2", 64, 10, 1280, 150, 3, "mfma3_2x5_6x2", 4),
       ("mbv2 Conv_1", 64, 10, 320, 1280, 1, "mfma3_4x4_2x4", 1), ("mbv2 Conv_1", 64, 10, 320, 1280, 1, "mfma3_2x4_6x2", 1), ("mbv2 Conv_1", 64, 10, 320, 1280, 1, "mfma3_4x2_3x4", 1),
       ("mbv2 Conv_1", 64, 10, 320, 1280, 1, "mfma3_3x4_4x3", 1),
       ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_4x4_4x2", 1), ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_2x4_6x2", 1), ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_4x2_3x4", 1),
       ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_3x4_4x3", 1),
       ("vgg conv3_2", 32, 75, 256, 256, 3, "mfma3_4x4_4x2", 1), ("vgg conv3_2", 32, 75, 256, 256, 3, "mfma3_2x4_6x2", 1), ("vgg conv3_2", 32, 75, 256, 256, 3, "mfma3_4x2_3x4", 1),
       ("vgg conv2_2", 32, 150, 128, 128, 3, "mfma3_4x4_4x2", 1), ("vgg conv2_2", 32, 150, 128, 128, 3, "mfma3_2x4_6x2", 1), ("vgg conv2_2", 32, 150, 128, 128, 3, "mfma3_4x2_3x4", 1),
       ("vgg conv1_2", 32, 300, 64, 64, 3, "mfma3_4x2_4x2", 1), ("vgg conv1_2", 32, 300, 64, 64, 3, "mfma3_2x4_6x2", 1), ("vgg conv1_2", 32, 300, 64, 64, 3, "mfma3_4x2_3x4", 1),
       ("vgg fc7", 32, 19, 1024, 1024, 1, "mfma3_4x4_4x2", 1), ("vgg fc7", 32, 19, 1024, 1024, 1, "mfma3_2x4_6x2", 1), ("vgg fc7", 32, 19, 1024, 1024, 1, "mfma3_3x4_4x3", 1),
       ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_1x7_16x1", 5), ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x4_8x2", 5), ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x4_8x2", 3),
       ("mbv2 head2", 64, 10, 1280, 150, 3, "mfma3_2x5_8x2", 10), ("mbv2 Conv_1", 64, 10, 320, 1280, 1, "mfma3_2x4_8x2", 1),
       ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_2x4_8x2", 1), ("vgg conv3_2", 32, 75, 256, 256, 3, "mfma3_2x4_8x2", 1), ("vgg conv2_2", 32, 150, 128, 128, 3, "mfma3_2x4_8x2", 1),
       ("vgg conv1_2", 32, 300, 64, 64, 3, "mfma3_2x4_8x2", 1), ("vgg fc7", 32, 19, 1024, 1024, 1, "mfma3_2x4_8x2", 1)]
SHAPES = [("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x7_8x1", 5), ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x7_4x1", 3),
          ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x7_4x1", 2),
          ("mbv2 head2", 64, 10, 1280, 150, 3, "mfma3_4x5_4x2", 10), ("mbv2 head2", 64, 10, 1280, 150, 3, "mfma3_4x5_2x2", 5),
          ("mbv2 head2", 64, 10, 1280, 150, 3, "mfma3_2x5_2x2", 5),
          ("mbv2 Conv_1", 64, 10, 320, 1280, 1, "mfma3_4x4_2x4", 1), ("mbv2 Conv_1", 64, 10, 320, 1280, 1, "mfma3_4x4_2x2", 1),
          ("mbv2 extra1_1", 64, 10, 1280, 256, 1, "mfma3_4x4_2x4", 5), ("mbv2 extra1_1", 64, 10, 1280, 256, 1, "mfma3_4x4_2x2", 3),
          ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_4x4_4x2", 1), ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_4x4_2x2", 1),
          ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_2x4_2x2", 1),
          ("vgg conv3_2", 32, 75, 256, 256, 3, "mfma3_4x4_4x2", 1), ("vgg conv3_2", 32, 75, 256, 256, 3, "mfma3_4x4_2x2", 1),
          ("vgg conv2_2", 32, 150, 128, 128, 3, "mfma3_4x4_4x2", 1), ("vgg conv2_2", 32, 150, 128, 128, 3, "mfma3_4x4_2x2", 1),
          ("vgg conv1_2", 32, 300, 64, 64, 3, "mfma3_4x2_4x2", 1), ("vgg conv1_2", 32, 300, 64, 64, 3, "mfma3_4x2_2x2", 1),
          ("vgg fc7", 32, 19, 1024, 1024, 1, "mfma3_4x4_4x2", 1), ("vgg fc7", 32, 19, 1024, 1024, 1, "mfma3_4x4_2x2", 1)]


def child():
    sys.path[:0] = [REPO, os.path.join(REPO, "tf-ssd_amd")]
    import torch
    import ssd_hip as h
    lib = h.lib()
    names = [lib.ssd_conv_config_name(c).decode() for c in range(lib.ssd_conv_num_configs())]
    out = {}
    for name, B, H, Cin, Cout, k, cfg, split in (W12 if os.environ.get("C3_SHAPES") == "w12" else SHAPES):
        pad = (k - 1) // 2
        d = h.ConvDesc(B, H, H, Cin, Cout, k, k, 1, 1, pad, pad, pad, pad, 0, 0)
        torch.manual_seed(0)
        x = torch.randn(B, H, H, Cin, device=h.device())
        w = torch.randn(k, k, Cin, Cout, device=x.device) / (k * k * Cin) ** 0.5
        packed = torch.empty(lib.ssd_conv_packed_weight_floats(k, k, Cin, Cout), device=x.device)
        h.check(lib.ssd_conv_pack_weights(h.ptr(w), k, k, Cin, Cout, h.ptr(packed), h.stream()), "pack")
        o = torch.empty(B, H, H, Cout, device=x.device)
        ws = torch.empty(max(1, split * B * H * H * Cout), device=x.device)
        c = names.index(cfg)

        def run():
            rc = lib.ssd_conv2d_ex(ctypes.byref(d), h.ptr(x), h.ptr(packed), None, None, None, h.ptr(o), 0, 0, c, split, h.ptr(ws), h.stream())
            assert rc == 0, lib.ssd_last_error()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        out[name + " " + cfg + "/s%d" % split] = (best, float(o.double().abs().sum().item()))
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    if "w12" in sys.argv:           # the 12-wave tiles against the 8-wave ones (production library)
        sys.argv.remove("w12")
        os.environ["C3_SHAPES"] = "w12"
    variants = [int(v) for v in sys.argv[1:]] or [0]
    rows = {}
    for v in variants:
        env = dict(os.environ)
        if v:
            env["SSD_HIP_LIBRARY"] = os.path.join(REPO, "tests", "micro", "bin", "libssd_hip_c3v%d.so" % v)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        if r.returncode:
            print("variant", v, "failed:", r.stderr[-400:])
            continue
        rows[v] = json.loads(r.stdout.strip().splitlines()[-1])
    keys = list(next(iter(rows.values())).keys())
    print("%-40s" % "shape / config" + "".join("%14s" % ("v%d us" % v) for v in rows))
    for k in keys:
        base_sum = rows[variants[0]][k][1]
        print("%-40s" % k + "".join("%10.1f%s" % (rows[v][k][0], " =  " if rows[v][k][1] == base_sum else " != ") for v in rows))
