"""Phase ablation of the split-bf16 conv kernel (diagnostic; not a test):
    bash tf-ssd_amd/csrc/build_ablate3.sh 0 1 2 4 8 16 24 && python tests/micro/conv3_ablate.py
Each variant library drops phases of the main loop (SSD_C3_ABLATE bits: 1 MFMAs, 2 fragment reads + MFMAs, 4 the
fp32 -> 3 x bf16 split, 8 global loads, 16 LDS stores)."""
import ctypes
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = [(0, "full"), (1, "no MFMA"), (2, "no frag reads, no MFMA"), (4, "no split"), (8, "no global loads"),
            (16, "no LDS stores"), (24, "no loads, no LDS stores"), (26, "barriers + address math only")]
SHAPES = [("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_4x4_4x2", 1), ("vgg conv4_2", 32, 38, 512, 512, 3, "mfma3_2x4_2x2", 1),
          ("mbv2 head1", 64, 19, 576, 100, 3, "mfma3_2x7_8x1", 8), ("mbv2 Conv_1", 64, 10, 320, 1280, 1, "mfma3_4x4_2x4", 1)]


def child():
    sys.path[:0] = [REPO, os.path.join(REPO, "tf-ssd_amd")]
    import torch
    import ssd_hip as h
    lib = h.lib()
    names = [lib.ssd_conv_config_name(c).decode() for c in range(lib.ssd_conv_num_configs())]
    out = {}
    for name, B, H, Cin, Cout, k, cfg, split in SHAPES:
        pad = (k - 1) // 2
        d = h.ConvDesc(B, H, H, Cin, Cout, k, k, 1, 1, pad, pad, pad, pad, 0, 0)
        x = torch.randn(B, H, H, Cin, device=h.device())
        w = torch.randn(k, k, Cin, Cout, device=x.device) / (k * k * Cin) ** 0.5
        packed = torch.empty(lib.ssd_conv_packed_weight_floats(k, k, Cin, Cout), device=x.device)
        h.check(lib.ssd_conv_pack_weights(h.ptr(w), k, k, Cin, Cout, h.ptr(packed), h.stream()), "pack")
        o = torch.empty(B, H, H, Cout, device=x.device)
        ws = torch.empty(max(1, split * B * H * H * Cout), device=x.device)
        c = names.index(cfg)

        def run():
            rc = lib.ssd_conv2d_ex(ctypes.byref(d), h.ptr(x), h.ptr(packed), None, None, None, h.ptr(o), 0, 0, c, split, h.ptr(ws), h.stream())
            assert rc == 0, h.last_error()
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
        out[name + " " + cfg] = best
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    rows = {}
    for bits, label in VARIANTS:
        lib = os.path.join(REPO, "tf-ssd_amd", "csrc", "build", "ablate", "libssd_hip_c3ab%d.so" % bits)
        if not os.path.exists(lib):
            continue
        env = dict(os.environ, SSD_HIP_LIBRARY=lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(label, "FAILED", r.stderr[-300:])
            continue
        rows[label] = json.loads(r.stdout.strip().splitlines()[-1])
    keys = list(next(iter(rows.values())).keys())
    print("%-34s" % "variant" + "".join("%28s" % k for k in keys))
    for label, v in rows.items():
        print("%-34s" % label + "".join("%25.1f us" % v[k] for k in keys))
