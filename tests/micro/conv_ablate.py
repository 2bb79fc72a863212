"""Phase ablation of conv_mfma_kernel (diagnostic; not a test).  Each variant library
(tf-ssd_amd/csrc/build_ablate.sh) drops one phase of the main loop; timing every variant on the
same layer shapes shows which phase bounds the kernel.

    bash tf-ssd_amd/csrc/build_ablate.sh 0 1 3 11 4 20 12 && python tests/micro/conv_ablate.py
"""
import ctypes
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = [(0, "full"), (1, "no global loads"), (3, "no loads, no LDS stores"),
            (11, "no loads/stores/barriers (ds_read+MFMA)"), (4, "no MFMA/frag reads (staging only)"),
            (12, "no MFMA, no barrier"), (32, "loads issued, stores independent of them"), (64, "full, LDS-only raw barrier"), (256, "1x1: predicated 64-bit W loads"), (128, "address math only, no load issued"),
            (96, "independent stores + raw barrier"), (20, "frag reads kept, no MFMA")]
# name, B, H, Cin, Cout, k, cfg name, split
SHAPES = [
    ("2_conv_heads", 64, 10, 1280, 150, 3, "mfma_2x5_2x2_k32", 16),
    ("1_conv_heads", 64, 19, 576, 100, 3, "mfma_1x7_4x1_k32", 4),
    ("Conv_1", 64, 10, 320, 1280, 1, "mfma_4x4_2x2_k32", 1),
    ("block_14_project", 64, 10, 960, 160, 1, "mfma_1x1_4x1_k32", 1),
    ("block_11_project", 64, 19, 576, 96, 1, "mfma_2x2_2x2_k32", 1),
    ("block_11_expand", 64, 19, 96, 576, 1, "mfma_1x2_2x2_k32", 1),
]


def child():
    sys.path[:0] = [REPO, os.path.join(REPO, "tf-ssd_amd")]
    import numpy as np
    import torch
    import ssd_hip as h
    lib = h.lib()
    names = [lib.ssd_conv_config_name(c).decode() for c in range(lib.ssd_conv_num_configs())]
    out = {}
    for name, B, H, Cin, Cout, k, cfg, split in SHAPES:
        pad = (k - 1) // 2
        d = h.ConvDesc(B, H, H, Cin, Cout, k, k, 1, 1, pad, pad, pad, pad, 0, 0)
        x = torch.randn(B, H, H, Cin, device=h.device())
        packed = torch.randn(lib.ssd_conv_packed_weight_floats(k, k, Cin, Cout), device=x.device)
        o = torch.empty(B, H, H, Cout, device=x.device)
        ws = torch.empty(max(1, split * B * H * H * Cout), device=x.device)
        c = names.index(cfg)

        def run():
            rc = lib.ssd_conv2d_ex(ctypes.byref(d), h.ptr(x), h.ptr(packed), None, None, None, h.ptr(o), 0, 0, c, split,
                                   h.ptr(ws), h.stream())
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
            best = min(best, e0.elapsed_time(e1) / 10)
        out[name] = best
    print(json.dumps(out))


def main():
    rows = {}
    only = os.environ.get("ABLATE_ONLY")
    for bits, label in VARIANTS:
        if only and str(bits) not in only.split(","):
            continue
        path = os.path.join(REPO, "tf-ssd_amd", "csrc", "build", "ablate", "libssd_hip_ab%d.so" % bits)
        if not os.path.exists(path):
            continue
        env = dict(os.environ, SSD_HIP_LIBRARY=path)
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        if r.returncode:
            print(label, "FAILED", r.stderr[-400:])
            continue
        rows[label] = json.loads(r.stdout.strip().splitlines()[-1])
    names = [s[0] for s in SHAPES]
    print("%-42s" % "variant (ms)" + "".join("%18s" % n for n in names))
    for label, v in rows.items():
        print("%-42s" % label + "".join("%18.4f" % v[n] for n in names))


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
