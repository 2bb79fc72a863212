#!/usr/bin/env python
"""Regenerate the kernel-choice tables shipped in tf-ssd_amd/tables/ (tuning.py).  Runs on an MI355X:

    SSD_HIP_IGNORE_SHIPPED=1 python tools/make_tuning_tables.py --out gpurun_out/tables [--shapes ...]

For every (backbone, image size, batch) it finalizes a net with seeded synthetic weights -- the on-device
autotune times every valid (tile configuration, split-K) of every conv layer, races the whole-image block
kernel against the layer kernels and graph replay against direct launches -- `--repeats` times, keeps per
line the choice the majority of the runs made (ties: the first run), and writes `<key>.tune` with a
provenance header.  Copy the files into tf-ssd_amd/tables/ and commit them: from then on finalize times
nothing for these shapes and every process runs the same kernels (bit-identical results).

    python tools/make_tuning_tables.py --adopt DIR     # CPU: copy SSD_HIP_TUNE_CACHE-style files from DIR
"""
import argparse
import collections
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tf-ssd_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# BASELINE.json configs (C2 B=64, C3 VGG B=32, C4 per-GPU B=32, C5 512^2 B=16), their neighbours, and the
# batch sizes the GPU tests / smoke / bench's CPU-sample legs finalize at
DEFAULT_SHAPES = [("mobilenet_v2", 300, b, "fp32") for b in (1, 2, 3, 4, 5, 6, 8, 16, 24, 32, 64, 128, 232, 256)] + \
                 [("vgg16", 300, b, "fp32") for b in (1, 2, 3, 6, 8, 16, 32)] + \
                 [("mobilenet_v2", 512, b, "fp32") for b in (1, 16)] + \
                 [("mobilenet_v2", 300, b, "bf16") for b in (4, 64)] + [("mobilenet_v2", 512, 16, "bf16")] + \
                 [("vgg16", 300, b, "bf16") for b in (8, 32)]      # the bf16 mode (BASELINE configs[3] / [4]): its tests' and bench shapes


def majority(tables):
    lines = collections.OrderedDict()
    for t in tables:
        for l in t.splitlines():
            name = l.split(" ", 1)[0]
            lines.setdefault(name, []).append(l)
    out = []
    for name, ls in lines.items():
        c = collections.Counter(ls)
        best = max(c.values())
        out.append(next(l for l in ls if c[l] == best))
    return "\n".join(out) + "\n"


def generate(args):
    import torch
    import helpers
    import ssd_hip
    import tuning
    from models._net import SSDModel
    os.makedirs(args.out, exist_ok=True)
    build = ssd_hip.lib().ssd_build_id().decode()
    props = torch.cuda.get_device_properties(0)
    dev = "%s_cu%d" % (str(getattr(props, "gcnArchName", "gpu")).split(":")[0], props.multi_processor_count)
    shapes = DEFAULT_SHAPES
    if args.shapes:
        shapes = []
        for s in args.shapes:
            parts = s.split(":")
            shapes.append((parts[0], int(parts[1]), int(parts[2]), parts[3] if len(parts) > 3 else "fp32"))
    for bb, S, B, prec in shapes:
        hp = helpers.hyper_params(bb)
        if S == 512:
            hp["img_size"] = 512
            hp["feature_map_shapes"] = [32, 16, 8, 4, 2, 1]
        w = helpers.synthetic_weights(bb, hp)
        tabs = []
        for r in range(args.repeats):
            m = SSDModel(bb, hp, max_batch=B, precision=prec)
            m.set_weights(w)
            m.set_tuning("")                   # nothing preset: time everything
            m._ensure(B)
            tabs.append(m.get_tuning())
            del m
            torch.cuda.empty_cache()
        table = majority(tabs)
        agree = sum(t == tabs[0] for t in tabs)
        key = tuning.table_key(bb, S, hp["total_labels"], hp["aspect_ratios"], B) + ("" if prec == "fp32" else "_" + prec)
        with open(os.path.join(args.out, key + ".tune"), "w") as f:
            f.write(tuning.with_header(table, key=key, build=build, device=dev, repeats=args.repeats,
                                       version=ssd_hip.lib().ssd_version().decode().replace(" ", "_")))
        print("%-40s %d lines, sha16 %s, %d/%d runs identical" % (key, len(table.splitlines()), tuning.sha16(table), agree,
                                                                 args.repeats), flush=True)


def adopt(args):
    import tuning
    n = 0
    for f in sorted(os.listdir(args.adopt)):
        if not f.endswith(".tune"):
            continue
        m = re.match(r"^((?:mobilenet_v2|vgg16)_\d+_\d+_a[\d-]+_b\d+(?:_bf16)?)(?:_.*)?\.tune$", f)
        if not m or "=" in f:                       # tables of non-default option sets are not shipped
            continue
        text = open(os.path.join(args.adopt, f)).read()
        dst = tuning.shipped_path(m.group(1))
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as g:
            g.write(text if text.startswith("#") else tuning.with_header(text, key=m.group(1)))
        n += 1
    print("adopted %d tables into %s" % (n, tuning.SHIPPED_DIR))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "tables"))
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--shapes", nargs="*", help="backbone:size:batch[:bf16] ...")
    ap.add_argument("--adopt")
    a = ap.parse_args()
    adopt(a) if a.adopt else generate(a)
