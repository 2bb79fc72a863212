#!/usr/bin/env python
"""Headline benchmark: images/sec of the SSD300 forward + decode/NMS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one synthetic batch already resident in HBM:
``get_decoder_model(ssd_model, priors, hp)`` applied to ``[B,300,300,3]`` fp32 images
(BASELINE.json configs[1]: SSD300 MobileNetV2, batch 64, fp32, fwd + decode/NMS).  Images
shard by batch (one process per GPU, weights + priors replicated, NO data-path collective:
inference is embarrassingly parallel); scaling is weak (B per GPU fixed).

Rank 0 prints ONE JSON line with the whole-job images/sec, the roofline of the dominant
kernel family (fp32 MFMA implicit-GEMM conv, measured live with hipEvents on the launch
stream) and, at N=1, the CPU baseline (torch-CPU/oneDNN port of the same graph + the plain-C
decode/NMS oracle on the host cores; the literal TF-CPU path cannot run here).
"""
import argparse
import json
import os
import sys
import time

# N batches in flight (--lanes N, default 3) = N in-order streams; with the HIP runtime limited to N hardware queues each
# lane owns one, reproducibly (DecoderModel; measured: 3 / 3 1.51-1.53 ms, 2 / 2 1.60 ms per step).  The variable is read when the runtime starts:
# it has to be in the environment before torch is imported.  One step at a time is unaffected by it (measured).
def _lanes_from_argv(default=3):
    for i, a in enumerate(sys.argv):
        if a == "--lanes" and i + 1 < len(sys.argv) and sys.argv[i + 1].isdigit():
            return int(sys.argv[i + 1])
        if a.startswith("--lanes=") and a[8:].isdigit():
            return int(a[8:])
    return default


if "--train" not in sys.argv:            # (the training step keeps the runtime's default: RCCL brings its own streams)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2" if _lanes_from_argv() == 2 else "3")

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "tf-ssd_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32)
# split-bf16 kernels (csrc/ssd_bf16x3.h): every fp32 product is SIX v_mfma_f32_16x16x32_bf16 products, so their
# fp32-equivalent matrix peak is the dense bf16 peak (16 x the fp32 MFMA rate, same guide) / 6
PEAK_BF16_MFMA_TFLOPS = 16 * PEAK_FP32_MFMA_TFLOPS
PEAK_SPLIT_BF16_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def matrix_peak(config):
    """Matrix-core peak of a conv kernel family (by config name): fp32 MFMA tiles, split-bf16 tiles (fp32 results through
    six bf16 products: fp32-equivalent = dense bf16 / 6), bf16 tiles (--dtype bf16: one product, the dense bf16 peak)."""
    if config.startswith(("bf16_", "dmab_")):
        return PEAK_BF16_MFMA_TFLOPS
    return PEAK_SPLIT_BF16_TFLOPS if config.startswith(("mfma3_", "dma3_")) else PEAK_FP32_MFMA_TFLOPS


def fused_peak(config):
    """... of a fused inverted-residual kernel family (ssd_net_layer_config of a fused layer)."""
    if config.endswith("_bf16"):
        return PEAK_BF16_MFMA_TFLOPS
    return PEAK_SPLIT_BF16_TFLOPS if config in ("band3", "image_split", "stem_split") else PEAK_FP32_MFMA_TFLOPS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 64 / 32 for vgg16)")
    ap.add_argument("--backbone", default="mobilenet_v2", choices=["mobilenet_v2", "vgg16"])
    ap.add_argument("--img-size", type=int, default=300, help="300 (reference configs) or 512 (BASELINE configs[4] graph)")
    ap.add_argument("--cpu-sample", type=int, default=8, help="images per pass of the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=3,
                    help="batches in flight per GPU (DecoderModel.submit), each on its own in-order stream / hardware queue: step n+1's backbone overlaps step n's heads / decode / NMS; 1 = strictly one step at a time")
    ap.add_argument("--no-other-leg", action="store_true", help="skip the informational second mode (two lanes / one lane)")
    ap.add_argument("--no-overlap", action="store_true", help="profiling runs: no intra-step side streams (option overlap_heads 0), so per-kernel durations are uncontended")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostics / A-B runs: a net option (ssd_net_set_option) set before finalize, e.g. --opt image_ticket=1; recorded in config.options")
    ap.add_argument("--train", action="store_true", help="time the training step (SURVEY 8f N1) instead of inference")
    ap.add_argument("--no-h2d", dest="h2d", action="store_false",
                    help="skip the separately labelled leg that feeds the K steps from pinned HOST batches (PCIe-inclusive; never `value`)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 (default: the reference's arithmetic, BASELINE configs[1] / [2]) or bf16 (configs[3] / [4]: every matrix "
                         "operand of the dense / 1x1 convs rounded once to bf16, one bf16 MFMA per product, fp32 accumulation "
                         "and epilogues; activations stay fp32 in HBM)")
    ap.add_argument("--inputs", type=int, default=4,
                    help="distinct resident input batches rotated through the timed steps (config.inputs_rotated)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is run this many times (barrier + synchronize around each); `value` / "
                         "`ms_per_step` come from the MEDIAN region, `timing_spread` reports min / median / max")
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"],
                    help="after the headline (BASELINE configs[1]) also time the other single-GPU BASELINE configs -- VGG16 B=32 fp32 "
                         "(configs[2]), the 512x512 B=16 shard in fp32 and bf16 (configs[4]), the training step B=32 in fp32 and bf16 "
                         "(configs[3]), the MobileNetV2 B=64 step in bf16 and the reference's own batch 32 -- each in a fresh process "
                         "(3 regions x >= 10 steps) and append them to the JSON line as `other_configs`; `value` stays configs[1].  "
                         "auto = only for the default workload at N = 1")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code paths at world size 1: RCCL communicator alive (init, first collective, barriers, "
                         "MAX-reduce of the time; --train: bucketed gradient all-reduce on the communication stream)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: re-launch under torchrun, one rank per GPU (RCCL)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                   "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if args.force_dist:
        os.environ["SSD_HIP_FORCE_DIST"] = "1"          # parallel.py: collectives also at world size 1
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # RCCL creates its communicator (and its own streams) at the first collective: do that NOW, before the lanes'
        # streams exist, so that it cannot disturb which hardware queue each lane gets
        _t = torch.zeros(1, device="cuda")
        dist.all_reduce(_t)
        dist.barrier()
        torch.cuda.synchronize()

    import ssd_hip
    from utils import bbox_utils, train_utils, data_utils
    from models.decoder import get_decoder_model
    if args.backbone == "mobilenet_v2":
        from models.ssd_mobilenet_v2 import get_model
    else:
        from models.ssd_vgg16 import get_model

    B = args.batch or (64 if args.backbone == "mobilenet_v2" else 32)
    hp = dict(train_utils.get_hyper_params(args.backbone))
    hp["total_labels"] = 21                      # "bg" + 20 VOC classes (predictor.py:25-27)
    if args.img_size != 300:                     # not a reference config: same graph at another size
        hp["img_size"] = args.img_size
        s_, fm = args.img_size, []
        if args.backbone == "mobilenet_v2":
            for div in (16, 32):
                fm.append(-(-s_ // div))
            t = fm[-1]
            for _ in range(4):
                t = -(-t // 2)
                fm.append(t)
        else:
            raise SystemExit("--img-size is wired for mobilenet_v2 only")
        hp["feature_map_shapes"] = fm
    if args.train:
        return train_bench(args, hp, get_model, rank, world, dist)
    model = get_model(hp, max_batch=B, precision="bf16" if args.dtype == "bf16" else "fp32")
    if args.no_overlap:
        model.set_option("overlap_heads", 0)
    for o in args.opt:
        k, v = o.split("=", 1)
        model.set_option(k, int(v))
    weights = data_utils.synthetic_weights(model, seed=1)
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    decoder_model = get_decoder_model(model, priors, hp, lanes=args.lanes)
    # ROTATE distinct resident batches through the timed steps (four by default): with one batch every lane would re-read the
    # same 69 MB from the Infinity Cache step after step
    n_inputs = max(1, args.inputs)
    xs = [ssd_hip.to_dev(data_utils.synthetic_images(B, hp["img_size"], seed=rank + 1000 * k)) for k in range(n_inputs)]   # resident in HBM
    x = xs[0]
    step_no = [0]

    def next_x():
        step_no[0] += 1
        return xs[step_no[0] % n_inputs]

    def barrier():
        if dist is not None:
            dist.barrier()

    # a step = one full pass (forward + decode/NMS) over one batch.  Default (--lanes 3): consecutive steps run on three
    # replicas of the net, each on its own in-order stream / hardware queue, so a step may start before the previous
    # one has finished; every one of the K steps is complete at the synchronize that closes the timed region.
    # --lanes 1: strictly one step at a time on the caller's stream.  The other mode is always reported beside it.
    def run_steps(dm, n, lanes):
        if lanes > 1:
            for _ in range(n):
                out = dm.submit(next_x(), sync_input=False)        # resident and complete (contract: inputs in HBM)
            dm.wait()
        else:
            for _ in range(n):
                out = dm(next_x())
        return out

    def region(dm, lanes):
        """EXACTLY K steps between barrier + synchronize on both sides."""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(dm, args.steps, lanes)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def reduce_max(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    local_regions = []

    def timed(dm, lanes, repeats=1, keep_local=False):
        """W untimed warm-up steps, then `repeats` timed regions of K steps each (a 20-step region is ~27 ms: one
        region alone is at the mercy of a clock ramp).  Every region's time is the MAX over ranks; returns all."""
        run_steps(dm, max(args.warmup, 2 * lanes), lanes)
        out = []
        for _ in range(max(1, repeats)):
            t = region(dm, lanes)
            if keep_local:
                local_regions.append(t)
            out.append(reduce_max(t))
        return out

    regions = timed(decoder_model, args.lanes, args.repeats, keep_local=True)
    elapsed = sorted(regions)[len(regions) // 2]            # the median region is the headline
    # the other mode beside it (informational): two batches in flight when the headline is one step at a time,
    # one step at a time when the headline keeps two in flight
    other = None
    if not args.no_other_leg:
        if args.lanes == 1:
            dm2 = get_decoder_model(model, priors, hp, lanes=3)
            e2 = sorted(timed(dm2, 3, 3))[1]
            other = {"mode": "three batches in flight (three net replicas on three in-order streams, DecoderModel.submit)",
                     "ms_per_step": 1e3 * e2 / args.steps, "images_per_sec": world * B * args.steps / e2,
                     "lane_calibration": getattr(dm2, "lane_calibration", None)}
            del dm2
        else:
            r1 = sorted(timed(decoder_model, 1, 3))
            e1 = r1[1]
            other = {"mode": "one step at a time", "ms_per_step": 1e3 * e1 / args.steps,
                     "images_per_sec": world * B * args.steps / e1,
                     "ms_per_step_min_median_max": [1e3 * v / args.steps for v in (r1[0], r1[1], r1[-1])]}
    valid = decoder_model.decoder.last_valid_detections
    mean_det = float(valid.float().mean().item())

    # ---- separately labelled (never `value`): the same K steps fed from PINNED HOST batches -- every step's 4 B S S 3
    # bytes cross PCIe inside the timed region, copied by the lane's own stream beside the other lanes' kernels
    # (DecoderModel.submit of an ssd_hip.pinned_empty buffer); what a predictor.py-style caller with host data gets
    h2d = None
    if args.h2d and args.lanes > 1:
        hosts = []
        for k in range(args.lanes + 1):
            hb = ssd_hip.pinned_empty((B, hp["img_size"], hp["img_size"], 3))
            hb.copy_(x.cpu() if k == 0 else torch.from_numpy(data_utils.synthetic_images(B, hp["img_size"], seed=100 + rank + k)))
            hosts.append(hb)

        def h2d_steps(n):
            for i in range(n):
                decoder_model.submit(hosts[i % len(hosts)])
            decoder_model.wait()

        h2d_steps(2 * args.lanes)
        rs = []
        for _ in range(3):
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            h2d_steps(args.steps)
            torch.cuda.synchronize()
            barrier()
            rs.append(reduce_max(time.perf_counter() - t0))
        e = sorted(rs)[1]
        h2d = {"mode": "pinned host batches, H2D copy on each lane's stream inside the timed region (PCIe-inclusive)",
               "ms_per_step": 1e3 * e / args.steps, "images_per_sec": world * B * args.steps / e,
               "bytes_per_step": 4 * B * hp["img_size"] * hp["img_size"] * 3,
               "h2d_GB_per_s": 4e-9 * B * hp["img_size"] * hp["img_size"] * 3 * args.steps / e}

        # the same steps fed from pinned UINT8 batches (the reference's images before `preprocessing`): a quarter of the
        # bytes over PCIe, converted (x 1/255; same-size resize = identity) by ssd_preprocess on the lane's stream
        hosts8 = []
        for k in range(args.lanes + 1):
            hb = ssd_hip.pinned_empty((B, hp["img_size"], hp["img_size"], 3), dtype=torch.uint8)
            hb.copy_((hosts[k] * 255.0 + 0.5).to(torch.uint8))
            hosts8.append(hb)
        hosts_f = hosts
        hosts = hosts8
        h2d_steps(2 * args.lanes)
        rs = []
        for _ in range(3):
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            h2d_steps(args.steps)
            torch.cuda.synchronize()
            barrier()
            rs.append(reduce_max(time.perf_counter() - t0))
        e = sorted(rs)[1]
        h2d["uint8"] = {"mode": "pinned host UINT8 batches: H2D copy + ssd_preprocess (x 1/255) on each lane's stream inside the "
                                "timed region (PCIe-inclusive)",
                        "ms_per_step": 1e3 * e / args.steps, "images_per_sec": world * B * args.steps / e,
                        "bytes_per_step": B * hp["img_size"] * hp["img_size"] * 3,
                        "h2d_GB_per_s": 1e-9 * B * hp["img_size"] * hp["img_size"] * 3 * args.steps / e}
        hosts = hosts_f
        del hosts8

    # ---- roofline leg: same K steps with per-layer hipEvents on the launch stream
    model.set_timing(True)
    for _ in range(args.steps):
        decoder_model(next_x())
    info, nfw = model.read_timing(B)
    model.set_timing(False)
    # the dense-conv family on the fp32 matrix cores: implicit-GEMM tiles and Winograd F(2x2,3x3) tiles
    mfma = [r for r in info if r["kind"] == "conv" and r["config"].startswith(("mfma_", "mfma3_", "bf16_", "wino_", "skinny_", "dma3_", "dmab_")) and r["flops"] > 0]
    mfma_ms = sum(r["ms"] for r in mfma)
    mfma_flops = sum(r["flops"] for r in mfma)
    mfma_exec = sum(r["executed_flops"] for r in mfma)
    wino = [r for r in mfma if r["config"].startswith("wino_")]
    total_ms = sum(r["ms"] for r in info if r["flops"] > 0 or r["bytes"] > 0 or r["kind"] == "nms")
    kinds = {}
    for r in info:
        if r["flops"] == 0 and r["bytes"] == 0 and r["kind"] != "nms":
            continue                      # layer replaced by a fused kernel (only event overhead)
        k = r["kind"]
        if k == "conv" and r["config"].startswith("wino_"):
            k = "conv_winograd"
        elif k == "conv" and r["config"].startswith("mfma3_"):
            k = "conv_split_bf16"
        elif k == "conv" and r["config"].startswith("dma3_"):
            k = "conv_split_bf16_ldsdma"
        elif k == "conv" and r["config"].startswith("bf16_"):
            k = "conv_bf16"
        elif k == "conv" and r["config"].startswith("dmab_"):
            k = "conv_bf16_ldsdma"
        elif k == "conv" and not r["config"].startswith(("mfma_", "skinny_")):
            k = "conv_direct"
        kinds[k] = kinds.get(k, 0.0) + r["ms"]
    achieved = mfma_flops / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0
    executed = mfma_exec / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0
    # time the matrix cores would need at their peak (each layer against the peak of ITS instruction: fp32 MFMA, or
    # six bf16 MFMAs per product) over the time the launches took: a fraction, never above 1
    ideal_ms = sum(r["executed_flops"] / (matrix_peak(r["config"]) * 1e12) * 1e3 for r in mfma)
    frac = ideal_ms / mfma_ms if mfma_ms > 0 else 0.0
    peak_eff = executed / frac if frac > 0 else PEAK_FP32_MFMA_TFLOPS
    split = [r for r in mfma if r["config"].startswith(("mfma3_", "dma3_"))]
    # the single most expensive launch of the family: what `rocprofv3 --kernel-trace --stats` of
    # `bench.py --no-overlap --no-other-leg` lists as that kernel's average duration (profiles/)
    dom = max(mfma, key=lambda r: r["ms"]) if mfma else None
    fused = [r for r in info if r["kind"] == "fused" and r["flops"] > 0]
    fused_ms = sum(r["ms"] for r in fused)
    fused_flops = sum(r["flops"] for r in fused)
    # each fused block priced at the peak of ITS matrix instruction (split-bf16 forms: band3 = row-band kernel of blocks 3-6,
    # image_split = whole-image kernel where it won the race, stem_split)
    fused_ideal_ms = sum(r["flops"] / (fused_peak(r["config"]) * 1e12) * 1e3 for r in fused)
    fused_split = [r for r in fused if fused_peak(r["config"]) != PEAK_FP32_MFMA_TFLOPS]
    step_flops = sum(r["flops"] for r in info)           # algorithmic conv FLOPs of the whole step
    step_tflops = step_flops / (elapsed / args.steps) / 1e12
    if args.layers and rank == 0:
        for r in info:
            sys.stderr.write("%-28s %-8s %-22s %8.4f ms %8.2f GFLOP %7.1f MB  %6.1f TF/s %6.0f GB/s\n" % (
                r["name"], r["kind"], r["config"], r["ms"], r["flops"] / 1e9, r["bytes"] / 1e6,
                r["flops"] / max(r["ms"], 1e-9) / 1e9, r["bytes"] / max(r["ms"], 1e-9) / 1e6))

    # HBM-side traffic of the family, per launch: PMC counters cannot be read from inside this
    # process, so the value comes from the committed rocprofv3 --pmc passes of this same command
    # (profiles/traffic_<backbone>_b<B>.json, written by profiles/collect.sh); null if absent.
    traffic, traffic_detail = None, None
    tpath = os.path.join(REPO, "profiles", "traffic_%s_b%d.json" % (args.backbone, B))
    if os.path.exists(tpath) and hp["img_size"] == 300:
        try:
            fams = json.load(open(tpath))["families"]
            fam_names = [k for k in ("conv_mfma_kernel", "conv_mfma3_kernel", "conv_bf16_kernel", "conv_dma_kernel", "conv_wino_kernel", "conv_skinny_kernel") if k in fams]
            nl = sum(fams[k]["FETCH_SIZE"]["launches"] for k in fam_names)
            # launch-weighted mean over the family's kernels (implicit-GEMM + Winograd tiles)
            fetch = 2.0 * 1024.0 * sum(fams[k]["FETCH_SIZE"]["KB_per_launch_reported"] * fams[k]["FETCH_SIZE"]["launches"]
                                       for k in fam_names) / nl      # gfx950: x2 for wide reads
            write = 1024.0 * sum(fams[k]["WRITE_SIZE"]["KB_per_launch_reported"] * fams[k]["WRITE_SIZE"]["launches"]
                                 for k in fam_names) / sum(fams[k]["WRITE_SIZE"]["launches"] for k in fam_names)
            traffic = fetch + write
            tj = json.load(open(tpath))
            traffic_detail = {"fetch_bytes_per_launch_x2_corrected": fetch, "write_bytes_per_launch": write,
                              "algorithmic_bytes_per_launch": sum(r["bytes"] for r in mfma) / max(len(mfma), 1),
                              "source": "profiles/" + os.path.basename(tpath),
                              "collected_at_commit": tj.get("commit"),
                              "kernel_sources_sha16": tj.get("kernel_sources_sha16")}
            # the PMC pass is only evidence for THESE kernels: a kernel-source change since the
            # collection (profiles/collect.sh) makes the figure stale -> report null
            if tj.get("kernel_sources_sha16") != kernel_sources_sha16():
                traffic_detail["stale"] = "kernel sources changed since the PMC pass; re-run profiles/collect.sh"
                traffic = None
        except (KeyError, ValueError):
            pass

    result = {
        "metric": "images/sec SSD%d (%s) fwd+NMS" % (hp["img_size"], "MobileNetV2" if args.backbone == "mobilenet_v2" else "VGG16"),
        "value": world * B * args.steps / elapsed,
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        # the K-step region repeated: `value` / `ms_per_step` are the MEDIAN region's (max over ranks per region)
        "timing_spread": {"repeats": len(regions), "ms_per_step_min": 1e3 * min(regions) / args.steps,
                          "ms_per_step_median": 1e3 * elapsed / args.steps, "ms_per_step_max": 1e3 * max(regions) / args.steps,
                          "images_per_sec_min_median_max": [world * B * args.steps / max(regions), world * B * args.steps / elapsed,
                                                            world * B * args.steps / min(regions)]},
        # informational: the same K steps in the OTHER launch mode (see --lanes)
        "other_mode": other,
        "h2d_overlapped_predict": h2d,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic (seeded uniform [0,1) images, seeded random weights; no dataset/checkpoint offline)",
        "config": {"workload": "SSD%d %s inference, batch=%d per GPU, %dx%d %s, fwd + decode/NMS (BASELINE.json configs[%d]%s)" % (
                       hp["img_size"], args.backbone, B, hp["img_size"], hp["img_size"], "fp32" if args.dtype == "f32" else "bf16",
                       (1 if args.backbone == "mobilenet_v2" else 2) if hp["img_size"] == 300 else 4,
                       "" if (args.dtype == "bf16") == (hp["img_size"] != 300) else " shape, in %s" % args.dtype),
                   "global_batch": world * B, "priors": model.num_priors, "labels": hp["total_labels"],
                   "mean_detections_per_image": mean_det, "nms_active": mean_det > 0, "parallelism": "batch-sharded x%d, no collective" % world,
                   "inputs_rotated": n_inputs, "results": "device-resident ([B,200,4] boxes, [B,200] labels / scores, [B] valid counts stay in HBM; the "
                                                          "reference's predict() returns them as host arrays: 307 KB per 64-image step, not copied here)",
                   "batches_in_flight_per_gpu": args.lanes, "options": dict(getattr(model, "_options", {})),
                   "lane_calibration": getattr(decoder_model, "lane_calibration", None),
                   # where the kernel choices came from (tuning.py): a shipped table = nothing timed on the
                   # device = the same kernels and bits in every process
                   "kernel_table": getattr(model, "tuning_info", None),
                   # ssd_build_id() of the loaded library recomputed from the checked-out sources (csrc/*.hip|*.h +
                   # include/ssd_hip.h + the compile flags): equal = the library was built from these sources
                   "build_id_from_sources": build_id_from_sources()},
        "roofline": {"bound": "mfma", "kernel": "conv_mfma3_kernel (implicit-GEMM tiles, every fp32 product as six v_mfma_f32_16x16x32_bf16 of an exact "
                                                 "3-way operand split; conv_dma_kernel: the same tiles with both operands copied global -> LDS by LDS-DMA from pre-split / bf16 planes) "
                                                 "+ conv_mfma_kernel / conv_wino_kernel / conv_skinny_kernel (fp32 v_mfma_f32_16x16x4: "
                                                 "implicit-GEMM, Winograd F(2x2,3x3), in-workgroup-split-K tiles), all configs",
                     # `achieved` / `frac` count the FLOPs the matrix cores actually ISSUED (Winograd layers: 16
                     # multiplies per 2x2 output tile instead of 36, whole border tiles) over the hipEvent time of
                     # the family's launches on their stream: a fraction of the peak, never above 1.  The
                     # ALGORITHMIC figure (MACs x 2 of the direct convolution) is reported beside
                     # it as `achieved_algorithmic`; for Winograd layers it is an "effective" rate and may exceed
                     # the peak.
                     # `peak` is the family's blended fp32-equivalent peak: each layer's executed FLOPs priced at the
                     # peak of the instruction it runs on (fp32 MFMA 157.3; split-bf16 = dense bf16 2516.8 / 6 products
                     # = 419.5), so that achieved / peak == frac == matrix time at peak / measured time.
                     "achieved": executed, "peak": peak_eff, "unit": "TFLOP/s",
                     "frac": frac, "frac_kind": "sum(executed FLOPs / peak of the layer's matrix instruction) / measured time",
                     "peak_detail": {"fp32_mfma": PEAK_FP32_MFMA_TFLOPS, "bf16_mfma_dense": PEAK_BF16_MFMA_TFLOPS,
                                     "split_bf16_fp32_equivalent": PEAK_SPLIT_BF16_TFLOPS,
                                     "bf16_layers": len([r for r in mfma if r["config"].startswith(("bf16_", "dmab_"))]),
                                     "lds_dma_layers": len([r for r in mfma if r["config"].startswith(("dma3_", "dmab_"))]),
                                     "split_bf16_layers": len(split), "split_bf16_ms_per_step": sum(r["ms"] for r in split),
                                     "split_bf16_executed_gflop_per_step": sum(r["executed_flops"] for r in split) / 1e9},
                     "achieved_algorithmic": achieved, "frac_algorithmic_vs_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                     "winograd_layers": len(wino), "winograd_ms_per_step": sum(r["ms"] for r in wino),
                     "traffic": traffic, "traffic_detail": traffic_detail,
                     "launches_per_step": len(mfma), "kernel_ms_per_step": mfma_ms,
                     "algorithmic_gflop_per_step": mfma_flops / 1e9, "executed_gflop_per_step": mfma_exec / 1e9,
                     "dominant_kernel": None if dom is None else {
                         "layer": dom["name"], "config": dom["config"], "ms_per_launch": dom["ms"],
                         "algorithmic_gflop": dom["flops"] / 1e9, "executed_gflop": dom["executed_flops"] / 1e9,
                         "achieved": dom["executed_flops"] / (dom["ms"] * 1e-3) / 1e12,
                         "peak": matrix_peak(dom["config"]),
                         "frac": dom["executed_flops"] / (dom["ms"] * 1e-3) / 1e12 / matrix_peak(dom["config"]),
                         "achieved_algorithmic": dom["flops"] / (dom["ms"] * 1e-3) / 1e12},
                     # the whole step (every kernel, incl. softmax/decode/NMS time) against the same peak, algorithmic FLOPs
                     "achieved_step": step_tflops, "frac_step_vs_fp32_mfma_peak": step_tflops / PEAK_FP32_MFMA_TFLOPS,
                     # ... and against the peak of the instruction most of the step's FLOPs now run on (six bf16 MFMAs per product)
                     "frac_step_vs_split_bf16_peak": step_tflops / PEAK_SPLIT_BF16_TFLOPS,
                     "frac_step_vs_bf16_peak": step_tflops / PEAK_BF16_MFMA_TFLOPS,
                     "algorithmic_gflop_per_step_all": step_flops / 1e9},
        # the other half of the step: whole-block / depthwise+project / stem kernels (MFMA + VALU depthwise)
        # `frac` = matrix time at peak / measured time with every block priced at the peak of its instruction (the
        # split-bf16 forms -- row-band kernel of blocks 3-6, whole-image kernel where it won the race, stem -- at 419.5, the fp32-MFMA kernels at 157.3); `peak` is the blend
        "roofline_fused": {"bound": "mfma", "kernel": "mbv2_stem_kernel + mbv2_band_block_kernel + mbv2_band3_block_kernel + mbv2_image_block_kernel + mbv2_image16_block_kernel + dwproj8_kernel (fused inverted-residual family)",
                           "achieved": fused_flops / (fused_ms * 1e-3) / 1e12 if fused_ms > 0 else None,
                           "peak": (fused_flops / (fused_ideal_ms * 1e-3) / 1e12) if fused_ideal_ms > 0 else PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s",
                           "frac": (fused_ideal_ms / fused_ms) if fused_ms > 0 else None,
                           "frac_vs_fp32_mfma_peak": (fused_flops / (fused_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if fused_ms > 0 else None,
                           "split_bf16_blocks": [r["name"] for r in fused_split],
                           "launches_per_step": len(fused), "kernel_ms_per_step": fused_ms,
                           "algorithmic_gflop_per_step": fused_flops / 1e9},
        "gpu_ms_per_step_by_kind": {k: round(v, 4) for k, v in sorted(kinds.items())},
        "gpu_ms_per_step_sum": total_ms,
    }

    if dist is not None:
        result["multi_gpu"] = ranks_report(dist, sorted(local_regions)[len(local_regions) // 2], args.steps)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.backbone, hp, weights, priors.cpu().numpy(), args.cpu_sample)
    decoder_model.close()
    default_workload = (args.backbone == "mobilenet_v2" and hp["img_size"] == 300 and args.dtype == "f32" and B == 64
                        and not args.opt and not args.no_overlap)
    if rank == 0 and world == 1 and dist is None and (args.other_configs == "on" or (args.other_configs == "auto" and default_workload)):
        del decoder_model, model, xs, x                   # the sub-runs get the whole device
        torch.cuda.empty_cache()
        result["other_configs"] = other_configs(args)
    emit(result, rank, dist)


# The other single-GPU BASELINE.json configs, timed by the SAME invocation (the driver's command is fixed): each is this
# script again in a fresh process (its own runtime setup: the training step keeps the runtime's default hardware queues),
# 3 timed regions x >= 10 steps, no CPU leg.  `value` of the parent line stays configs[1].
OTHER_CONFIGS = [
    ("configs[2]: SSD300 VGG16 inference, batch 32, fp32", ["--backbone", "vgg16", "--batch", "32"]),
    ("configs[4] per-GPU shard: SSD512 MobileNetV2 inference, batch 16, fp32 (the reference's arithmetic)", ["--img-size", "512", "--batch", "16"]),
    ("configs[4] per-GPU shard: SSD512 MobileNetV2 inference, batch 16, bf16", ["--img-size", "512", "--batch", "16", "--dtype", "bf16"]),
    ("configs[3] per-GPU shape: SSD300 MobileNetV2 training step, batch 32, fp32 (the reference's arithmetic)", ["--train", "--batch", "32"]),
    ("configs[3] per-GPU shape: SSD300 MobileNetV2 training step, batch 32, bf16", ["--train", "--batch", "32", "--dtype", "bf16"]),
    ("configs[1] shape in bf16: SSD300 MobileNetV2 inference, batch 64", ["--batch", "64", "--dtype", "bf16"]),
    ("the reference's own batch (predictor.py:9): SSD300 MobileNetV2 inference, batch 32, fp32", ["--batch", "32"]),
    ("configs[0] shape on the GPU: SSD300 MobileNetV2 inference, batch 1, fp32 (latency)", ["--batch", "1"]),
]


def other_configs(args):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}       # each sub-run places its own
    steps = max(10, args.steps // 2)
    out = []
    for label, extra in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", "3", "--repeats", "3",
               "--no-cpu-baseline", "--no-h2d", "--other-configs", "off"] + extra
        t0 = time.perf_counter()
        rec = {"workload": label, "args": " ".join(extra)}
        try:
            pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
            lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
            if pr.returncode != 0 or not lines:
                raise RuntimeError("rc %d: %s" % (pr.returncode, pr.stderr.strip()[-300:]))
            r = json.loads(lines[-1])
            ro = r.get("roofline") or {}
            dk = ro.get("dominant_kernel") or {}
            rec.update({
                "dtype": r["dtype"], "ms_per_step": r["ms_per_step"], "images_per_sec": r["value"], "steps": r["steps"],
                "timing_spread": r.get("timing_spread"),
                "one_step_at_a_time_ms": (r.get("other_mode") or {}).get("ms_per_step"),
                "roofline": {"kernel": ("%s %s" % (dk.get("layer"), dk.get("config"))) if dk else ro.get("kernel"),
                             "achieved": dk.get("achieved", ro.get("achieved")), "peak": dk.get("peak", ro.get("peak")),
                             "frac": dk.get("frac", ro.get("frac")), "unit": "TFLOP/s",
                             "family_frac": ro.get("frac"), "family_achieved": ro.get("achieved"), "family_peak": ro.get("peak"),
                             "frac_step_vs_fp32_mfma_peak": ro.get("frac_step_vs_fp32_mfma_peak", ro.get("frac_algorithmic_vs_fp32_mfma_peak"))},
                "table_build": ((r.get("config") or {}).get("kernel_table") or {}).get("table_build"),
                "kernel_table": (r.get("config") or {}).get("kernel_table"),
                "build_id_from_sources": (r.get("config") or {}).get("build_id_from_sources"),
                "mean_detections_per_image": (r.get("config") or {}).get("mean_detections_per_image"),
                "loss_first_last": [(r.get("config") or {}).get("loss_first_step"), (r.get("config") or {}).get("loss_last_step")] if "--train" in extra else None,
            })
        except Exception as exc:                                  # never lose the headline over a sub-run
            rec["error"] = repr(exc)[:400]
        rec["wall_s"] = round(time.perf_counter() - t0, 2)
        out.append(rec)
    return out


def train_bench(args, hp, get_model, rank, world, dist):
    """--train: BASELINE.json configs[3] shape (SSD300-MobileNetV2 training, 32 images per GPU,
    batch data-parallel with an RCCL all-reduce of the flat gradient) -- in fp32, the reference's
    arithmetic (the bf16 variant of configs[3] is this build's extension and not built).  One step =
    GPU target assignment + training-mode forward + loss + backward + all-reduce + Adam."""
    import torch
    import ssd_hip
    import parallel
    from utils import bbox_utils, data_utils, train_utils
    from ssd_loss import CustomLoss
    B = args.batch or 32
    model = get_model(hp, max_batch=B, precision="bf16" if args.dtype == "bf16" else "fp32")
    cl = CustomLoss(hp["neg_pos_ratio"], hp["loc_loss_alpha"])
    model.compile(learning_rate=1e-3, loss=[cl.loc_loss_fn, cl.conf_loss_fn])
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    x = ssd_hip.to_dev(data_utils.synthetic_images(B, hp["img_size"], seed=rank))
    gt, gl = data_utils.synthetic_gt(B, total_labels=hp["total_labels"], seed=3 + rank)
    gt, gl = ssd_hip.to_dev(gt), ssd_hip.to_dev(gl, torch.int32)

    # the step's stream: a native NON-BLOCKING stream when a process group is alive (SSD_BENCH_TRAIN_STREAM=null keeps
    # torch's legacy NULL stream) -- the NULL stream synchronises implicitly with every blocking stream of the process,
    # RCCL's among them
    own_stream = None
    if os.environ.get("SSD_BENCH_TRAIN_STREAM", "native" if dist is not None else "null") == "native":
        own_stream = ssd_hip.new_stream()

    def step_body():
        yd, yl = train_utils.calculate_actual_outputs(priors, gt, gl, hp)
        model.plan_gradient_exchange(B)                 # N > 1 (or --force-dist): gradient buckets exchanged as the backward finishes them
        loc, conf, g = model.forward_backward(x, yd, yl)
        w = model.exchange_gradients(g)
        model.apply_gradients(g, 1e-3, 1.0 / w)
        return loc, conf

    def step():
        if own_stream is None:
            return step_body()
        with torch.cuda.stream(own_stream):
            return step_body()

    for _ in range(max(args.warmup, 1)):
        loc, conf = step()
    first = float((loc + conf).mean().item())
    regions = []
    local_regions = []
    for _ in range(max(1, args.repeats)):                     # K steps per region, barrier + synchronize on both sides
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loc, conf = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        e = time.perf_counter() - t0
        local_regions.append(e)
        if dist is not None:
            t = torch.tensor([e], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        regions.append(e)
    elapsed = sorted(regions)[len(regions) // 2]
    last = float((loc + conf).mean().item())
    fwd_gflop = (2.026 if args.backbone == "mobilenet_v2" else 62.747) * B     # conv MACs x 2 per image (forward)
    step_tflops = 3.0 * fwd_gflop * 1e9 / (elapsed / args.steps) / 1e12      # forward + backward-data + backward-weights
    # what the matrix cores were actually handed, per instruction family (ssd_net_train_matrix_flops), each priced at
    # the peak of its instruction: fp32-MFMA convs and weight gradients at 157.3, split-bf16 conv tiles at 419.5
    import ctypes
    mf = (ctypes.c_double * 3)()
    ssd_hip.check(ssd_hip.lib().ssd_net_train_matrix_flops(model._net, mf), "train_matrix_flops")
    peak1 = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_SPLIT_BF16_TFLOPS        # the bf16-pipe family of this dtype
    ideal_ms = 1e3 * ((mf[0] + mf[2]) / (PEAK_FP32_MFMA_TFLOPS * 1e12) + mf[1] / (peak1 * 1e12))
    ms_step = 1e3 * elapsed / args.steps
    issued_tflops = (mf[0] + mf[1] + mf[2]) / (ms_step * 1e-3) / 1e12
    result = {
        "metric": "images/sec SSD300 (%s) training step" % ("MobileNetV2" if args.backbone == "mobilenet_v2" else "VGG16"),
        "value": world * B * args.steps / elapsed,
        "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "timing_spread": {"repeats": len(regions), "ms_per_step_min": 1e3 * min(regions) / args.steps,
                          "ms_per_step_median": 1e3 * elapsed / args.steps, "ms_per_step_max": 1e3 * max(regions) / args.steps},
        "dtype": args.dtype, "data": "synthetic (seeded images + ground truth, Keras-default initial weights)",
        "config": {"workload": "SSD300 %s training, batch=%d per GPU, %s, target assignment + fwd + loss + bwd + "
                               "grad all-reduce + Adam (BASELINE.json configs[3] shape%s)" % (
                                   args.backbone, B, "fp32" if args.dtype == "f32" else "bf16 forward / backward-data convs, fp32 master weights / weight gradients / Adam",
                                   "; fp32 instead of bf16" if args.dtype == "f32" else ""),
                   "global_batch": world * B, "parallelism": "batch-DP x%d, RCCL all-reduce of %d fp32 gradients" % (
                       world, ssd_hip.lib().ssd_net_trainable_floats(model._net)),
                   "loss_first_step": first, "loss_last_step": last,
                   "conv_tiles": "cost model (SSD_HIP_TRAIN_AUTOTUNE=0)" if os.environ.get("SSD_HIP_TRAIN_AUTOTUNE") == "0"
                                 else "timed on the device at the first step, kept for the process (csrc/ssd_train.hip pick_measured)"},
        # whole step against the fp32 MFMA peak (the convs the cost model hands to the split-bf16 tiles run above that
        # rate; the step is bound by the elementwise / BatchNorm passes)
        # `frac` = time the matrix cores would need at peak for the FLOPs the step issued (each family at the peak of
        # its instruction) over the measured step; `peak` is the blend (achieved / frac).  The 3x-forward algorithmic
        # figure against the fp32 MFMA peak is kept beside it.
        "roofline": {"bound": "mfma", "kernel": "training step (conv fwd + dgrad via conv_mfma_kernel / conv_mfma3_kernel, wgrad_mfma_kernel)",
                     "achieved": issued_tflops, "peak": issued_tflops / (ideal_ms / ms_step) if ideal_ms > 0 else PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": ideal_ms / ms_step, "traffic": None,
                     "frac_kind": "sum(issued FLOPs / peak of the family's matrix instruction) / measured step time",
                     "issued_gflop_per_step": {"conv_fp32_mfma": mf[0] / 1e9, ("conv_bf16" if args.dtype == "bf16" else "conv_split_bf16"): mf[1] / 1e9,
                                               "wgrad_fp32_mfma": mf[2] / 1e9},
                     "peak_detail": {"fp32_mfma": PEAK_FP32_MFMA_TFLOPS, "split_bf16_fp32_equivalent": PEAK_SPLIT_BF16_TFLOPS,
                                     "bf16_mfma_dense": PEAK_BF16_MFMA_TFLOPS},
                     "achieved_algorithmic_3x_forward": step_tflops,
                     "frac_algorithmic_vs_fp32_mfma_peak": step_tflops / PEAK_FP32_MFMA_TFLOPS,
                     "algorithmic_gflop_per_step": 3.0 * fwd_gflop},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle's training step (torch-CPU autograd graph of the identical network in training mode + loss;
        # oracle/train_oracle.py) on the host cores: a bounded sample of the same workload
        import numpy as np
        from oracle import train_oracle as to
        n = min(B, 4)
        yd, yl = train_utils.calculate_actual_outputs(priors, gt[:n], gl[:n], hp)
        w = model.get_weights()
        xs, yds, yls = x[:n].cpu().numpy(), yd.cpu().numpy(), yl.cpu().numpy()
        threads = min(16, os.cpu_count() or 1)
        to.train_step(args.backbone, hp, w, xs, yds, yls, hp["neg_pos_ratio"], hp["loc_loss_alpha"], threads=threads)   # warm-up
        t1 = time.perf_counter()
        passes = 0
        while passes < 4 and (passes == 0 or time.perf_counter() - t1 < 15.0):
            to.train_step(args.backbone, hp, w, xs, yds, yls, hp["neg_pos_ratio"], hp["loc_loss_alpha"], threads=threads)
            passes += 1
        dt = time.perf_counter() - t1
        result["cpu_baseline"] = {"value": n * passes / dt, "unit": "images/sec", "cores": threads, "host_cores": os.cpu_count(),
                                  "kind": "port", "sample": "%d passes of the oracle's forward + loss + backward on %d images (torch-CPU "
                                  "autograd, no optimiser step; TensorFlow itself is not installable here)" % (passes, n)}
    if dist is not None:
        result["multi_gpu"] = ranks_report(dist, sorted(local_regions)[len(local_regions) // 2], args.steps)
    emit(result, rank, dist)


def ranks_report(dist, local_region_s, steps):
    """N > 1 (or --force-dist) lines say what the collective actually saw: `ranks_seen` = an all-reduce (SUM) of ones
    over RCCL -- it equals n_gpus only if every rank took part -- and the spread of the per-rank median region time
    (the headline takes the MAX over ranks of every region)."""
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"          # (gloo: the world-size-2 CPU test)
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    lo = torch.tensor([local_region_s], dtype=torch.float64, device=dev)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return {"ranks_seen": int(round(float(ones.item()))), "world_size": dist.get_world_size(), "backend": dist.get_backend(),
            "ms_per_step_rank_min": 1e3 * float(lo.item()) / steps, "ms_per_step_rank_max": 1e3 * float(hi.item()) / steps}


def emit(result, rank, dist):
    """Rank 0 prints the ONE JSON line as the LAST line of the job's stdout.  RCCL writes a version banner through C
    stdio when its communicator comes up; redirected to a pipe that text sits in each process's C buffer until exit
    and would land BEHIND the JSON line: every rank flushes C stdio first, the ranks meet, the process group is torn
    down (nothing prints after that), and only then does rank 0 print."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def kernel_sources_sha16():
    """Fingerprint of the kernel sources a PMC traffic pass belongs to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "tf-ssd_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(REPO, "tf-ssd_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_id_from_sources():
    """What csrc/build.sh stamps into the library as ssd_build_id(): sha256 over the kernel sources (the bytes of
    kernel_sources_sha16), include/ssd_hip.h and the compile-flag line of build.sh."""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "tf-ssd_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(REPO, "tf-ssd_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    h.update(open(os.path.join(REPO, "include", "ssd_hip.h"), "rb").read())
    m = re.search(r'^COMMON="([^"]*)"', open(os.path.join(REPO, "tf-ssd_amd", "csrc", "build.sh")).read(), re.M)
    h.update(((m.group(1) if m else "") + "\n").encode())
    return h.hexdigest()[:16]


def _cpu_shard_worker(job):
    """One process of the sharded CPU baseline: `threads` oneDNN threads, its own copy of the port, timed passes of
    `sample` images between a common start barrier and `seconds` of wall time."""
    backbone, hp, wpath, priors, sample, threads, seconds, barrier, seed = job
    import numpy as np
    import torch
    torch.set_num_threads(threads)
    from oracle import torch_cpu_graph as tg
    from oracle import c_oracle as co
    from utils import data_utils
    with np.load(wpath) as z:
        weights = {k: z[k] for k in z.files}
    x = data_utils.synthetic_images(sample, hp["img_size"], seed=seed)

    def one_pass():
        d, p = tg.forward(backbone, hp, weights, x)
        co.decode_nms(d, p, priors, hp["variances"])
    one_pass()                                           # warm-up (oneDNN primitive creation)
    barrier.wait(timeout=300)
    t0 = time.perf_counter()
    n = 0
    while n == 0 or time.perf_counter() - t0 < seconds:
        one_pass()
        n += 1
    return n * sample, t0, time.perf_counter()


def cpu_baseline(backbone, hp, weights, priors, sample):
    """The oracle's port timed on the host cores: torch-CPU (oneDNN) convs of the identical graph + the plain-C
    decode/NMS restatement, on a bounded sample of the same workload.  Two legs: (a) ONE process, the fastest thread
    count of a sweep that stops at the first count that gets slower (oneDNN over-subscribes a many-core host: 0.15
    img/s at 256 threads); (b) the same port SHARDED over processes -- P workers x that thread count, each on its own
    images, common start -- which is how a CPU deployment would use such a host.  `value` is the better of the two,
    `cores` the threads it used (P x T), `host_cores` = os.cpu_count()."""
    import multiprocessing as mp
    import tempfile
    import numpy as np
    import torch
    from oracle import torch_cpu_graph as tg
    from oracle import c_oracle as co
    from utils import data_utils
    ncpu = os.cpu_count() or 1
    x = data_utils.synthetic_images(max(sample, 32), hp["img_size"], seed=0)

    def one_pass(xb):
        d, p = tg.forward(backbone, hp, weights, xb)
        co.decode_nms(d, p, priors, hp["variances"])

    sweep = {}
    prev = 0.0
    for th in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        one_pass(x[:sample])                          # warm-up (oneDNN primitive creation)
        dt = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            one_pass(x[:sample])
            dt = min(dt, time.perf_counter() - t0)
        sweep[th] = sample / dt
        if sweep[th] < prev:                          # the first thread count that gets slower ends the climb
            break
        prev = sweep[th]
    threads = max(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    passes = 0
    while True:
        one_pass(x[:sample])
        passes += 1
        if time.perf_counter() - t0 > 8.0 or passes >= 20:
            break
    dt = time.perf_counter() - t0
    single = sample * passes / dt

    def rate(B, min_s, max_passes):
        one_pass(x[:B])
        t1 = time.perf_counter()
        n = 0
        while True:
            one_pass(x[:B])
            n += 1
            if time.perf_counter() - t1 > min_s or n >= max_passes:
                break
        return B * n / (time.perf_counter() - t1)
    b1 = rate(1, 3.0, 50)        # BASELINE configs[0]: batch 1
    b32 = rate(32, 4.0, 5)       # the reference's own batch_size (predictor.py:9)

    # (b) sharded over processes: P workers x T threads (T = the sweep's best, at most 8), P x T <= host cores
    sharded = None
    tsh = min(threads, 8)
    procs = max(1, min(32, ncpu // tsh))
    if procs > 1:
        try:
            ctx = mp.get_context("spawn")
            with tempfile.TemporaryDirectory() as d:
                wpath = os.path.join(d, "w.npz")
                np.savez(wpath, **weights)
                mgr = ctx.Manager()
                barrier = mgr.Barrier(procs)
                jobs = [(backbone, hp, wpath, priors, sample, tsh, 10.0, barrier, 100 + i) for i in range(procs)]
                with ctx.Pool(procs) as pool:
                    res = pool.map_async(_cpu_shard_worker, jobs).get(timeout=240)
                mgr.shutdown()
            imgs = sum(r[0] for r in res)
            wall = max(r[2] for r in res) - min(r[1] for r in res)
            sharded = {"images_per_sec": imgs / wall, "processes": procs, "threads_per_process": tsh,
                       "seconds": wall, "images": imgs}
        except Exception as exc:                        # a baseline, not the product: report and move on
            sharded = {"error": repr(exc)[:200], "processes": procs, "threads_per_process": tsh}
    best_sharded = sharded and sharded.get("images_per_sec", 0.0) > single
    return {"value": sharded["images_per_sec"] if best_sharded else single, "unit": "images/sec",
            "cores": procs * tsh if best_sharded else threads, "threads": threads,
            "host_cores": ncpu, "kind": "port",
            "single_process_images_per_sec": single, "sharded": sharded,
            "images_per_sec_batch1": b1, "images_per_sec_batch32": b32,
            "thread_sweep_images_per_sec": {str(k): round(v, 2) for k, v in sweep.items()},
            "sample": "%s (torch-CPU oneDNN graph with TF padding + C decode/NMS oracle; TensorFlow itself is not installable "
                      "here); single process: %d passes of %d images at the fastest thread count of a sweep that stops at the "
                      "first slower count; batch-1 and batch-32 rates from 3-4 s samples each" % (
                          ("%d processes x %d threads, ~10 s of passes of %d images each, common start" % (procs, tsh, sample))
                          if best_sharded else "one process", passes, sample)}


if __name__ == "__main__":
    main()
