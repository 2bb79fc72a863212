"""Per-tensor deviation of the bf16 net from the fp32 net along the MobileNetV2 chain (diagnostics):
python tests/micro/bf16_layers.py [B] [S]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "tf-ssd_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np
import helpers
from models.ssd_mobilenet_v2 import get_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hp = helpers.hyper_params("mobilenet_v2")
if S != 300:
    hp["img_size"] = S
    hp["feature_map_shapes"] = [32, 16, 8, 4, 2, 1]
w = helpers.synthetic_weights("mobilenet_v2", hp)
x = helpers.images(B, S, seed=0)
names = ["expanded_conv_project_BN"] + ["block_%d_out" % k for k in range(1, 17)] + ["block_13_expand_relu", "out_relu", "extra1_2", "extra2_2", "extra3_2", "extra4_2"]


def run(prec, opts):
    m = get_model(hp, max_batch=B, precision=prec)
    for k, v in opts.items():
        m.set_option(k, v)
    m.set_weights(w)
    d, p = m(x)
    acts = {n: m.fetch_activation(n).copy() for n in names}
    return d.cpu().numpy(), p.cpu().numpy(), acts, {r["name"]: r["config"] for r in m.layers(B) if r["flops"] > 0}


d32, p32, a32, _ = run("fp32", {})
for label, opts in (("default", {}), ("no image kernel", {"fuse_image": 0}), ("fp32 band kernel", {"fuse_band": 1}),
                    ("layer by layer", {"fuse_blocks": 0})):
    d16, p16, a16, cfg = run("bf16", opts)
    print("== bf16 %s: probs max %.3e, deltas max %.3e" % (label, np.abs(p16 - p32).max(), np.abs(d16 - d32).max()))
    for n in names:
        e = np.abs(a16[n] - a32[n])
        print("   %-28s max|ref| %8.3f  max err %9.3e  rms err / rms ref %9.3e" % (
            n, np.abs(a32[n]).max(), e.max(), np.sqrt((e ** 2).mean()) / max(1e-30, np.sqrt((a32[n] ** 2).mean()))))
    if label == "default":
        print("   configs:", {k: v for k, v in cfg.items() if "fused" in k or "heads" in k or k in ("Conv_1", "extra1_2")})

# sensitivity of the FP32 net to a bf16-sized perturbation of its input only (relative 2^-9 per element, one injection)
def rne(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32).reshape(a.shape)
m = get_model(hp, max_batch=B)
m.set_weights(w)
d0, p0 = [t.cpu().numpy() for t in m(x)]
a0 = {n: m.fetch_activation(n).copy() for n in names}
d1, p1 = [t.cpu().numpy() for t in m(rne(x))]
a1 = {n: m.fetch_activation(n).copy() for n in names}
print("== fp32 net, input rounded to bf16 (one injection of 2^-9 relative noise): probs max %.3e deltas max %.3e" % (np.abs(p1 - p0).max(), np.abs(d1 - d0).max()))
for n in names:
    e = np.abs(a1[n] - a0[n])
    print("   %-28s rms err / rms ref %9.3e" % (n, np.sqrt((e ** 2).mean()) / max(1e-30, np.sqrt((a0[n] ** 2).mean()))))
