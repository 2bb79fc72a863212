"""CPU oracle (TEST INFRASTRUCTURE, not product code) for the SSD forward graphs.

NumPy restatement of the TF/Keras ops the reference's models dispatch and of the two
graphs themselves:

* ``models/ssd_mobilenet_v2.py:7-35`` on top of **[3P]** ``keras-applications==1.0.8``
  ``MobileNetV2(alpha=1.0, include_top=False)`` (pinned by ``environment.yml:32``; not
  under /root/reference, restated from its published architecture: SURVEY.md Appendix A);
* ``models/ssd_vgg16.py:33-97``;
* ``models/header.py:4-67`` (12 head convs, reshape/concat, softmax).

PARITY STATUS: **parity unpinned** -- TensorFlow/Keras cannot be imported in the build
container and the reference ships no golden vectors, so conv padding rules, BatchNorm
inference formula, ``l2_normalize`` epsilon and MaxPool SAME semantics are restated from
the TF 2.0 documentation/source knowledge recorded in SURVEY.md Appendix A.  The conv
arithmetic is cross-checked against an independent implementation (torch CPU/oneDNN) in
``tests/test_oracle_net.py``.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

import numpy as np

F32 = np.float32
BN_EPS = 1e-3   # keras-applications MobileNetV2: BatchNormalization(epsilon=1e-3, momentum=0.999)


# ------------------------------------------------------------------ TF padding rules
def same_pads(size, k, stride, dilation=1):
    """TF SAME: out=ceil(in/s); p=max((out-1)*s+(k-1)*d+1-in,0); before=p//2, after=p-before."""
    out = -(-size // stride)
    keff = (k - 1) * dilation + 1
    p = max((out - 1) * stride + keff - size, 0)
    return out, p // 2, p - p // 2


def correct_pad(size, k=3):
    """[3P] keras-applications ``correct_pad`` used by MobileNetV2 before stride-2 convs:
    even input -> (0,1), odd -> (1,1) for k=3."""
    adjust = 1 - size % 2
    correct = k // 2
    return correct - adjust, correct


# ------------------------------------------------------------------ ops
def _pad(x, pt, pb, pl, pr, value=0.0):
    if pt == pb == pl == pr == 0:
        return x
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=value)


def conv2d(x, w, bias=None, stride=1, dilation=1, padding="same"):
    """Keras Conv2D, NHWC x HWIO, fp32.  padding: 'same' | 'valid' | (pt, pb, pl, pr)."""
    x = np.asarray(x, F32)
    w = np.asarray(w, F32)
    B, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    if padding == "same":
        Ho, pt, pb = same_pads(H, kh, stride, dilation)
        Wo, pl, pr = same_pads(W, kw, stride, dilation)
    elif padding == "valid":
        pt = pb = pl = pr = 0
    else:
        pt, pb, pl, pr = padding
    xp = _pad(x, pt, pb, pl, pr)
    Hp, Wp = xp.shape[1], xp.shape[2]
    Ho = (Hp - ((kh - 1) * dilation + 1)) // stride + 1
    Wo = (Wp - ((kw - 1) * dilation + 1)) // stride + 1
    out = np.empty((B, Ho, Wo, Cout), F32)
    wm = w.reshape(kh * kw * Cin, Cout)
    for b in range(B):
        cols = np.empty((Ho, Wo, kh, kw, Cin), F32)
        for ky in range(kh):
            for kx in range(kw):
                ys, xs = ky * dilation, kx * dilation
                cols[:, :, ky, kx, :] = xp[b, ys:ys + (Ho - 1) * stride + 1:stride,
                                           xs:xs + (Wo - 1) * stride + 1:stride, :]
        out[b] = (cols.reshape(Ho * Wo, kh * kw * Cin) @ wm).reshape(Ho, Wo, Cout)
    if bias is not None:
        out = out + np.asarray(bias, F32)
    return out


def depthwise_conv2d(x, w, stride=1, padding="same"):
    """Keras DepthwiseConv2D 3x3 (depth_multiplier 1), w [kh,kw,C,1], no bias."""
    x = np.asarray(x, F32)
    w = np.asarray(w, F32)[..., 0]
    B, H, W, C = x.shape
    kh, kw, _ = w.shape
    if padding == "same":
        Ho, pt, pb = same_pads(H, kh, stride)
        Wo, pl, pr = same_pads(W, kw, stride)
    elif padding == "valid":
        pt = pb = pl = pr = 0
    else:
        pt, pb, pl, pr = padding
    xp = _pad(x, pt, pb, pl, pr)
    Ho = (xp.shape[1] - kh) // stride + 1
    Wo = (xp.shape[2] - kw) // stride + 1
    out = np.zeros((B, Ho, Wo, C), F32)
    for ky in range(kh):
        for kx in range(kw):
            out += xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride, :] * w[ky, kx]
    return out


def batch_norm(x, gamma, beta, mean, var, eps=BN_EPS):
    """Inference BatchNormalization, tf.nn.batch_normalization form:
    inv = gamma * rsqrt(var + eps);  y = x * inv + (beta - mean * inv)."""
    inv = (np.asarray(gamma, F32) / np.sqrt(np.asarray(var, F32) + F32(eps))).astype(F32)
    return (np.asarray(x, F32) * inv + (np.asarray(beta, F32) - np.asarray(mean, F32) * inv)).astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


def relu6(x):
    return np.minimum(np.maximum(x, F32(0)), F32(6))


def max_pool(x, k, stride, padding="same"):
    """Keras MaxPool2D; SAME pads are ignored (-inf), as TF does."""
    B, H, W, C = x.shape
    if padding == "same":
        Ho, pt, pb = same_pads(H, k, stride)
        Wo, pl, pr = same_pads(W, k, stride)
    else:
        pt = pb = pl = pr = 0
    xp = _pad(x, pt, pb, pl, pr, value=-np.inf)
    Ho = (xp.shape[1] - k) // stride + 1
    Wo = (xp.shape[2] - k) // stride + 1
    out = np.full((B, Ho, Wo, C), -np.inf, F32)
    for ky in range(k):
        for kx in range(k):
            out = np.maximum(out, xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride, :])
    return out


def l2_normalize_scale(x, gamma):
    """models/ssd_vgg16.py:31: tf.nn.l2_normalize(x, -1) * scale
    [3P] = x * rsqrt(max(sum(x^2), 1e-12)) * gamma."""
    sq = np.sum(x * x, axis=-1, keepdims=True, dtype=F32)
    return (x * (F32(1) / np.sqrt(np.maximum(sq, F32(1e-12)))) * np.asarray(gamma, F32)).astype(F32)


def softmax(x):
    e = np.exp(x - np.max(x, -1, keepdims=True))
    return (e / np.sum(e, -1, keepdims=True, dtype=F32)).astype(F32)


class NumpyOps(object):
    """Default op set of the graphs below (plain NumPy).  oracle/torch_cpu_graph.py provides
    the same interface on torch-CPU (oneDNN) for the multi-threaded CPU baseline."""
    conv2d = staticmethod(conv2d)
    depthwise_conv2d = staticmethod(depthwise_conv2d)
    batch_norm = staticmethod(batch_norm)
    relu = staticmethod(relu)
    relu6 = staticmethod(relu6)
    max_pool = staticmethod(max_pool)
    l2_normalize_scale = staticmethod(l2_normalize_scale)
    softmax = staticmethod(softmax)

    @staticmethod
    def add(a, b):
        return a + b

    @staticmethod
    def reshape(a, shape):
        return a.reshape(shape)

    @staticmethod
    def concat(parts, axis):
        return np.concatenate(parts, axis)

    @staticmethod
    def hw(x):
        return x.shape[1], x.shape[2]

    @staticmethod
    def input(x):
        return np.asarray(x, F32)

    @staticmethod
    def output(x):
        return np.asarray(x, F32)


# ------------------------------------------------------------------ graph specs
# MobileNetV2 inverted-residual table (t, c, n, s) -> 16 blocks after expanded_conv.
_MBV2_BLOCKS = [(24, 2), (24, 1), (32, 2), (32, 1), (32, 1), (64, 2), (64, 1), (64, 1), (64, 1),
                (96, 1), (96, 1), (96, 1), (160, 2), (160, 1), (160, 1), (320, 1)]
_EXTRAS = [(256, 512), (128, 256), (128, 256), (128, 256)]


def mobilenet_v2_param_specs(hyper_params):
    """(name, shape) in Keras layer order: backbone, extras, heads."""
    specs = []

    def bn(name, c):
        specs.extend([(name + "/gamma", (c,)), (name + "/beta", (c,)),
                      (name + "/moving_mean", (c,)), (name + "/moving_variance", (c,))])
    specs.append(("Conv1/kernel", (3, 3, 3, 32)))
    bn("bn_Conv1", 32)
    specs.append(("expanded_conv_depthwise/depthwise_kernel", (3, 3, 32, 1)))
    bn("expanded_conv_depthwise_BN", 32)
    specs.append(("expanded_conv_project/kernel", (1, 1, 32, 16)))
    bn("expanded_conv_project_BN", 16)
    cin = 16
    for k, (cout, s) in enumerate(_MBV2_BLOCKS, start=1):
        p = "block_%d_" % k
        specs.append((p + "expand/kernel", (1, 1, cin, 6 * cin)))
        bn(p + "expand_BN", 6 * cin)
        specs.append((p + "depthwise/depthwise_kernel", (3, 3, 6 * cin, 1)))
        bn(p + "depthwise_BN", 6 * cin)
        specs.append((p + "project/kernel", (1, 1, 6 * cin, cout)))
        bn(p + "project_BN", cout)
        cin = cout
    specs.append(("Conv_1/kernel", (1, 1, 320, 1280)))
    bn("Conv_1_bn", 1280)
    cin = 1280
    for i, (c1, c2) in enumerate(_EXTRAS, start=1):
        specs += [("extra%d_1/kernel" % i, (1, 1, cin, c1)), ("extra%d_1/bias" % i, (c1,)),
                  ("extra%d_2/kernel" % i, (3, 3, c1, c2)), ("extra%d_2/bias" % i, (c2,))]
        cin = c2
    specs += head_param_specs(hyper_params, [576, 1280, 512, 256, 256, 256])
    return specs


def head_param_specs(hyper_params, in_channels):
    L = hyper_params["total_labels"]
    specs = []
    for i, (c, ars) in enumerate(zip(in_channels, hyper_params["aspect_ratios"]), start=1):
        a = len(ars) + 1
        specs += [("%d_conv_label_output/kernel" % i, (3, 3, c, a * L)), ("%d_conv_label_output/bias" % i, (a * L,)),
                  ("%d_conv_boxes_output/kernel" % i, (3, 3, c, a * 4)), ("%d_conv_boxes_output/bias" % i, (a * 4,))]
    return specs


_VGG = [("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128),
        ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256),
        ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512),
        ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512)]


def vgg16_param_specs(hyper_params):
    specs = []
    for name, ci, co in _VGG:
        specs += [(name + "/kernel", (3, 3, ci, co)), (name + "/bias", (co,))]
    specs += [("conv6/kernel", (3, 3, 512, 1024)), ("conv6/bias", (1024,)),
              ("conv7/kernel", (1, 1, 1024, 1024)), ("conv7/bias", (1024,))]
    for name, k, ci, co in [("conv8_1", 1, 1024, 256), ("conv8_2", 3, 256, 512), ("conv9_1", 1, 512, 128),
                            ("conv9_2", 3, 128, 256), ("conv10_1", 1, 256, 128), ("conv10_2", 3, 128, 256),
                            ("conv11_1", 1, 256, 128), ("conv11_2", 3, 128, 256)]:
        specs += [(name + "/kernel", (k, k, ci, co)), (name + "/bias", (co,))]
    specs.append(("l2_normalization/scale", (512,)))
    specs += head_param_specs(hyper_params, [512, 1024, 512, 256, 256, 256])
    return specs


def param_specs(backbone, hyper_params):
    return mobilenet_v2_param_specs(hyper_params) if backbone == "mobilenet_v2" else vgg16_param_specs(hyper_params)


# ------------------------------------------------------------------ forward graphs
def heads_forward(hyper_params, feats, P, acts=None, ops=NumpyOps):
    """models/header.py:43-67 + HeadWrapper (:34-41) + softmax (:64)."""
    L = hyper_params["total_labels"]
    labels, boxes = [], []
    for i, f in enumerate(feats, start=1):
        B = f.shape[0]
        lab = ops.conv2d(f, P["%d_conv_label_output/kernel" % i], P["%d_conv_label_output/bias" % i])
        box = ops.conv2d(f, P["%d_conv_boxes_output/kernel" % i], P["%d_conv_boxes_output/bias" % i])
        labels.append(ops.reshape(lab, (B, -1, L)))
        boxes.append(ops.reshape(box, (B, -1, 4)))
    logits = ops.concat(labels, 1)
    if acts is not None:
        acts["labels_head"] = ops.output(logits)
    return ops.output(ops.concat(boxes, 1)), ops.output(ops.softmax(logits))


def mobilenet_v2_ssd_forward(hyper_params, P, x, acts=None, ops=NumpyOps):
    """models/ssd_mobilenet_v2.py:7-35 + [3P] MobileNetV2 (Appendix A).  x [B,S,S,3] in [0,1]."""
    def rec(name, v):
        if acts is not None:
            acts[name] = ops.output(v)
        return v

    def bn(name, v):
        return ops.batch_norm(v, P[name + "/gamma"], P[name + "/beta"], P[name + "/moving_mean"],
                              P[name + "/moving_variance"])
    x = ops.input(x)
    pt, pb = correct_pad(ops.hw(x)[0])
    pl, pr = correct_pad(ops.hw(x)[1])
    y = ops.conv2d(x, P["Conv1/kernel"], stride=2, padding=(pt, pb, pl, pr))
    y = rec("Conv1_relu", ops.relu6(bn("bn_Conv1", y)))
    y = ops.depthwise_conv2d(y, P["expanded_conv_depthwise/depthwise_kernel"])
    y = rec("expanded_conv_depthwise_relu", ops.relu6(bn("expanded_conv_depthwise_BN", y)))
    y = rec("expanded_conv_project_BN", bn("expanded_conv_project_BN", ops.conv2d(y, P["expanded_conv_project/kernel"])))
    cin = 16
    tap1 = None
    for k, (cout, s) in enumerate(_MBV2_BLOCKS, start=1):
        p = "block_%d_" % k
        inp = y
        y = rec(p + "expand_relu", ops.relu6(bn(p + "expand_BN", ops.conv2d(y, P[p + "expand/kernel"]))))
        if k == 13:
            tap1 = y                                   # block_13_expand_relu (ssd_mobilenet_v2.py:18)
        if s == 2:
            pt, pb = correct_pad(ops.hw(y)[0])
            pl, pr = correct_pad(ops.hw(y)[1])
            y = ops.depthwise_conv2d(y, P[p + "depthwise/depthwise_kernel"], stride=2, padding=(pt, pb, pl, pr))
        else:
            y = ops.depthwise_conv2d(y, P[p + "depthwise/depthwise_kernel"])
        y = rec(p + "depthwise_relu", ops.relu6(bn(p + "depthwise_BN", y)))
        y = bn(p + "project_BN", ops.conv2d(y, P[p + "project/kernel"]))
        if cin == cout and s == 1:
            y = ops.add(inp, y)
        y = rec(p + "out", y)
        cin = cout
    y = rec("out_relu", ops.relu6(bn("Conv_1_bn", ops.conv2d(y, P["Conv_1/kernel"]))))
    feats = [tap1, y]
    for i in range(1, 5):
        y = rec("extra%d_1" % i, ops.relu(ops.conv2d(y, P["extra%d_1/kernel" % i], P["extra%d_1/bias" % i], padding="valid")))
        y = rec("extra%d_2" % i, ops.relu(ops.conv2d(y, P["extra%d_2/kernel" % i], P["extra%d_2/bias" % i], stride=2, padding="same")))
        feats.append(y)
    return heads_forward(hyper_params, feats, P, acts, ops)


def vgg16_ssd_forward(hyper_params, P, x, acts=None, ops=NumpyOps):
    """models/ssd_vgg16.py:33-97."""
    def rec(name, v):
        if acts is not None:
            acts[name] = ops.output(v)
        return v

    def c(name, v, **kw):
        return rec(name, ops.relu(ops.conv2d(v, P[name + "/kernel"], P[name + "/bias"], **kw)))
    y = ops.input(x)
    y = c("conv1_2", c("conv1_1", y)); y = rec("pool1", ops.max_pool(y, 2, 2))
    y = c("conv2_2", c("conv2_1", y)); y = rec("pool2", ops.max_pool(y, 2, 2))
    y = c("conv3_3", c("conv3_2", c("conv3_1", y))); y = rec("pool3", ops.max_pool(y, 2, 2))
    conv4_3 = c("conv4_3", c("conv4_2", c("conv4_1", y))); y = rec("pool4", ops.max_pool(conv4_3, 2, 2))
    y = c("conv5_3", c("conv5_2", c("conv5_1", y))); y = rec("pool5", ops.max_pool(y, 3, 1))
    y = c("conv6", y, dilation=6)
    conv7 = c("conv7", y)
    conv8_2 = c("conv8_2", c("conv8_1", conv7, padding="valid"), stride=2, padding="same")
    conv9_2 = c("conv9_2", c("conv9_1", conv8_2, padding="valid"), stride=2, padding="same")
    conv10_2 = c("conv10_2", c("conv10_1", conv9_2, padding="valid"), padding="valid")
    conv11_2 = c("conv11_2", c("conv11_1", conv10_2, padding="valid"), padding="valid")
    norm = rec("l2_normalization", ops.l2_normalize_scale(conv4_3, P["l2_normalization/scale"]))
    return heads_forward(hyper_params, [norm, conv7, conv8_2, conv9_2, conv10_2, conv11_2], P, acts, ops)


def forward(backbone, hyper_params, P, x, acts=None, ops=NumpyOps):
    fn = mobilenet_v2_ssd_forward if backbone == "mobilenet_v2" else vgg16_ssd_forward
    return fn(hyper_params, P, x, acts, ops)


def count_macs(backbone, hyper_params, S=300):
    """Per-image conv MACs from the spec (cross-check vs SURVEY.md Appendix C totals)."""
    P = {n: np.zeros(s, F32) for n, s in param_specs(backbone, hyper_params)}
    for n in P:
        if n.endswith("moving_variance"):
            P[n] += 1
    macs = {"total": 0}

    class Counting(NumpyOps):
        @staticmethod
        def conv2d(x, w, *a, **k):
            out = conv2d(x, w, *a, **k)
            macs["total"] += out.shape[1] * out.shape[2] * int(np.prod(w.shape))
            return out

        @staticmethod
        def depthwise_conv2d(x, w, *a, **k):
            out = depthwise_conv2d(x, w, *a, **k)
            macs["total"] += out.shape[1] * out.shape[2] * 9 * out.shape[3]
            return out
    forward(backbone, hyper_params, P, np.zeros((1, S, S, 3), F32), ops=Counting)
    return macs["total"]
