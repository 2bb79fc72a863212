"""Drop-in for the reference's ``models/header.py``.  In the reference ``HeadWrapper``
reshapes + concatenates the six head feature maps and ``get_head_from_outputs`` adds the 12
head convs and the softmax.  Here the head convs' epilogue writes each level directly at
its offset of the concatenated ``[B, N, K]`` buffer (no reshape / concat pass) and the
softmax runs in place; inside ``get_model`` both live in the native graph (csrc/ssd_net.hip
``Builder::heads``, label + box conv of a level fused into one GEMM); ``get_head_from_outputs``
is the same composition through the op-level C ABI for callers that bring their own feature maps."""
import torch

import ssd_hip as _h


class HeadWrapper(object):
    """reference models/header.py:4-41: merge per-level maps ``[B,f,f,A*K]`` into
    ``[B, sum f*f*A, K]``.  NHWC makes this a pure view + concatenation."""

    def __init__(self, last_dimension, **kwargs):
        self.last_dimension = last_dimension
        self.name = kwargs.get("name", "head_wrapper")

    def get_config(self):
        return {"name": self.name, "last_dimension": self.last_dimension}

    def call(self, inputs):
        outs = []
        for t in inputs:
            t = _h.to_dev(t)
            outs.append(t.reshape(t.shape[0], -1, self.last_dimension))
        return torch.cat(outs, dim=1)

    __call__ = call


def head_layout(hyper_params):
    """Per level: (first prior index, anchors per cell) -- the offsets the head convs store at."""
    out, off = [], 0
    for f, ars in zip(hyper_params["feature_map_shapes"], hyper_params["aspect_ratios"]):
        a = len(ars) + 1
        out.append((off, a))
        off += f * f * a
    return out, off


def get_head_from_outputs(hyper_params, outputs, weights=None, seed=0):
    """reference models/header.py:43-67: per level ``Conv2D(A*L, 3x3, same)`` (labels) and
    ``Conv2D(A*4, 3x3, same)`` (boxes) on the feature map, HeadWrapper merge, softmax on the
    labels.  Runs as a composition of C-ABI calls: each head conv (``ssd_conv2d``, MFMA implicit
    GEMM) stores its level directly at its offset of the concatenated ``[B,N,L]`` / ``[B,N,4]``
    buffer (out_batch_stride / out_pixel_stride), then ``ssd_softmax`` in place.

    ``outputs``: list of NHWC feature maps ``[B,f,f,C]`` (any device / numpy).  ``weights``:
    dict ``"{i}_conv_label_output/kernel" | "/bias" | "{i}_conv_boxes_output/..."`` in Keras
    layouts; names that are missing are created the way Keras creates the layer's variables
    (glorot-uniform kernel, zero bias) and stored into the dict.  Returns
    ``(pred_deltas [B,N,4], pred_labels [B,N,L])`` like the reference."""
    import ctypes
    import numpy as np
    lib = _h.lib()
    L = int(hyper_params["total_labels"])
    n_anchors = [len(x) + 1 for x in hyper_params["aspect_ratios"]]
    if len(outputs) != len(n_anchors):
        raise ValueError("expected %d feature maps, got %d" % (len(n_anchors), len(outputs)))
    feats = [_h.to_dev(o) for o in outputs]
    for f in feats:
        if f.dim() != 4 or f.shape[0] != feats[0].shape[0]:
            raise ValueError("feature maps must be [B,f,f,C] with one batch size")
    B = feats[0].shape[0]
    N = sum(f.shape[1] * f.shape[2] * a for f, a in zip(feats, n_anchors))
    dev = feats[0].device
    pred_labels = torch.empty((B, N, L), dtype=torch.float32, device=dev)
    pred_deltas = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    weights = weights if weights is not None else {}
    rng = np.random.default_rng(seed)
    keep = []
    off = 0
    for i, (f, A) in enumerate(zip(feats, n_anchors)):
        _, H, W, C = f.shape
        for kind, K, out in (("label", L, pred_labels), ("boxes", 4, pred_deltas)):
            name = "%d_conv_%s_output" % (i + 1, kind)
            if name + "/kernel" not in weights:
                lim = np.sqrt(6.0 / (9 * C + 9 * A * K))
                weights[name + "/kernel"] = rng.uniform(-lim, lim, (3, 3, C, A * K)).astype(np.float32)
            if name + "/bias" not in weights:
                weights[name + "/bias"] = np.zeros((A * K,), np.float32)
            w = _h.to_dev(weights[name + "/kernel"])
            b = _h.to_dev(weights[name + "/bias"])
            if tuple(w.shape) != (3, 3, C, A * K) or tuple(b.shape) != (A * K,):
                raise ValueError("%s: kernel %s / bias %s do not fit a [%d ch] map with %d anchors" % (
                    name, tuple(w.shape), tuple(b.shape), C, A))
            packed = torch.empty((lib.ssd_conv_packed_weight_floats(3, 3, C, A * K),), dtype=torch.float32, device=dev)
            _h.check(lib.ssd_conv_pack_weights(_h.ptr(w), 3, 3, C, A * K, _h.ptr(packed), _h.stream()), name)
            pb = ctypes.c_int()
            pa = ctypes.c_int()
            lib.ssd_same_pads(H, 3, 1, 1, ctypes.byref(pb), ctypes.byref(pa))
            pl, pr = ctypes.c_int(), ctypes.c_int()
            lib.ssd_same_pads(W, 3, 1, 1, ctypes.byref(pl), ctypes.byref(pr))
            d = _h.ConvDesc(B, H, W, C, A * K, 3, 3, 1, 1, pb.value, pl.value, pa.value, pr.value, _h.ACT_NONE, 0)
            optr = _h.vp(out.data_ptr() + 4 * off * K)
            _h.check(lib.ssd_conv2d(ctypes.byref(d), _h.ptr(f), _h.ptr(packed), None, _h.ptr(b), None, optr,
                                    N * K, A * K, _h.stream()), name)
            keep.append((w, b, packed))
        off += H * W * A
    _h.check(lib.ssd_softmax(_h.ptr(pred_labels), B * N, L, _h.ptr(pred_labels), _h.stream()), "conf")
    torch.cuda.current_stream().synchronize()      # the packed temporaries may be released now
    return pred_deltas, pred_labels
