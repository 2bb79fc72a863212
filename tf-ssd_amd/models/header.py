"""Drop-in for the reference's ``models/header.py``.  In the reference ``HeadWrapper``
reshapes + concatenates the six head feature maps and ``get_head_from_outputs`` adds the 12
head convs and the softmax.  Here the head convs' epilogue writes each level directly at
its offset of the concatenated ``[B, N, K]`` buffer (no reshape / concat pass) and the
softmax runs in place; both live inside the native graph (csrc/ssd_net.hip ``Builder::heads``).
This module keeps the reference's names for code that imports them and exposes the
head layout arithmetic."""
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


def get_head_from_outputs(hyper_params, outputs):
    """reference models/header.py:43-67.  The head convs are part of the native graph built by
    ``get_model``; building them from loose feature tensors is not a supported entry point."""
    raise NotImplementedError("heads are built inside get_model() (csrc/ssd_net.hip Builder::heads)")
