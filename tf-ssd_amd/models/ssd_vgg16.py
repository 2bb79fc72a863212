"""Drop-in for the reference's ``models/ssd_vgg16.py``."""
import numpy as np
import torch

import ssd_hip as _h
from models._net import SSDModel, keras_default_init


class L2Normalization(object):
    """reference models/ssd_vgg16.py:7-31: ``tf.nn.l2_normalize(x, -1) * scale`` with a
    learnable per-channel scale initialised to ``scale_factor``."""

    def __init__(self, scale_factor, **kwargs):
        self.scale_factor = scale_factor
        self.name = kwargs.get("name", "l2_normalization")
        self.scale = None

    def get_config(self):
        return {"name": self.name, "scale_factor": self.scale_factor}

    def build(self, input_shape):
        self.scale = torch.full((int(input_shape[-1]),), float(self.scale_factor), dtype=torch.float32,
                                device=_h.device())

    def call(self, inputs):
        x = _h.to_dev(inputs)
        if self.scale is None:
            self.build(x.shape)
        C = x.shape[-1]
        out = torch.empty_like(x)
        _h.check(_h.lib().ssd_l2norm(_h.ptr(x), x.numel() // C, C, _h.ptr(self.scale), _h.ptr(out), _h.stream()),
                 "L2Normalization")
        return out

    __call__ = call


def get_model(hyper_params, max_batch=None, precision="fp32"):
    """reference models/ssd_vgg16.py:33-97."""
    model = SSDModel("vgg16", hyper_params, max_batch=max_batch, precision=precision)
    model.set_weights(keras_default_init(model.param_specs, "vgg16"))
    return model


def init_model(model):
    """reference models/ssd_vgg16.py:99-105 feeds a 512x512 dummy into its fully
    convolutional graph; this net is planned for ``img_size`` so the dummy follows it."""
    s = model.img_size
    model(np.random.default_rng(0).random((1, s, s, 3), dtype=np.float32))
