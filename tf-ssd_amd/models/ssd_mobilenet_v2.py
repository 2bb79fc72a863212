"""Drop-in for the reference's ``models/ssd_mobilenet_v2.py``."""
import numpy as np

from models._net import SSDModel, keras_default_init


def get_model(hyper_params, max_batch=None, precision="fp32"):
    """reference models/ssd_mobilenet_v2.py:7-35: MobileNetV2 (alpha 1, no top) tapped at
    ``block_13_expand_relu`` and the final ``out_relu``, 4 extra (1x1 -> 3x3 s2) blocks, 12
    head convs, softmax.  Returns a callable model: images [B,S,S,3] in [0,1] ->
    (pred_deltas [B,N,4], pred_labels [B,N,L])."""
    model = SSDModel("mobilenet_v2", hyper_params, max_batch=max_batch, precision=precision)
    model.set_weights(keras_default_init(model.param_specs, "mobilenet_v2"))
    return model


def init_model(model):
    """reference models/ssd_mobilenet_v2.py:37-43: one dummy forward (builds/plans the net)."""
    s = model.img_size
    model(np.random.default_rng(0).random((1, s, s, 3), dtype=np.float32))
