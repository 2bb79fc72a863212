"""Drop-in for the reference's ``utils/io_utils.py`` (same flags, paths and asserts)."""
import argparse
import os
from datetime import datetime


def get_log_path(model_type, custom_postfix=""):
    """reference utils/io_utils.py:6-15."""
    return "logs/{}{}/{}".format(model_type, custom_postfix, datetime.now().strftime("%Y%m%d-%H%M%S"))


def get_model_path(model_type):
    """reference utils/io_utils.py:17-29.  The weights container here is a NumPy ``.npz``
    keyed by the Keras variable names (h5py is not available); the reference's ``.h5``
    suffix is kept so callers see the same path."""
    main_path = "trained"
    if not os.path.exists(main_path):
        os.makedirs(main_path)
    model_path = os.path.join(main_path, "ssd_{}_model_weights.h5".format(model_type))
    return model_path


def handle_args(argv=None):
    """reference utils/io_utils.py:31-44."""
    parser = argparse.ArgumentParser(description="SSD: Single Shot MultiBox Detector Implementation")
    parser.add_argument("-handle-gpu", action="store_true", help="GPU compatibility flag (selects/probes the HIP device)")
    parser.add_argument("--backbone", required=False,
                        default="mobilenet_v2",
                        metavar="['mobilenet_v2', 'vgg16']",
                        help="Which backbone used for the ssd")
    args = parser.parse_args(argv)
    return args


def is_valid_backbone(backbone):
    """reference utils/io_utils.py:46-52."""
    assert backbone in ["mobilenet_v2", "vgg16"]


def handle_gpu_compatibility():
    """reference utils/io_utils.py:54-61: the reference toggles TF memory growth; here it
    initialises the HIP device of this process and prints (not raises) any failure."""
    try:
        import ssd_hip
        ssd_hip.device()
    except Exception as e:
        print(e)
