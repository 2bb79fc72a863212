"""Host-side mirror of the reference's ``utils/io_utils.py``: same function names, CLI flags,
directory layout and failure modes, so ``trainer.py`` / ``predictor.py`` callers are unchanged.
Cited line numbers are the reference's."""
import argparse
import os
import time

BACKBONES = ("mobilenet_v2", "vgg16")
_WEIGHTS_DIR = "trained"


def get_log_path(model_type, custom_postfix=""):
    """``logs/<model_type><postfix>/<YYYYmmdd-HHMMSS>`` (utils/io_utils.py:6-15)."""
    stamp = time.strftime("%Y%m%d-%H%M%S", time.localtime())
    return "/".join(("logs", "%s%s" % (model_type, custom_postfix), stamp))


def get_model_path(model_type):
    """``trained/ssd_<model_type>_model_weights.h5`` and make sure the directory exists
    (utils/io_utils.py:17-29).  The file is a real HDF5 container in the Keras ``save_weights``
    layout (utils/h5_writer.py / utils/h5_reader.py: pure Python, h5py is not available), so a
    checkpoint trained by the reference loads here and vice versa."""
    os.makedirs(_WEIGHTS_DIR, exist_ok=True)
    return os.path.join(_WEIGHTS_DIR, "ssd_%s_model_weights.h5" % model_type)


def handle_args(argv=None):
    """The reference's two flags: ``-handle-gpu`` and ``--backbone`` (utils/io_utils.py:31-44)."""
    cli = argparse.ArgumentParser(description="SSD: Single Shot MultiBox Detector Implementation")
    cli.add_argument("-handle-gpu", action="store_true",
                     help="GPU compatibility flag (selects/probes the HIP device)")
    cli.add_argument("--backbone", default=BACKBONES[0], required=False, metavar=str(list(BACKBONES)),
                     help="Which backbone used for the ssd")
    return cli.parse_args(argv)


def is_valid_backbone(backbone):
    """AssertionError for anything but the two supported backbones (utils/io_utils.py:46-52)."""
    assert backbone in BACKBONES


def handle_gpu_compatibility():
    """The reference switches TF memory growth on and prints any failure
    (utils/io_utils.py:54-61); the equivalent here is initialising this process's HIP device."""
    try:
        import ssd_hip
        ssd_hip.device()
    except Exception as exc:      # printed, not raised: same contract as the reference
        print(exc)
