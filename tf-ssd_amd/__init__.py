"""MI355X-native SSD hot path behind the FurkanOM/tf-ssd Python surface.

The directory mirrors the reference's root layout (``utils/``, ``models/``,
``trainer.py``, ``predictor.py``) so that code written against the reference
(``from utils import bbox_utils``; ``from models.decoder import get_decoder_model``)
runs unchanged with this directory on ``sys.path`` -- which importing this package
arranges.  All compute goes through ``libssd_hip.so`` (``ssd_hip.py``); there is no CPU
fallback.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
