"""Host-side mirror of the reference's ``predictor.py`` (predictor.py:5-57): the same CLI
(``-handle-gpu``, ``--backbone``), the same knobs (batch 32, ``evaluate`` switch, "bg" + VOC
labels) and the same call order -- hyper-parameters, model, weights, priors, decoder model,
``predict`` over the test split, optional VOC07 mAP.

Offline differences: VOC through tfds is not available, so the test split is a seeded synthetic stand-in
for its items (``SSD_SYNTHETIC_ITEMS`` uint8 images of VOC-like sizes with boxes / labels / difficult flags,
default 128; VOC2007 test has 4952) that goes through the same ``preprocessing`` (GPU convert + bilinear
resize) -> padded batches -> ``predict`` -> optional ``evaluate_predictions``; ``use_custom_images`` reads
``custom_image_path`` like the reference (PIL + LANCZOS);
trained weights are loaded when ``trained/ssd_<backbone>_model_weights.h5`` exists, otherwise
seeded synthetic weights are used and that is said on stdout."""
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import ssd_hip  # noqa: E402
ssd_hip.configure_serving()        # the serving entry point opts in to three lanes / three hardware queues before the runtime starts
from utils import bbox_utils, data_utils, eval_utils, io_utils, train_utils  # noqa: E402
from models.decoder import get_decoder_model  # noqa: E402

# the reference's script-level knobs (predictor.py:9-12), same names and defaults
batch_size = 32
evaluate = False
use_custom_images = False
custom_image_path = "data/images/"


def _model_factory(backbone):
    if backbone == "mobilenet_v2":
        from models.ssd_mobilenet_v2 import get_model
    else:
        from models.ssd_vgg16 import get_model
    return get_model


def _load_or_synthesise_weights(model, backbone):
    path = io_utils.get_model_path(backbone)
    if os.path.exists(path):
        model.load_weights(path)
    else:
        print("no trained weights at %s: using seeded synthetic weights" % path)
        data_utils.synthetic_weights(model)


def main(argv=None, **knobs):
    """``knobs`` override the module-level switches for one call (``evaluate``, ``use_custom_images``,
    ``custom_image_path``, ``batch_size``)."""
    args = io_utils.handle_args(argv)
    if args.handle_gpu:
        io_utils.handle_gpu_compatibility()
    io_utils.is_valid_backbone(args.backbone)
    bs = int(knobs.get("batch_size", batch_size))
    do_eval = bool(knobs.get("evaluate", evaluate))
    custom = bool(knobs.get("use_custom_images", use_custom_images))
    custom_path = knobs.get("custom_image_path", custom_image_path)

    labels = ["bg"] + data_utils.get_labels()
    hyper_params = train_utils.get_hyper_params(args.backbone)
    hyper_params["total_labels"] = len(labels)
    img_size = hyper_params["img_size"]
    padding_values = data_utils.get_padding_values()

    if custom:                                            # predictor.py:35-39
        img_paths = data_utils.get_custom_imgs(custom_path)
        total_items = len(img_paths)
        items = data_utils.custom_data_generator(img_paths, img_size, img_size)
    else:                                                 # predictor.py:22-23, 40-41 (voc/2007 test through tfds there)
        total_items = int(os.environ.get("SSD_SYNTHETIC_ITEMS", "128"))
        raw = data_utils.synthetic_voc_items(total_items, len(labels))
        items = (data_utils.preprocessing(x, img_size, img_size, evaluate=do_eval) for x in raw)
    # predictor.py:43 -- materialised: predict() and evaluate_predictions() both walk it
    test_data = list(data_utils.padded_batch(items, bs, padding_values))

    ssd_model = _model_factory(args.backbone)(hyper_params, max_batch=bs)
    _load_or_synthesise_weights(ssd_model, args.backbone)
    prior_boxes = bbox_utils.generate_prior_boxes(hyper_params["feature_map_shapes"], hyper_params["aspect_ratios"])
    ssd_decoder_model = get_decoder_model(ssd_model, prior_boxes, hyper_params)

    t0 = time.perf_counter()
    boxes, classes, scores = ssd_decoder_model.predict(
        test_data, steps=train_utils.get_step_size(total_items, bs), verbose=1)
    dt = time.perf_counter() - t0
    print("predicted %d images in %.3f s (%.1f images/sec incl. host transfers); mean detections/image %.1f" % (
        boxes.shape[0], dt, boxes.shape[0] / max(dt, 1e-9), float((classes > 0).sum(-1).mean()) if boxes.shape[0] else 0.0))

    if do_eval:                                           # predictor.py:54-55
        stats = eval_utils.evaluate_predictions(test_data, boxes, classes, scores, labels, bs)
        return boxes, classes, scores, stats
    # predictor.py:56-57 draws the boxes (utils/drawing_utils.py: out of scope)
    return boxes, classes, scores


if __name__ == "__main__":
    main()
