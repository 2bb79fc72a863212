"""Host-side mirror of the reference's ``predictor.py`` (predictor.py:5-57): the same CLI
(``-handle-gpu``, ``--backbone``), the same knobs (batch 32, ``evaluate`` switch, "bg" + VOC
labels) and the same call order -- hyper-parameters, model, weights, priors, decoder model,
``predict`` over the test split, optional VOC07 mAP.

Offline differences: VOC through tfds is not available, so the test split is a seeded synthetic
dataset of the same shapes (``SSD_SYNTHETIC_ITEMS`` images, default 128; VOC2007 test has 4952);
trained weights are loaded when ``trained/ssd_<backbone>_model_weights.h5`` exists, otherwise
seeded synthetic weights are used and that is said on stdout."""
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from utils import bbox_utils, data_utils, eval_utils, io_utils, train_utils  # noqa: E402
from models.decoder import get_decoder_model  # noqa: E402

BATCH_SIZE = 32
EVALUATE = False


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


def main(argv=None):
    args = io_utils.handle_args(argv)
    if args.handle_gpu:
        io_utils.handle_gpu_compatibility()
    io_utils.is_valid_backbone(args.backbone)

    labels = ["bg"] + data_utils.get_labels()
    hyper_params = train_utils.get_hyper_params(args.backbone)
    hyper_params["total_labels"] = len(labels)
    n_items = int(os.environ.get("SSD_SYNTHETIC_ITEMS", "128"))
    test_data = list(data_utils.synthetic_dataset(n_items, BATCH_SIZE, hyper_params["img_size"], len(labels)))

    ssd_model = _model_factory(args.backbone)(hyper_params, max_batch=BATCH_SIZE)
    _load_or_synthesise_weights(ssd_model, args.backbone)
    prior_boxes = bbox_utils.generate_prior_boxes(hyper_params["feature_map_shapes"], hyper_params["aspect_ratios"])
    # two batches in flight (H2D copy + backbone of batch n+1 beside heads / decode / NMS of batch n)
    ssd_decoder_model = get_decoder_model(ssd_model, prior_boxes, hyper_params,
                                          lanes=int(os.environ.get("SSD_HIP_LANES", "2")))

    t0 = time.perf_counter()
    boxes, classes, scores = ssd_decoder_model.predict(
        test_data, steps=train_utils.get_step_size(n_items, BATCH_SIZE), verbose=1)
    dt = time.perf_counter() - t0
    print("predicted %d images in %.3f s (%.1f images/sec incl. host transfers); mean detections/image %.1f" % (
        boxes.shape[0], dt, boxes.shape[0] / dt, float((classes > 0).sum(-1).mean())))

    if EVALUATE:
        eval_utils.evaluate_predictions(test_data, boxes, classes, scores, labels, BATCH_SIZE)
    return boxes, classes, scores


if __name__ == "__main__":
    main()
