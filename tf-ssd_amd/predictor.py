"""Drop-in for the reference's ``predictor.py``: same flags (``-handle-gpu``, ``--backbone``),
same knobs and call order (reference predictor.py:5-57).  VOC via tfds is not available, so
the test split is a seeded synthetic dataset of the same shape; trained weights are loaded
when ``trained/ssd_{backbone}_model_weights.h5`` exists, else seeded synthetic weights."""
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from utils import bbox_utils, data_utils, io_utils, train_utils, eval_utils  # noqa: E402
from models.decoder import get_decoder_model  # noqa: E402


def main(argv=None):
    args = io_utils.handle_args(argv)
    if args.handle_gpu:
        io_utils.handle_gpu_compatibility()

    batch_size = 32
    evaluate = False
    total_items = int(os.environ.get("SSD_SYNTHETIC_ITEMS", "128"))   # VOC2007 test has 4952
    backbone = args.backbone
    io_utils.is_valid_backbone(backbone)
    #
    if backbone == "mobilenet_v2":
        from models.ssd_mobilenet_v2 import get_model, init_model
    else:
        from models.ssd_vgg16 import get_model, init_model
    #
    hyper_params = train_utils.get_hyper_params(backbone)
    labels = ["bg"] + data_utils.get_labels()
    hyper_params["total_labels"] = len(labels)
    img_size = hyper_params["img_size"]

    test_data = list(data_utils.synthetic_dataset(total_items, batch_size, img_size, len(labels)))

    ssd_model = get_model(hyper_params, max_batch=batch_size)
    ssd_model_path = io_utils.get_model_path(backbone)
    if os.path.exists(ssd_model_path):
        ssd_model.load_weights(ssd_model_path)
    else:
        print("no trained weights at %s: using seeded synthetic weights" % ssd_model_path)
        data_utils.synthetic_weights(ssd_model)

    prior_boxes = bbox_utils.generate_prior_boxes(hyper_params["feature_map_shapes"], hyper_params["aspect_ratios"])
    ssd_decoder_model = get_decoder_model(ssd_model, prior_boxes, hyper_params)

    step_size = train_utils.get_step_size(total_items, batch_size)
    t0 = time.perf_counter()
    pred_bboxes, pred_labels, pred_scores = ssd_decoder_model.predict(test_data, steps=step_size, verbose=1)
    dt = time.perf_counter() - t0
    print("predicted %d images in %.3f s (%.1f images/sec incl. host transfers); mean detections/image %.1f" % (
        pred_bboxes.shape[0], dt, pred_bboxes.shape[0] / dt, float((pred_labels > 0).sum(-1).mean())))

    if evaluate:
        eval_utils.evaluate_predictions(test_data, pred_bboxes, pred_labels, pred_scores, labels, batch_size)
    return pred_bboxes, pred_labels, pred_scores


if __name__ == "__main__":
    main()
