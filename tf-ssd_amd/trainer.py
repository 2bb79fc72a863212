"""Drop-in for the reference's ``trainer.py`` entry point: same flags and knobs, same call
order up to the training step (reference trainer.py:8-63).  What runs here: synthetic padded
batches -> GPU target assignment (``calculate_actual_outputs``) -> HIP forward -> loss values.
The optimisation step itself (backward convs, Adam, RCCL gradient all-reduce) is SURVEY.md 8f
row N1 and is not built in this round: the script evaluates ``steps`` batches and exits."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from ssd_loss import CustomLoss  # noqa: E402
from utils import bbox_utils, data_utils, io_utils, train_utils  # noqa: E402


def main(argv=None):
    args = io_utils.handle_args(argv)
    if args.handle_gpu:
        io_utils.handle_gpu_compatibility()

    batch_size = 32
    epochs = 150            # kept for parity with the reference's knobs; unused until N1 lands
    load_weights = False
    steps = int(os.environ.get("SSD_TRAINER_STEPS", "4"))
    backbone = args.backbone
    io_utils.is_valid_backbone(backbone)
    if backbone == "mobilenet_v2":
        from models.ssd_mobilenet_v2 import get_model, init_model
    else:
        from models.ssd_vgg16 import get_model, init_model
    hyper_params = train_utils.get_hyper_params(backbone)
    labels = ["bg"] + data_utils.get_labels()
    hyper_params["total_labels"] = len(labels)
    img_size = hyper_params["img_size"]

    train_data = data_utils.synthetic_dataset(steps * batch_size, batch_size, img_size, len(labels))
    ssd_model = get_model(hyper_params, max_batch=batch_size)
    ssd_custom_losses = CustomLoss(hyper_params["neg_pos_ratio"], hyper_params["loc_loss_alpha"])
    init_model(ssd_model)
    ssd_model_path = io_utils.get_model_path(backbone)
    if load_weights and os.path.exists(ssd_model_path):
        ssd_model.load_weights(ssd_model_path)
    prior_boxes = bbox_utils.generate_prior_boxes(hyper_params["feature_map_shapes"], hyper_params["aspect_ratios"])
    ssd_train_feed = train_utils.generator(train_data, prior_boxes, hyper_params)

    for step, (img, (actual_deltas, actual_labels)) in zip(range(steps), ssd_train_feed):
        pred_deltas, pred_labels = ssd_model(img)
        loc = ssd_custom_losses.loc_loss_fn(actual_deltas, pred_deltas).mean()
        conf = ssd_custom_losses.conf_loss_fn(actual_labels, pred_labels).mean()
        print("step %d  lr %.0e  loc_loss %.4f  conf_loss %.4f  positives/img %.1f" % (
            step, train_utils.scheduler(0), float(loc), float(conf),
            float((actual_labels[..., 1:] != 0).any(-1).float().sum(1).mean())))
    print("forward/target/loss evaluation done; the optimisation step (SURVEY.md 8f N1) is not built yet")


if __name__ == "__main__":
    main()
