"""Drop-in for the reference's ``trainer.py`` entry point (reference trainer.py:8-76): same
flags and knobs (batch 32, 150 epochs, ``load_weights``), same call order -- hyper-parameters,
data, model, ``compile(Adam(1e-3), [loc_loss_fn, conf_loss_fn])``, ``init_model``, prior boxes,
target-encoding generators, ``fit`` with the three callbacks: ``ModelCheckpoint(monitor=
"val_loss", save_best_only, save_weights_only)``, ``LearningRateScheduler(train_utils.scheduler)``
and a scalar log in place of TensorBoard.

Every step is native: GPU target assignment -> training-mode forward -> HIP loss -> backward ->
(RCCL all-reduce of the flat gradient when launched with torchrun, one rank per GPU) -> Adam.
Offline differences: VOC through tfds is unavailable, so the splits are seeded synthetic padded
batches (``SSD_TRAINER_ITEMS`` training images per epoch); ``SSD_TRAINER_EPOCHS`` /
``SSD_TRAINER_STEPS`` / ``SSD_TRAINER_BATCH`` shorten a run (tests, smoke)."""
import json
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import parallel  # noqa: E402
from ssd_loss import CustomLoss  # noqa: E402
from utils import bbox_utils, data_utils, io_utils, train_utils  # noqa: E402


def fit(model, train_feed, steps_per_epoch, val_feed, validation_steps, epochs, model_path, log_path, rank=0):
    """Keras ``Model.fit`` with the reference's callbacks (trainer.py:65-76): per epoch the
    LearningRateScheduler sets the rate, ``steps_per_epoch`` optimisation steps run, then
    ``validation_steps`` validation batches; the weights are saved whenever ``val_loss`` improves."""
    history = {"loss": [], "loc_loss": [], "conf_loss": [], "val_loss": [], "lr": []}
    best = float("inf")
    log = None
    if rank == 0 and log_path:
        os.makedirs(log_path, exist_ok=True)
        log = open(os.path.join(log_path, "scalars.jsonl"), "w")     # TensorBoard callback stand-in
    for epoch in range(epochs):
        lr = train_utils.scheduler(epoch)
        t0 = time.perf_counter()
        tot = loc = conf = 0.0
        for _ in range(steps_per_epoch):
            img, targets = next(train_feed)
            l, a, b = model.train_on_batch(img, targets, learning_rate=lr)
            tot += l; loc += a; conf += b
        n = max(steps_per_epoch, 1)
        vtot = 0.0
        for _ in range(validation_steps):
            img, targets = next(val_feed)
            vtot += model.evaluate_on_batch(img, targets)[0]
        val = parallel.mean_over_ranks(vtot / max(validation_steps, 1))
        rec = {"epoch": epoch, "loss": tot / n, "loc_loss": loc / n, "conf_loss": conf / n, "val_loss": val, "lr": lr}
        for k in history:
            history[k].append(rec[k])
        if rank == 0:
            print("Epoch %d/%d - %.1fs - loss: %.4f - loc_loss: %.4f - conf_loss: %.4f - val_loss: %.4f - lr: %.0e" % (
                epoch + 1, epochs, time.perf_counter() - t0, rec["loss"], rec["loc_loss"], rec["conf_loss"], val, lr),
                flush=True)
            if log:
                log.write(json.dumps(rec) + "\n")
                log.flush()
            if val < best:          # ModelCheckpoint(monitor="val_loss", save_best_only=True, save_weights_only=True)
                model.save_weights(model_path)
        best = min(best, val)
    if log:
        log.close()
    return history


def main(argv=None):
    args = io_utils.handle_args(argv)
    if args.handle_gpu:
        io_utils.handle_gpu_compatibility()
    rank, _, world = parallel.init_distributed()

    batch_size = int(os.environ.get("SSD_TRAINER_BATCH", "32"))
    epochs = int(os.environ.get("SSD_TRAINER_EPOCHS", "150"))
    load_weights = False
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

    train_total_items = int(os.environ.get("SSD_TRAINER_ITEMS", "512"))
    val_total_items = max(batch_size, train_total_items // 8)
    step_size_train = int(os.environ.get("SSD_TRAINER_STEPS", "0")) or train_utils.get_step_size(train_total_items, batch_size)
    step_size_val = min(2, train_utils.get_step_size(val_total_items, batch_size))
    # every rank draws its own shard of the synthetic stream (batch data-parallel)
    train_data = list(data_utils.synthetic_dataset(step_size_train * batch_size, batch_size, img_size, len(labels),
                                                   seed=1000 * rank))
    val_data = list(data_utils.synthetic_dataset(step_size_val * batch_size, batch_size, img_size, len(labels),
                                                 seed=777 + 1000 * rank))

    ssd_model = get_model(hyper_params, max_batch=batch_size)
    ssd_custom_losses = CustomLoss(hyper_params["neg_pos_ratio"], hyper_params["loc_loss_alpha"])
    ssd_model.compile(learning_rate=1e-3, loss=[ssd_custom_losses.loc_loss_fn, ssd_custom_losses.conf_loss_fn])
    init_model(ssd_model)
    ssd_model_path = io_utils.get_model_path(backbone)
    if load_weights:
        ssd_model.load_weights(ssd_model_path)
    ssd_log_path = io_utils.get_log_path(backbone)
    prior_boxes = bbox_utils.generate_prior_boxes(hyper_params["feature_map_shapes"], hyper_params["aspect_ratios"])
    # reference trainer.py:42: the TRAINING stream is augmented (fresh draws every epoch), the validation stream is not.
    # SSD_TRAINER_AUGMENT=0 turns it off (deterministic smoke runs).
    if os.environ.get("SSD_TRAINER_AUGMENT", "1") != "0":
        import augmentation
        augmentation.seed(4242 + rank)
        train_data = augmentation.augmented(train_data)
    ssd_train_feed = train_utils.generator(train_data, prior_boxes, hyper_params)
    ssd_val_feed = train_utils.generator(val_data, prior_boxes, hyper_params)

    return fit(ssd_model, ssd_train_feed, step_size_train, ssd_val_feed, step_size_val, epochs, ssd_model_path,
               ssd_log_path, rank)


if __name__ == "__main__":
    main()
