"""Data side of the drop-in.  The reference's ``utils/data_utils.py`` loads PASCAL VOC via
tensorflow_datasets (not available offline: ``get_dataset`` raises) and converts / resizes every
image with TF ops; here ``preprocessing`` runs that conversion + bilinear resize as one HIP kernel
(SURVEY.md 8f N4).  Also kept: the VOC label list, the padded-batch conventions (gt boxes padded with 0, labels with
-1), and seeded synthetic generators shaped like the reference's batches."""
import numpy as np

VOC_LABELS = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
              "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa",
              "train", "tvmonitor"]


def preprocessing(image_data, final_height, final_width, augmentation_fn=None, evaluate=False):
    """reference utils/data_utils.py:7-30: uint8 image -> float32 [0,1] resized (bilinear) to
    ``(final_height, final_width)`` ON THE GPU (``ssd_preprocess``: one kernel, no float copy of
    the original image), labels shifted by +1 (0 is the background), difficult objects dropped when
    ``evaluate``.  ``image_data`` is the tfds-style dict ``{"image": uint8 [H,W,3], "objects":
    {"bbox": [G,4], "label": [G], "is_difficult": [G]}}``.  Returns (img device tensor, gt_boxes,
    gt_labels)."""
    import torch
    import ssd_hip as _h
    img = image_data["image"]
    img = img if isinstance(img, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(img))
    if img.dtype != torch.uint8 or img.dim() != 3:
        raise ValueError("image must be uint8 [H,W,C], got %s %s" % (img.dtype, tuple(img.shape)))
    src = img.to(_h.device()).contiguous()
    H, W, C = src.shape
    out = torch.empty((int(final_height), int(final_width), C), dtype=torch.float32, device=src.device)
    _h.check(_h.lib().ssd_preprocess(_h.ptr(src), 1, H, W, C, int(final_height), int(final_width), _h.ptr(out),
                                     _h.stream()), "preprocessing")
    gt_boxes = np.asarray(image_data["objects"]["bbox"], np.float32)
    gt_labels = (np.asarray(image_data["objects"]["label"]) + 1).astype(np.int32)
    if evaluate:
        not_diff = np.logical_not(np.asarray(image_data["objects"]["is_difficult"], bool))
        gt_boxes, gt_labels = gt_boxes[not_diff], gt_labels[not_diff]
    if augmentation_fn:
        out, gt_boxes = augmentation_fn(out, gt_boxes)
    return out, gt_boxes, gt_labels


def preprocess_batch(images_u8, final_height, final_width):
    """A batch of same-sized uint8 images [B,H,W,3] -> float32 [B,final_height,final_width,3] in one launch."""
    import torch
    import ssd_hip as _h
    x = images_u8 if isinstance(images_u8, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(images_u8))
    if x.dtype != torch.uint8 or x.dim() != 4:
        raise ValueError("images must be uint8 [B,H,W,C]")
    x = x.to(_h.device()).contiguous()
    B, H, W, C = x.shape
    out = torch.empty((B, int(final_height), int(final_width), C), dtype=torch.float32, device=x.device)
    _h.check(_h.lib().ssd_preprocess(_h.ptr(x), B, H, W, C, int(final_height), int(final_width), _h.ptr(out),
                                     _h.stream()), "preprocess_batch")
    return out


def get_labels(info=None):
    """reference utils/data_utils.py:61-69: ``info.features["labels"].names`` of a tfds info object; without one
    (tfds is not available here) the 20 VOC class names the reference's datasets carry."""
    if info is not None and hasattr(info, "features"):
        return info.features["labels"].names
    return list(VOC_LABELS)


def get_dataset(name, split, data_dir="~/tensorflow_datasets"):
    """reference utils/data_utils.py:32-45 -- tfds is not available in this build."""
    assert split in ["train", "train+validation", "validation", "test"]
    raise RuntimeError("tensorflow_datasets is not available; use synthetic_dataset() or feed arrays "
                       "[B,S,S,3] float32 in [0,1] directly")


def get_total_item_size(info, split):
    """reference utils/data_utils.py:47-59.  ``info``: a tfds info object (``info.splits[name].num_examples``, the
    reference's argument), or -- tfds is not available here -- a dict ``{"splits": {name: count}}`` or an int."""
    assert split in ["train", "train+validation", "validation", "test"]
    if isinstance(info, int):
        return info
    splits = info.splits if hasattr(info, "splits") else info["splits"]

    def count(name):
        v = splits[name]
        return v.num_examples if hasattr(v, "num_examples") else v
    if split == "train+validation":
        return count("train") + count("validation")
    return count(split)


def get_custom_imgs(custom_image_path):
    """reference utils/data_utils.py:80-91: the files directly inside ``custom_image_path`` (no recursion)."""
    import os
    img_paths = []
    for path, _dirs, filenames in os.walk(custom_image_path):
        for filename in sorted(filenames):
            img_paths.append(os.path.join(path, filename))
        break
    return img_paths


def custom_data_generator(img_paths, final_height, final_width):
    """reference utils/data_utils.py:93-108: every image opened with PIL and resized with LANCZOS on the host
    (the reference's choice for custom images -- dataset images take the bilinear ``preprocessing`` path),
    then uint8 -> float32 [0,1] (``tf.image.convert_image_dtype``) on the GPU (``ssd_preprocess`` at equal
    sizes is exactly that conversion).  ``*.npy`` files (uint8 [H,W,3]) are accepted as well.  Yields
    ``(img [final_height, final_width, 3] device tensor, gt_boxes [0,4], gt_labels [0])``."""
    from PIL import Image
    for img_path in img_paths:
        if img_path.endswith(".npy"):
            image = Image.fromarray(np.load(img_path))
        else:
            image = Image.open(img_path).convert("RGB")
        resized = np.ascontiguousarray(np.array(image.resize((final_width, final_height), Image.LANCZOS), dtype=np.uint8))
        img = preprocess_batch(resized[None], final_height, final_width)[0]
        yield img, np.zeros((0, 4), np.float32), np.zeros((0,), np.int32)


def padded_batch(items, batch_size, padding_values=None):
    """``Dataset.padded_batch(batch_size, padded_shapes=data_shapes, padding_values=...)`` of the reference
    scripts (predictor.py:43, trainer.py:33-34): consecutive ``(img, gt_boxes, gt_labels)`` items are stacked,
    ground truth padded to the longest of the batch with 0 / -1.  Images stay where they are (device tensors
    are stacked on the device)."""
    import torch
    pv = padding_values or get_padding_values()
    batch = []

    def flush():
        g = max([len(b[1]) for b in batch] + [1])
        gt = np.full((len(batch), g, 4), pv[1], np.float32)
        gl = np.full((len(batch), g), pv[2], np.int32)
        for i, (_, bb, ll) in enumerate(batch):
            gt[i, :len(bb)] = np.asarray(bb, np.float32).reshape(-1, 4)
            gl[i, :len(ll)] = np.asarray(ll, np.int32)
        imgs = [b[0] for b in batch]
        x = torch.stack(imgs) if isinstance(imgs[0], torch.Tensor) else np.stack(imgs)
        return x, gt, gl

    for it in items:
        batch.append(it)
        if len(batch) == batch_size:
            yield flush()
            batch = []
    if batch:
        yield flush()


def synthetic_voc_items(total_items, total_labels=21, seed=0):
    """Seeded stand-in for ``tfds.load("voc/2007")`` items: dicts ``{"image": uint8 [H,W,3] (VOC-like sizes, H / W
    in 300..500), "objects": {"bbox" [G,4] normalised, "label" [G] in 0..total_labels-2 (``preprocessing`` adds
    1), "is_difficult" [G]}}`` -- what ``preprocessing`` consumes (utils/data_utils.py:7-30)."""
    rng = np.random.default_rng(seed)
    for _ in range(total_items):
        h, w = int(rng.integers(300, 501)), int(rng.integers(300, 501))
        g = int(rng.integers(1, 9))
        c = rng.uniform(0.15, 0.85, (g, 2))
        sz = rng.uniform(0.08, 0.5, (g, 2))
        yield {"image": rng.integers(0, 256, (h, w, 3), dtype=np.uint8),
               "objects": {"bbox": np.clip(np.concatenate([c - sz / 2, c + sz / 2], -1), 0, 1).astype(np.float32),
                           "label": rng.integers(0, total_labels - 1, g).astype(np.int64),
                           "is_difficult": rng.random(g) < 0.2}}


def get_padding_values():
    """reference utils/data_utils.py:117-122: image 0, gt boxes 0, gt labels -1."""
    return (np.float32(0), np.float32(0), np.int32(-1))


def synthetic_images(batch, img_size=300, seed=0):
    """uint8->float32 [0,1] images like ``preprocessing`` yields (utils/data_utils.py:22)."""
    return np.random.default_rng(seed).random((batch, img_size, img_size, 3), dtype=np.float32)


def synthetic_gt(batch, max_boxes=16, total_labels=21, seed=3):
    rng = np.random.default_rng(seed)
    gt = np.zeros((batch, max_boxes, 4), np.float32)
    gl = -np.ones((batch, max_boxes), np.int32)
    for b in range(batch):
        g = int(rng.integers(1, max_boxes + 1))
        c = rng.uniform(0.1, 0.9, (g, 2))
        s = rng.uniform(0.05, 0.5, (g, 2))
        gt[b, :g] = np.clip(np.concatenate([c - s / 2, c + s / 2], -1), 0, 1).astype(np.float32)
        gl[b, :g] = rng.integers(1, total_labels, g)
    return gt, gl


def synthetic_dataset(total_items, batch_size, img_size=300, total_labels=21, seed=0):
    """Finite iterable of (img, gt_boxes, gt_labels) padded batches."""
    for i in range(0, total_items, batch_size):
        b = min(batch_size, total_items - i)
        gt, gl = synthetic_gt(b, total_labels=total_labels, seed=seed + 1000 + i)
        yield synthetic_images(b, img_size, seed + i), gt, gl


def synthetic_weights(model, seed=1, target_frac=0.05):
    """Seeded random weights for benchmarks (no pretrained weights offline): He-normal conv
    kernels, BatchNorm gamma 1+-0.1 / beta,mean +-0.1 / var in [0.5,1.5].  The label-head
    background bias is then calibrated ON THE DEVICE (bisection on the model's own output
    for one image) so that ~target_frac of the anchors carry a non-background probability
    above 0.5 -- with purely random weights NMS would have nothing to do."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in model.param_specs:
        var = name.rsplit("/", 1)[1]
        if var == "kernel":
            scale = 0.5 if "label_output" in name else (0.25 if "boxes_output" in name else 1.0)
            w[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] * shape[1] * shape[2])) * scale).astype(np.float32)
        elif var == "depthwise_kernel":
            w[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / 9.0)).astype(np.float32)
        elif var == "gamma":
            w[name] = rng.uniform(0.9, 1.1, shape).astype(np.float32)
        elif var in ("beta", "moving_mean"):
            w[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif var == "moving_variance":
            w[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif var == "scale":
            w[name] = np.full(shape, 20.0, np.float32)
        else:
            w[name] = rng.uniform(-0.05, 0.05, shape).astype(np.float32)
    model.set_weights(w)
    _, probs = model(synthetic_images(1, model.img_size, seed=0))
    logp = np.log(np.maximum(probs[0].double().cpu().numpy(), 1e-300))

    def frac(t, g=1.0):
        lg = logp * g
        lg[:, 0] += t
        e = np.exp(lg - lg.max(-1, keepdims=True))
        p = e / e.sum(-1, keepdims=True)
        return float(((p.argmax(-1) != 0) & (p.max(-1) > 0.5)).mean())
    # log-probabilities are the logits up to a per-row constant (valid while the softmax is not
    # saturated, which holds for the un-gained random heads).  If the class spread is too small
    # for any class to pass 0.5 (VGG16), scale the label kernels and biases by g first: the
    # logits scale by g exactly.
    g = 1.0
    while frac(-1e4, g) < 3 * target_frac and g < 64:
        g *= 2.0
    lo, hi = -50.0 * g, 50.0 * g
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if frac(mid, g) > target_frac:
            lo = mid
        else:
            hi = mid
    L = model.total_labels
    upd = {}
    for i in range(1, 7):
        b = w["%d_conv_label_output/bias" % i] * np.float32(g)
        b[0::L] += np.float32(0.5 * (lo + hi))
        w["%d_conv_label_output/bias" % i] = b
        upd["%d_conv_label_output/bias" % i] = b
        if g != 1.0:
            k = w["%d_conv_label_output/kernel" % i] * np.float32(g)
            w["%d_conv_label_output/kernel" % i] = k
            upd["%d_conv_label_output/kernel" % i] = k
    model.set_weights(upd)
    return w
