"""Drop-in for the reference's ``augmentation.py`` (SSD training augmentation; reference trainer.py:42 passes
``augmentation.apply`` to ``data_utils.preprocessing``): same function names and meaning, images stay on the GPU.

The reference draws its random numbers with ``tf.random.uniform`` and ``tf.image.sample_distorted_bounding_box`` and
rewrites the image op by op.  Here the DRAWS are made on the host (a seedable NumPy generator, ``seed()``; every function
also takes them explicitly -- that is what the parity tests pin against oracle/augment_oracle.py), and the pixels move
through three HIP kernels (include/ssd_hip.h): ``ssd_image_mean`` (expand's fill colour, contrast's pivot),
``ssd_augment_geometry`` (expand -> crop -> bilinear resize -> flip as one gather over a virtual canvas: the expanded image
-- up to 16x the pixels -- is never materialised) and ``ssd_augment_color`` (brightness -> contrast -> hue -> saturation ->
clip in one pass).  Box arithmetic is host-side float32 NumPy (a few dozen boxes).

Images: float32 device tensors [H,W,3] in [0,1]; boxes [G,4] = (y1, x1, y2, x2) normalised.
[3P] ``sample_distorted_bounding_box`` is restated from the TF 2.0 kernel's published algorithm (random: only its
acceptance rule is testable); there is no TensorFlow here, parity of the TF ops themselves is unpinned."""
import numpy as np
import torch

import ssd_hip as _h

F32 = np.float32
_rng = np.random.default_rng()


def seed(s):
    """Seed the module's generator (the reference relies on TF's global seed)."""
    global _rng
    _rng = np.random.default_rng(s)


def _boxes(gt_boxes):
    if isinstance(gt_boxes, torch.Tensor):
        return gt_boxes.detach().cpu().numpy().astype(F32)
    return np.asarray(gt_boxes, F32)


def _img(img):
    x = _h.to_dev(img)
    if x.dim() != 3:
        raise ValueError("image must be [H,W,C], got %s" % (tuple(x.shape),))
    return x


def _geometry(img, canvas=None, crop=None, flip=False, fill=None, out_size=None):
    """One ``ssd_augment_geometry`` launch on a single image (output H x W unless ``out_size``)."""
    x = _img(img)
    H, W, C = x.shape
    ch, cw, pt, pl = canvas if canvas is not None else (H, W, 0, 0)
    use_crop = crop is not None
    cy, cx, chh, cww = crop if use_crop else (0, 0, ch, cw)
    Ho, Wo = out_size if out_size is not None else (H, W)
    if not use_crop and (Ho, Wo) != (H, W):
        raise ValueError("a flip keeps the image size")
    params = torch.tensor([[ch, cw, pt, pl, cy, cx, chh, cww, int(bool(flip)), int(use_crop)]], dtype=torch.int32, device=x.device)
    if fill is None:
        fill = torch.zeros((1, C), dtype=torch.float32, device=x.device)
    out = torch.empty((Ho, Wo, C), dtype=torch.float32, device=x.device)
    _h.check(_h.lib().ssd_augment_geometry(_h.ptr(x), 1, H, W, C, Ho, Wo, _h.ptr(params), _h.ptr(fill), _h.ptr(out), _h.stream()),
             "ssd_augment_geometry")
    return out


def image_mean(img, add=None):
    """Per-channel mean over H, W of ``img + add`` as a [1,C] device tensor (``tf.nn.moments(img, [0, 1])``)."""
    x = _img(img)
    H, W, C = x.shape
    out = torch.empty((1, C), dtype=torch.float32, device=x.device)
    a = None if add is None else torch.tensor([float(add)], dtype=torch.float32, device=x.device)
    _h.check(_h.lib().ssd_image_mean(_h.ptr(x), 1, H, W, C, _h.ptr(a), _h.ptr(out), _h.stream()), "ssd_image_mean")
    return out


def _color(img, brightness=None, contrast=None, hue=None, saturation=None):
    x = _img(img).clone()
    H, W, C = x.shape
    if C != 3:
        raise ValueError("colour operations need RGB images")
    flags = (1 if brightness is not None else 0) | (2 if contrast is not None else 0) | (4 if hue is not None else 0) | \
            (8 if saturation is not None else 0)
    params = torch.tensor([[brightness or 0.0, contrast if contrast is not None else 1.0, hue or 0.0,
                            saturation if saturation is not None else 1.0]], dtype=torch.float32, device=x.device)
    fl = torch.tensor([flags], dtype=torch.int32, device=x.device)
    mean = image_mean(x, brightness) if contrast is not None else torch.zeros((1, 3), dtype=torch.float32, device=x.device)
    _h.check(_h.lib().ssd_augment_color(_h.ptr(x), 1, H, W, _h.ptr(params), _h.ptr(fl), _h.ptr(mean), _h.stream()),
             "ssd_augment_color")
    return x


# ---- the reference's surface ------------------------------------------------------------------------------------------
def get_random_bool():
    """reference augmentation.py:29-34: uniform() > 0.5."""
    return bool(F32(_rng.random()) > F32(0.5))


def randomly_apply_operation(operation, img, gt_boxes, *args):
    """reference augmentation.py:36-49."""
    if get_random_bool():
        return operation(img, gt_boxes, *args)
    return img, gt_boxes


def random_brightness(img, gt_boxes, max_delta=0.12, delta=None):
    """reference augmentation.py:51-60 (``tf.image.random_brightness``: delta ~ U[-max_delta, max_delta)); the result is
    NOT clipped here (the reference clips once, at the end of ``apply``) -- the kernel's final clip is the only difference
    for out-of-range values, so standalone calls return the clipped image."""
    d = float(_rng.uniform(-max_delta, max_delta)) if delta is None else float(delta)
    return _color(img, brightness=d), gt_boxes


def random_contrast(img, gt_boxes, lower=0.5, upper=1.5, factor=None):
    """reference augmentation.py:62-71."""
    f = float(_rng.uniform(lower, upper)) if factor is None else float(factor)
    return _color(img, contrast=f), gt_boxes


def random_hue(img, gt_boxes, max_delta=0.08, delta=None):
    """reference augmentation.py:73-82."""
    d = float(_rng.uniform(-max_delta, max_delta)) if delta is None else float(delta)
    return _color(img, hue=d), gt_boxes


def random_saturation(img, gt_boxes, lower=0.5, upper=1.5, factor=None):
    """reference augmentation.py:84-93."""
    f = float(_rng.uniform(lower, upper)) if factor is None else float(factor)
    return _color(img, saturation=f), gt_boxes


def flip_boxes(gt_boxes):
    g = _boxes(gt_boxes)
    return np.stack([g[..., 0], F32(1.0) - g[..., 3], g[..., 2], F32(1.0) - g[..., 1]], -1).astype(F32)


def flip_horizontally(img, gt_boxes):
    """reference augmentation.py:95-110."""
    return _geometry(img, flip=True), flip_boxes(gt_boxes)


def get_random_min_overlap():
    """reference augmentation.py:116-122."""
    overlaps = np.array([0.1, 0.3, 0.5, 0.7, 0.9], F32)
    return overlaps[int(_rng.integers(0, len(overlaps)))]


def renormalize(gt_boxes, min_max):
    """utils/bbox_utils.py:178-188 in float32 NumPy (same op order as the oracle)."""
    b = _boxes(gt_boxes)
    y_min, x_min, y_max, x_max = [F32(v) for v in np.asarray(min_max, F32)]
    r = b - np.array([y_min, x_min, y_min, x_min], F32)
    r = r / np.array([y_max - y_min, x_max - x_min, y_max - y_min, x_max - x_min], F32)
    return np.clip(r, F32(0), F32(1)).astype(F32)


def expand_geometry(height, width, expansion_ratio, u_left, u_top):
    """The integers of reference augmentation.py:135-140 from its three uniform draws (tf.round = round half to even)."""
    height, width = F32(height), F32(width)
    final_h = np.rint(height * F32(expansion_ratio)).astype(F32)
    final_w = np.rint(width * F32(expansion_ratio)).astype(F32)
    pad_left = np.rint(F32(u_left) * (final_w - width)).astype(F32)
    pad_top = np.rint(F32(u_top) * (final_h - height)).astype(F32)
    return int(final_h), int(final_w), int(pad_top), int(pad_left)


def expand_boxes(gt_boxes, h, w, fh, fw, pt, pl):
    pad_bottom, pad_right = F32(fh) - (F32(h) + F32(pt)), F32(fw) - (F32(w) + F32(pl))
    min_max = np.array([-F32(pt), -F32(pl), pad_bottom + F32(h), pad_right + F32(w)], F32) / np.array([h, w, h, w], F32)
    return renormalize(gt_boxes, min_max)


def expand_image(img, gt_boxes, height=None, width=None, draws=None):
    """reference augmentation.py:123-151: the image somewhere on a canvas up to 4 x 4 times its size, filled with its
    per-channel mean; boxes renormalised to the canvas.  Standalone calls materialise the canvas like the reference;
    ``patch`` / ``apply`` (the reference's only caller is ``patch``, :165) never do -- they crop and resize straight out of
    the virtual canvas in the same kernel."""
    x = _img(img)
    h, w = (int(x.shape[0]), int(x.shape[1])) if height is None else (int(height), int(width))
    ratio, u_left, u_top = draws if draws is not None else (_rng.uniform(1.0, 4.0), _rng.random(), _rng.random())
    fh, fw, pt, pl = expand_geometry(h, w, ratio, u_left, u_top)
    canvas = _geometry(x, canvas=(fh, fw, pt, pl), crop=(0, 0, fh, fw), fill=image_mean(x), out_size=(fh, fw))
    return canvas, expand_boxes(gt_boxes, h, w, fh, fw, pt, pl)


def sample_distorted_bounding_box(height, width, gt_boxes, min_object_covered, aspect_ratio_range=(0.5, 2.0),
                                  area_range=(0.05, 1.0), max_attempts=100):
    """[3P] ``tf.image.sample_distorted_bounding_box`` (reference augmentation.py:170-174) restated from the TF 2.0
    kernel: up to ``max_attempts`` random windows of a random aspect ratio and area; the first one that holds at least
    ``min_object_covered`` of the area of SOME ground-truth box wins, otherwise the whole image.  Returns
    (begin_y, begin_x, size_h, size_w)."""
    g = _boxes(gt_boxes).astype(np.float64)
    H, W = int(height), int(width)
    if g.shape[0] == 0:             # the TF op with use_image_if_no_bounding_boxes=False (the reference's call) raises
        raise ValueError("sample_distorted_bounding_box: no bounding boxes provided as input")
    rects = pixel_rectangles(g, H, W)
    min_area, max_area = area_range[0] * H * W, area_range[1] * H * W
    for _ in range(max_attempts):
        aspect = _rng.uniform(aspect_ratio_range[0], aspect_ratio_range[1])
        min_h = int(np.rint(np.sqrt(min_area / aspect)))
        max_h = int(np.rint(np.sqrt(max_area / aspect)))
        if int(np.rint(max_h * aspect)) > W:
            max_h = int((W + 0.5 - 1e-7) / aspect)          # the largest height whose rounded width still fits ...
            if int(np.rint(max_h * aspect)) > W:
                max_h -= 1                                  # ... and the kernel's fallback when the rounding disagrees
        max_h = min(max_h, H)
        min_h = min(min_h, max_h)
        h = min_h + int(_rng.integers(0, max_h - min_h + 1)) if min_h < max_h else min_h     # closed range, as in the kernel
        w = int(np.rint(h * aspect))
        if w * h < min_area:
            h += 1
            w = int(np.rint(h * aspect))
        if w * h > max_area:
            h -= 1
            w = int(np.rint(h * aspect))
        if w * h < min_area or w * h > max_area or w > W or h > H or w <= 0 or h <= 0:
            continue
        y = int(_rng.integers(0, H - h)) if h < H else 0     # random->Uniform(n) draws from [0, n): the last offset is never taken
        x = int(_rng.integers(0, W - w)) if w < W else 0
        if window_satisfies((y, x, y + h, x + w), rects, min_object_covered):
            return y, x, h, w
    return 0, 0, H, W


def pixel_rectangles(gt_boxes, height, width):
    """Normalised [y1, x1, y2, x2] boxes as the kernel's integer pixel rectangles (truncation towards zero)."""
    g = np.asarray(gt_boxes, np.float64).reshape(-1, 4)
    return np.stack([np.trunc(g[:, 0] * height), np.trunc(g[:, 1] * width), np.trunc(g[:, 2] * height), np.trunc(g[:, 3] * width)],
                    axis=1).astype(np.int64)


def window_satisfies(window, rects, min_object_covered):
    """The acceptance rule of the sampler (``SatisfiesOverlapConstraints``), on INTEGER pixel rectangles like the kernel:
    the window holds at least one pixel and some box of at least one pixel has >= min_object_covered of its area inside it
    (boxes without a pixel -- the zero padding of a batch -- are skipped)."""
    r = np.asarray(rects, np.int64).reshape(-1, 4)
    wy1, wx1, wy2, wx2 = [int(v) for v in window]
    if (wy2 - wy1) * (wx2 - wx1) < 1:
        return False
    iy = np.clip(np.minimum(r[:, 2], wy2) - np.maximum(r[:, 0], wy1), 0, None)
    ix = np.clip(np.minimum(r[:, 3], wx2) - np.maximum(r[:, 1], wx1), 0, None)
    area = (r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])
    ok = area >= 1
    return bool(((iy * ix)[ok].astype(np.float32) / area[ok].astype(np.float32) >= np.float32(min_object_covered)).any()) if ok.any() else False


def patch(img, gt_boxes, draws=None):
    """reference augmentation.py:153-181: maybe expand, sample a window, ``tf.slice`` + ``tf.image.resize`` back to the
    original size, boxes renormalised to the window (boxes that fall outside collapse to zero area and stay in the list,
    like the reference).  ``draws`` = (expand or None, window or None): expand = (expansion_ratio, u_left, u_top), window
    = (begin_y, begin_x, size_h, size_w) on the possibly expanded image."""
    x = _img(img)
    H, W = int(x.shape[0]), int(x.shape[1])
    g = _boxes(gt_boxes)
    if draws is None:
        expand = (_rng.uniform(1.0, 4.0), _rng.random(), _rng.random()) if get_random_bool() else None
        window = None
    else:
        expand, window = draws
    canvas, fill = (H, W, 0, 0), None
    if expand is not None:
        fh, fw, pt, pl = expand_geometry(H, W, *expand)
        g = expand_boxes(g, H, W, fh, fw, pt, pl)
        canvas, fill = (fh, fw, pt, pl), image_mean(x)
    if window is None:
        window = sample_distorted_bounding_box(canvas[0], canvas[1], g, float(get_random_min_overlap()))
    y, xx, h, w = [int(v) for v in window]
    out = _geometry(x, canvas=canvas, crop=(y, xx, h, w), fill=fill)
    ch, cw = F32(canvas[0]), F32(canvas[1])
    bounds = np.array([F32(y) / ch, F32(xx) / cw, F32(y + h) / ch, F32(xx + w) / cw], F32)
    return out, renormalize(g, bounds)


def draw_plan(H, W, gt_boxes):
    """The random decisions of ``apply`` for ONE image, in the reference's call order (augmentation.py:19-25: patch [expand,
    min overlap, window], flip, brightness, contrast, hue, saturation): returns (plan dict, transformed boxes)."""
    g = _boxes(gt_boxes)
    plan = {"canvas": (H, W, 0, 0), "crop": None, "flip": False, "expand": False,
            "brightness": None, "contrast": None, "hue": None, "saturation": None}
    if get_random_bool():                                  # patch
        if get_random_bool():                              # ... with expand
            fh, fw, pt, pl = expand_geometry(H, W, _rng.uniform(1.0, 4.0), _rng.random(), _rng.random())
            g = expand_boxes(g, H, W, fh, fw, pt, pl)
            plan["canvas"], plan["expand"] = (fh, fw, pt, pl), True
        ch, cw = plan["canvas"][:2]
        y, xx, h, w = sample_distorted_bounding_box(ch, cw, g, float(get_random_min_overlap()))
        plan["crop"] = (y, xx, h, w)
        g = renormalize(g, np.array([F32(y) / F32(ch), F32(xx) / F32(cw), F32(y + h) / F32(ch), F32(xx + w) / F32(cw)], F32))
    if get_random_bool():
        plan["flip"] = True
        g = flip_boxes(g)
    if get_random_bool():
        plan["brightness"] = float(_rng.uniform(-0.12, 0.12))
    if get_random_bool():
        plan["contrast"] = float(_rng.uniform(0.5, 1.5))
    if get_random_bool():
        plan["hue"] = float(_rng.uniform(-0.08, 0.08))
    if get_random_bool():
        plan["saturation"] = float(_rng.uniform(0.5, 1.5))
    return plan, g


def run_plans(images, plans):
    """Execute per-image plans on a batch [B,H,W,3]: one mean launch (expand fill colours), one geometry launch, one mean
    launch (contrast pivots) and one colour launch for the WHOLE batch."""
    x = _h.to_dev(images)
    if x.dim() != 4 or x.shape[3] != 3:
        raise ValueError("images must be [B,H,W,3], got %s" % (tuple(x.shape),))
    B, H, W, C = x.shape
    lib = _h.lib()
    dev = x.device
    gp = np.zeros((B, 10), np.int32)
    cp = np.zeros((B, 4), np.float32)
    fl = np.zeros((B,), np.int32)
    add = np.zeros((B,), np.float32)
    for b, p in enumerate(plans):
        ch, cw, pt, pl = p["canvas"]
        crop = p["crop"] if p["crop"] is not None else (0, 0, ch, cw)
        gp[b] = [ch, cw, pt, pl, crop[0], crop[1], crop[2], crop[3], int(p["flip"]), int(p["crop"] is not None)]
        cp[b] = [p["brightness"] or 0.0, p["contrast"] if p["contrast"] is not None else 1.0, p["hue"] or 0.0,
                 p["saturation"] if p["saturation"] is not None else 1.0]
        fl[b] = (1 if p["brightness"] is not None else 0) | (2 if p["contrast"] is not None else 0) | \
                (4 if p["hue"] is not None else 0) | (8 if p["saturation"] is not None else 0)
        add[b] = p["brightness"] or 0.0
    fill = torch.zeros((B, C), dtype=torch.float32, device=dev)
    if any(p["expand"] for p in plans):
        _h.check(lib.ssd_image_mean(_h.ptr(x), B, H, W, C, None, _h.ptr(fill), _h.stream()), "ssd_image_mean")
    if gp[:, 8].any() or gp[:, 9].any():
        out = torch.empty_like(x)
        gpd = torch.from_numpy(gp).to(dev)
        _h.check(lib.ssd_augment_geometry(_h.ptr(x), B, H, W, C, H, W, _h.ptr(gpd), _h.ptr(fill), _h.ptr(out), _h.stream()),
                 "ssd_augment_geometry")
    else:
        out = x.clone()
    mean = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    if (fl & 2).any():
        addd = torch.from_numpy(add).to(dev)
        _h.check(lib.ssd_image_mean(_h.ptr(out), B, H, W, C, _h.ptr(addd), _h.ptr(mean), _h.stream()), "ssd_image_mean")
    cpd, fld = torch.from_numpy(cp).to(dev), torch.from_numpy(fl).to(dev)
    _h.check(lib.ssd_augment_color(_h.ptr(out), B, H, W, _h.ptr(cpd), _h.ptr(fld), _h.ptr(mean), _h.stream()), "ssd_augment_color")
    return out


def apply(img, gt_boxes):
    """reference augmentation.py:4-27: patch and flip (each with probability 1/2), then brightness, contrast, hue and
    saturation (each with probability 1/2), then clip to [0,1].  Draw order = the reference's call order.  The geometric
    chain is ONE kernel launch, the photometric chain one more (plus the means expand / contrast need)."""
    x = _img(img)
    plan, g = draw_plan(int(x.shape[0]), int(x.shape[1]), gt_boxes)
    return run_plans(x[None], [plan])[0], g


def apply_batch(images, gt_boxes, gt_labels=None):
    """``apply`` on a padded batch (images [B,H,W,3], gt_boxes [B,G,4], padding rows = label -1 / all-zero boxes, left
    untouched): every image gets its own draws, the pixels of the whole batch move in four launches.  What the reference
    does per image inside ``train_data.map`` (trainer.py:42) before ``padded_batch``."""
    x = _h.to_dev(images)
    B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
    gb = _boxes(gt_boxes).copy()
    gl = None if gt_labels is None else np.asarray(gt_labels.detach().cpu().numpy() if isinstance(gt_labels, torch.Tensor) else gt_labels)
    plans = []
    for b in range(B):
        valid = (gl[b] > 0) if gl is not None else (np.abs(gb[b]).sum(-1) > 0)
        plan, g = draw_plan(H, W, gb[b][valid])
        gb[b][valid] = g
        plans.append(plan)
    return run_plans(x, plans), gb


def augmented(dataset, fn=None):
    """A dataset of padded batches (img, gt_boxes, gt_labels) with fresh augmentation draws on every pass: the analogue of
    ``train_data.map(lambda x: preprocessing(x, S, S, augmentation.apply))`` for already batched data."""
    fn = fn or apply_batch

    class _Aug(object):
        def __iter__(self_inner):
            for img, gt_boxes, gt_labels in dataset:
                im, gb = fn(img, gt_boxes, gt_labels)
                yield im, gb, gt_labels
    return _Aug()
