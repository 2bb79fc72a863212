"""NumPy restatement of the reference's ``augmentation.py`` (SSD photometric / geometric augmentation used at
trainer.py:42: ``data_utils.preprocessing(x, img_size, img_size, augmentation.apply)``).

TEST INFRASTRUCTURE ONLY (tests/, nothing under tf-ssd_amd/ imports this).  **Parity unpinned**: TensorFlow cannot be
imported here and the reference ships no fixtures, so the TF image ops are restated from their published semantics [3P]
(TF 2.0, environment.yml:70-73) and every function cites the reference line it follows.  The RANDOM draws are arguments
here (the reference draws them with ``tf.random.uniform`` / ``sample_distorted_bounding_box``): given the same draws,
every function is deterministic -- that is what the HIP kernels are held to.

The colour ops in particular: TF 2.0 dispatches ``tf.image.adjust_hue`` / ``adjust_saturation`` / ``adjust_contrast`` to its
FUSED C++ kernels (``AdjustHue``, ``AdjustSaturation``, ``AdjustContrastv2``), each with its own operation order (the hue /
saturation kernels convert RGB -> HSV -> RGB per pixel in one pass; ``AdjustContrastv2`` computes the per-channel mean in
float and applies ``(x - mean) * factor + mean``).  The plain HSV path below restates their published semantics, not their
instruction order: agreement with a real TF trace is expected to the last few ulps, not bitwise, and stays unpinned until
``tools/make_tf_golden.py`` can run.

Images: float32 [H,W,3] in [0,1] (the reference augments AFTER convert + resize, utils/data_utils.py:22-26).
Boxes: float32 [G,4] = (y1, x1, y2, x2) normalised."""
import numpy as np

F32 = np.float32


def renormalize_bboxes_with_min_max(bboxes, min_max):
    """utils/bbox_utils.py:178-188: r = (x - min) / (max - min), clipped to [0,1]."""
    b = np.asarray(bboxes, F32)
    y_min, x_min, y_max, x_max = [F32(v) for v in np.asarray(min_max, F32)]
    r = b - np.array([y_min, x_min, y_min, x_min], F32)
    r = r / np.array([y_max - y_min, x_max - x_min, y_max - y_min, x_max - x_min], F32)
    return np.clip(r, F32(0), F32(1)).astype(F32)


def flip_horizontally(img, gt_boxes):
    """augmentation.py:95-110: ``tf.image.flip_left_right`` + boxes (y1, 1 - x2, y2, 1 - x1)."""
    g = np.asarray(gt_boxes, F32)
    flipped = np.stack([g[..., 0], F32(1.0) - g[..., 3], g[..., 2], F32(1.0) - g[..., 1]], -1).astype(F32)
    return np.ascontiguousarray(np.asarray(img, F32)[:, ::-1, :]), flipped


def expand_geometry(height, width, expansion_ratio, u_left, u_top):
    """augmentation.py:135-140 with its three uniform draws as arguments: ``expansion_ratio`` in [1, 4), ``u_left`` /
    ``u_top`` in [0, 1) standing for ``uniform(0, final - size)`` = u * (final - size).  tf.round is round-half-to-even
    (= np.rint).  Returns integers (final_h, final_w, pad_top, pad_left)."""
    height, width = F32(height), F32(width)
    final_h = np.rint(height * F32(expansion_ratio)).astype(F32)
    final_w = np.rint(width * F32(expansion_ratio)).astype(F32)
    pad_left = np.rint(F32(u_left) * (final_w - width)).astype(F32)
    pad_top = np.rint(F32(u_top) * (final_h - height)).astype(F32)
    return int(final_h), int(final_w), int(pad_top), int(pad_left)


def expand_image(img, gt_boxes, expansion_ratio, u_left, u_top):
    """augmentation.py:123-151: the image on a larger canvas filled with its per-channel mean (``tf.nn.moments(img,
    [0, 1])``; float64 mean here -- TF's fp32 reduction order is not reproducible), boxes renormalised to the canvas
    (min_max = [-pad_top, -pad_left, pad_bottom + h, pad_right + w] / [h, w, h, w])."""
    x = np.asarray(img, F32)
    h, w, _ = x.shape
    fh, fw, pt, pl = expand_geometry(h, w, expansion_ratio, u_left, u_top)
    mean = x.astype(np.float64).mean((0, 1)).astype(F32)
    canvas = np.empty((fh, fw, x.shape[2]), F32)
    canvas[...] = mean
    canvas[pt:pt + h, pl:pl + w] = x
    pad_bottom, pad_right = F32(fh) - (F32(h) + F32(pt)), F32(fw) - (F32(w) + F32(pl))
    min_max = np.array([-F32(pt), -F32(pl), pad_bottom + F32(h), pad_right + F32(w)], F32) / np.array([h, w, h, w], F32)
    return canvas, renormalize_bboxes_with_min_max(gt_boxes, min_max), mean


def resize_bilinear(img, out_h, out_w):
    """``tf.image.resize`` (bilinear, TF2 half-pixel centres, no antialias) of a float32 image -- the arithmetic of
    oracle/bbox_oracle.preprocess_image without the uint8 conversion."""
    f = np.asarray(img, F32)
    H, W, _ = f.shape

    def weights(out_size, in_size):
        scale = F32(in_size) / F32(out_size)
        src = (np.arange(out_size, dtype=F32) + F32(0.5)) * scale - F32(0.5)
        fl = np.floor(src)
        lo = np.maximum(fl.astype(np.int64), 0)
        hi = np.minimum(np.ceil(src).astype(np.int64), in_size - 1)
        return lo, hi, (src - fl).astype(F32)
    y0, y1, ly = weights(out_h, H)
    x0, x1, lx = weights(out_w, W)
    lx = lx[None, :, None]
    ly = ly[:, None, None]
    tl, tr = f[y0][:, x0], f[y0][:, x1]
    bl, br = f[y1][:, x0], f[y1][:, x1]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    return (top + (bot - top) * ly).astype(F32)


def crop_and_resize(img, gt_boxes, begin, size, out_h, out_w):
    """augmentation.py:176-178: ``tf.slice(img, begin, size)`` -> ``tf.image.resize(img, (org_height, org_width))`` ->
    boxes renormalised to the crop window.  ``begin`` / ``size`` = (y, x) / (h, w) of ``sample_distorted_bounding_box``;
    its third output, the window in normalised coordinates [y_min, x_min, y_max, x_max], is begin / size over the image
    being cropped."""
    x = np.asarray(img, F32)
    H, W, _ = x.shape
    y, xx = int(begin[0]), int(begin[1])
    h, w = int(size[0]), int(size[1])
    window = np.array([F32(y) / F32(H), F32(xx) / F32(W), F32(y + h) / F32(H), F32(xx + w) / F32(W)], F32)
    out = resize_bilinear(x[y:y + h, xx:xx + w], out_h, out_w)
    return out, renormalize_bboxes_with_min_max(gt_boxes, window), window


def geometry(img, gt_boxes, expand=None, crop=None, flip=False):
    """The geometric half of ``apply`` (augmentation.py:19-23: patch, then flip) with every decision and draw given:
    ``expand`` = None or (expansion_ratio, u_left, u_top); ``crop`` = None or (begin_y, begin_x, size_h, size_w) on the
    (possibly expanded) image -- None with ``expand`` set is not a reference path (patch always crops); ``flip`` bool.
    Output size = input size."""
    x = np.asarray(img, F32)
    g = np.asarray(gt_boxes, F32)
    H, W, _ = x.shape
    if crop is not None:
        if expand is not None:
            x, g, _ = expand_image(x, g, *expand)
        x, g, _ = crop_and_resize(x, g, crop[:2], crop[2:], H, W)
    if flip:
        x, g = flip_horizontally(x, g)
    return x, g


# ---- photometric half (augmentation.py:51-93) -- [3P] TF 2.0 image ops restated -----------------------------------------
def adjust_brightness(img, delta):
    """tf.image.random_brightness -> adjust_brightness: image + delta (float images: no scaling)."""
    return (np.asarray(img, F32) + F32(delta)).astype(F32)


def adjust_contrast(img, factor, mean=None):
    """tf.image.random_contrast -> adjust_contrast: (x - mean_c) * factor + mean_c, mean per channel over H, W."""
    x = np.asarray(img, F32)
    m = x.astype(np.float64).mean((0, 1)).astype(F32) if mean is None else np.asarray(mean, F32)
    return ((x - m) * F32(factor) + m).astype(F32)


def rgb_to_hsv(x):
    """tf.image.rgb_to_hsv (float32): v = max, s = (max - min) / max (0 where max == 0), h in [0, 1)."""
    x = np.asarray(x, F32)
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    rng = (v - mn).astype(F32)
    s = np.where(v > 0, rng / np.where(v > 0, v, F32(1)), F32(0)).astype(F32)
    norm = np.where(rng > 0, F32(1.0) / (F32(6.0) * np.where(rng > 0, rng, F32(1))), F32(0)).astype(F32)
    h = np.where(r == v, norm * (g - b), np.where(g == v, norm * (b - r) + F32(2.0 / 6.0), norm * (r - g) + F32(4.0 / 6.0)))
    h = np.where(rng > 0, h, F32(0)).astype(F32)
    h = np.where(h < 0, h + F32(1.0), h).astype(F32)
    return np.stack([h, s, v], -1).astype(F32)


def hsv_to_rgb(x):
    """tf.image.hsv_to_rgb (float32): c = s v, per channel clamp(|6 h - k| ...) form."""
    x = np.asarray(x, F32)
    h, s, v = x[..., 0], x[..., 1], x[..., 2]
    c = (s * v).astype(F32)
    m = (v - c).astype(F32)
    dh = (h * F32(6.0)).astype(F32)
    fmodu = dh.copy()
    while True:
        over = fmodu >= F32(2.0)
        if not over.any():
            break
        fmodu = np.where(over, fmodu - F32(2.0), fmodu).astype(F32)
    xx = (c * (F32(1.0) - np.abs(fmodu - F32(1.0)))).astype(F32)
    hc = np.floor(dh).astype(np.int64)
    z = np.zeros_like(c)
    rr = np.select([hc == 0, hc == 1, hc == 2, hc == 3, hc == 4], [c, xx, z, z, xx], c)
    gg = np.select([hc == 0, hc == 1, hc == 2, hc == 3, hc == 4], [xx, c, c, xx, z], z)
    bb = np.select([hc == 0, hc == 1, hc == 2, hc == 3, hc == 4], [z, z, xx, c, c], xx)
    return np.stack([rr + m, gg + m, bb + m], -1).astype(F32)


def adjust_hue(img, delta):
    """tf.image.random_hue -> adjust_hue: hue rotated by delta (mod 1) in HSV space."""
    hsv = rgb_to_hsv(img)
    h = hsv[..., 0] + F32(delta)
    h = np.where(h < 0, h + F32(1.0), h)
    h = np.where(h >= F32(1.0), h - F32(1.0), h).astype(F32)
    return hsv_to_rgb(np.stack([h, hsv[..., 1], hsv[..., 2]], -1))


def adjust_saturation(img, factor):
    """tf.image.random_saturation -> adjust_saturation: s * factor clipped to [0,1] in HSV space."""
    hsv = rgb_to_hsv(img)
    s = np.clip(hsv[..., 1] * F32(factor), F32(0), F32(1)).astype(F32)
    return hsv_to_rgb(np.stack([hsv[..., 0], s, hsv[..., 2]], -1))


def color(img, brightness=None, contrast=None, hue=None, saturation=None):
    """The photometric half of ``apply`` in the reference's order (augmentation.py:17, 22-25): brightness, contrast, hue,
    saturation, each applied when its draw is not None, then ``clip_by_value(img, 0, 1)``."""
    x = np.asarray(img, F32)
    if brightness is not None:
        x = adjust_brightness(x, brightness)
    if contrast is not None:
        x = adjust_contrast(x, contrast)
    if hue is not None:
        x = adjust_hue(x, hue)
    if saturation is not None:
        x = adjust_saturation(x, saturation)
    return np.clip(x, F32(0), F32(1)).astype(F32)


def satisfies_overlap(window, gt_boxes, min_object_covered, height, width):
    """[3P] SampleDistortedBoundingBox's acceptance test (``SatisfiesOverlapConstraints`` of the TF 2.0 kernel): window
    (y1, x1, y2, x2) in PIXELS, ground-truth boxes normalised; the kernel truncates the boxes to integer pixel rectangles
    first; boxes (or a window) without a single pixel never satisfy; some box with >= ``min_object_covered`` of its pixel
    area inside the window does."""
    g = np.asarray(gt_boxes, np.float64).reshape(-1, 4)
    wy1, wx1, wy2, wx2 = [int(v) for v in window]
    if (wy2 - wy1) * (wx2 - wx1) < 1:
        return False
    for y1, x1, y2, x2 in g:
        by1, bx1, by2, bx2 = int(y1 * height), int(x1 * width), int(y2 * height), int(x2 * width)
        area = (by2 - by1) * (bx2 - bx1)
        if area < 1:
            continue
        ih = max(0, min(by2, wy2) - max(by1, wy1))
        iw = max(0, min(bx2, wx2) - max(bx1, wx1))
        if np.float32(ih * iw) / np.float32(area) >= np.float32(min_object_covered):
            return True
    return False
