"""The augmentation half of the input pipeline (reference augmentation.py:4-183, used at trainer.py:42), the deterministic
pieces: given the random draws, the HIP kernels behind ``tf-ssd_amd/augmentation.py`` (``ssd_image_mean``,
``ssd_augment_geometry``, ``ssd_augment_color``) against the NumPy restatement oracle/augment_oracle.py -- geometry BIT-EXACT
(same per-op fp32 rounding: gather + bilinear interpolation + box arithmetic), colour within 1 ulp-scale tolerance (2e-6:
the per-channel mean is a float64 reduction in another order), boxes bit-exact.  The sampler itself is random: its
acceptance rule and its invariants are tested.  CPU tests pin the oracle to hand-computed known answers."""
import numpy as np
import pytest

from oracle import augment_oracle as ao

F32 = np.float32


def _image(h, w, seed=0):
    return np.random.default_rng(seed).random((h, w, 3), dtype=np.float32)


def _gt(seed=1, n=5):
    rng = np.random.default_rng(seed)
    c = rng.uniform(0.2, 0.8, (n, 2))
    s = rng.uniform(0.05, 0.3, (n, 2))
    return np.clip(np.concatenate([c - s, c + s], 1), 0, 1).astype(np.float32)


# ---------------------------------------------------------------------------------------------- CPU: the oracle itself
def test_oracle_known_answers():
    g = np.array([[0.1, 0.2, 0.5, 0.6]], F32)
    img = np.arange(2 * 3 * 3, dtype=F32).reshape(2, 3, 3)
    fi, fg = ao.flip_horizontally(img, g)
    np.testing.assert_array_equal(fi[:, 0], img[:, 2])
    np.testing.assert_allclose(fg, [[0.1, 0.4, 0.5, 0.8]], rtol=0, atol=1e-7)          # (y1, 1 - x2, y2, 1 - x1)
    # expand: 10 x 20 image, ratio 2 -> 20 x 40 canvas; u = 0.5 -> pad_left = rint(0.5 * 20) = 10, pad_top = rint(0.5 * 10) = 5
    assert ao.expand_geometry(10, 20, 2.0, 0.5, 0.5) == (20, 40, 5, 10)
    assert ao.expand_geometry(10, 20, 1.25, 0.5, 0.5) == (12, 25, 1, 2)               # rint(12.5) = 12, rint(2.5) = 2: half to even
    x = _image(10, 20)
    canvas, eg, mean = ao.expand_image(x, g, 2.0, 0.5, 0.5)
    assert canvas.shape == (20, 40, 3)
    np.testing.assert_array_equal(canvas[5:15, 10:30], x)
    np.testing.assert_array_equal(canvas[0, 0], mean)
    # a box at (0.1, 0.2, 0.5, 0.6) of the image sits at ((5 + 1) / 20, (10 + 4) / 40, (5 + 5) / 20, (10 + 12) / 40) of the canvas
    np.testing.assert_allclose(eg, [[0.3, 0.35, 0.5, 0.55]], atol=1e-6)
    # crop the image's own rectangle out of the canvas and resize to its size: the image comes back, boxes too
    back, bg, window = ao.crop_and_resize(canvas, eg, (5, 10), (10, 20), 10, 20)
    np.testing.assert_array_equal(back, x)
    np.testing.assert_allclose(bg, g, atol=1e-6)
    np.testing.assert_allclose(window, [0.25, 0.25, 0.75, 0.75])
    # boxes outside the window collapse onto its border (zero area) and stay in the list (augmentation.py:156-158)
    out = ao.renormalize_bboxes_with_min_max(np.array([[0.0, 0.0, 0.1, 0.1]], F32), np.array([0.5, 0.5, 1.0, 1.0], F32))
    np.testing.assert_array_equal(out, [[0, 0, 0, 0]])
    # [3P published] TF2 half-pixel bilinear resize of [[1,2],[3,4]] to 4x4
    r = ao.resize_bilinear(np.array([[[1.0], [2.0]], [[3.0], [4.0]]], F32), 4, 4)[..., 0]
    np.testing.assert_allclose(r[0], [1.0, 1.25, 1.75, 2.0])
    np.testing.assert_allclose(r[:, 0], [1.0, 1.5, 2.5, 3.0])


def test_oracle_colour_known_answers():
    px = np.array([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [0.5, 0.5, 0.5], [0.2, 0.4, 0.6]]], F32)
    hsv = ao.rgb_to_hsv(px)
    np.testing.assert_allclose(hsv[0, :3, 0], [0.0, 1 / 3, 2 / 3], atol=1e-7)          # pure red / green / blue hues
    np.testing.assert_allclose(hsv[0, 3], [0.0, 0.0, 0.5])                             # grey: no hue, no saturation
    np.testing.assert_allclose(ao.hsv_to_rgb(hsv), px, atol=1e-6)                      # round trip
    np.testing.assert_allclose(ao.adjust_hue(px, 1.0 / 3.0)[0, 0], [0.0, 1.0, 0.0], atol=1e-6)      # red -> green
    np.testing.assert_allclose(ao.adjust_saturation(px, 0.0)[0, 4], [0.6, 0.6, 0.6], atol=1e-6)     # fully desaturated: v
    np.testing.assert_allclose(ao.adjust_contrast(px, 1.0), px, atol=1e-7)
    np.testing.assert_allclose(ao.adjust_brightness(px, 0.1), px + F32(0.1))
    c = ao.color(px, brightness=0.9)
    assert c.max() <= 1.0 and c.min() >= 0.0
    assert ao.satisfies_overlap((0, 0, 50, 50), np.array([[0.0, 0.0, 0.4, 0.4]]), 0.9, 100, 100)
    assert not ao.satisfies_overlap((0, 0, 20, 20), np.array([[0.0, 0.0, 0.4, 0.4]]), 0.3, 100, 100)      # 25 % covered
    assert ao.satisfies_overlap((0, 0, 20, 20), np.array([[0.0, 0.0, 0.4, 0.4]]), 0.25, 100, 100)          # exactly 25 %


# ---------------------------------------------------------------------------------------------------------- GPU: kernels
@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (H, W, expand (ratio, u_left, u_top) or None, window (y, x, h, w) or None, flip)
    (40, 56, None, None, True),
    (40, 56, None, (3, 7, 20, 33), False),
    (40, 56, (2.37, 0.31, 0.77), (11, 20, 55, 70), False),
    (40, 56, (3.9, 0.0, 0.999), (0, 0, 156, 218), True),
    (300, 300, (1.0, 0.5, 0.5), (10, 20, 200, 150), True),
    (37, 29, (1.5, 0.2, 0.4), (5, 3, 41, 37), True),
])
def test_geometry_kernel_bit_exact(case):
    import augmentation as aug
    H, W, expand, window, flip = case
    x = _image(H, W, seed=H + W)
    g = _gt()
    ref_img, ref_g = ao.geometry(x, g, expand=expand, crop=window, flip=flip)
    out, og = x, g
    if window is not None:
        out, og = aug.patch(out, og, draws=(expand, window))
    if flip:
        out, og = aug.flip_horizontally(out, og)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref_img.view(np.uint32))
    np.testing.assert_array_equal(np.asarray(og).view(np.uint32), ref_g.view(np.uint32))
    # the fused batch path (one launch for expand + crop + resize + flip) gives the same pixels
    plan = {"canvas": (H, W, 0, 0), "crop": window, "flip": flip, "expand": expand is not None,
            "brightness": None, "contrast": None, "hue": None, "saturation": None}
    if expand is not None:
        plan["canvas"] = aug.expand_geometry(H, W, *expand)
    fused = aug.run_plans(x[None], [plan])[0].cpu().numpy()
    np.testing.assert_array_equal(fused.view(np.uint32), np.clip(ref_img, 0, 1).view(np.uint32))


@pytest.mark.gpu
def test_expand_image_materialised_and_mean():
    import augmentation as aug
    x = _image(24, 31, seed=3)
    g = _gt()
    canvas, eg = aug.expand_image(x, g, draws=(2.6, 0.4, 0.9))
    ref, rg, mean = ao.expand_image(x, g, 2.6, 0.4, 0.9)
    c = canvas.cpu().numpy()
    assert c.shape == ref.shape
    m = aug.image_mean(x).cpu().numpy()[0]
    assert np.abs(m - mean).max() <= 1.2e-7            # float64 sums in another order, rounded once to fp32
    inside = (c != c[0, 0]).any(-1) | (ref != ref[0, 0]).any(-1)
    np.testing.assert_array_equal(c[inside], ref[inside])
    assert np.abs(c - ref).max() <= 1.2e-7
    np.testing.assert_array_equal(eg, rg)


@pytest.mark.gpu
@pytest.mark.parametrize("draws", [
    dict(brightness=0.07), dict(contrast=1.37), dict(hue=-0.06), dict(saturation=0.55), dict(hue=0.08, saturation=1.5),
    dict(brightness=-0.12, contrast=0.5, hue=0.05, saturation=1.2), dict(),
])
def test_colour_kernel_vs_oracle(draws):
    import augmentation as aug
    x = _image(33, 47, seed=5)
    x[0, 0] = [0.0, 0.0, 0.0]
    x[0, 1] = [1.0, 1.0, 1.0]
    x[0, 2] = [0.3, 0.3, 0.3]                 # grey: hue undefined, saturation 0
    x[0, 3] = [1.0, 0.0, 0.5]
    ref = ao.color(x, **draws)
    out = aug._color(x, **draws).cpu().numpy()
    assert np.abs(out - ref).max() <= 2e-6, np.abs(out - ref).max()
    if "contrast" not in draws:               # no reduction involved: the same fp32 ops in the same order
        np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert out.min() >= 0.0 and out.max() <= 1.0


@pytest.mark.gpu
def test_apply_batch_invariants_and_padding():
    """``apply_batch`` = the reference's per-image ``apply`` on a padded batch: image shape and range kept, boxes stay
    normalised and ordered, padding rows untouched, different images get different draws, the same seed the same result."""
    import augmentation as aug
    import ssd_hip as h
    from utils import data_utils
    B, S = 6, 64
    imgs = data_utils.synthetic_images(B, S, seed=2)
    gt, gl = data_utils.synthetic_gt(B, total_labels=21, seed=3)
    aug.seed(11)
    a_img, a_gt = aug.apply_batch(imgs, gt, gl)
    a = a_img.cpu().numpy()
    assert a.shape == (B, S, S, 3) and a.min() >= 0.0 and a.max() <= 1.0 and np.isfinite(a).all()
    assert a_gt.shape == gt.shape and a_gt.min() >= 0.0 and a_gt.max() <= 1.0
    pad = np.asarray(gl) <= 0
    np.testing.assert_array_equal(a_gt[pad], np.asarray(gt)[pad])
    assert (a_gt[..., 2] >= a_gt[..., 0]).all() and (a_gt[..., 3] >= a_gt[..., 1]).all()
    assert sum(float(np.abs(a[b] - imgs[b]).max()) > 1e-3 for b in range(B)) >= 3          # most images changed
    aug.seed(11)
    b_img, b_gt = aug.apply_batch(imgs, gt, gl)
    np.testing.assert_array_equal(b_img.cpu().numpy(), a)
    np.testing.assert_array_equal(b_gt, a_gt)
    # the sampler's contract: the window it returns satisfies its own acceptance rule (or is the whole image)
    aug.seed(5)
    g = _gt(seed=9, n=4)
    for mo in (0.1, 0.5, 0.9):
        for _ in range(20):
            y, x, hh, ww = aug.sample_distorted_bounding_box(120, 160, g, mo)
            assert 0 <= y and 0 <= x and y + hh <= 120 and x + ww <= 160 and hh > 0 and ww > 0
            whole = (y, x, hh, ww) == (0, 0, 120, 160)
            assert whole or aug.window_satisfies((y, x, y + hh, x + ww), aug.pixel_rectangles(g, 120, 160), mo)
            assert whole or (0.5 - 0.05 <= ww / hh <= 2.0 + 0.05 and hh * ww >= 0.05 * 120 * 160 - 1)


@pytest.mark.gpu
def test_trainer_runs_with_augmentation(tmp_path, monkeypatch, capsys):
    """reference trainer.py:42: the training stream passes through ``augmentation.apply`` (here: ``apply_batch``)."""
    import importlib
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SSD_TRAINER_EPOCHS", "1")
    monkeypatch.setenv("SSD_TRAINER_STEPS", "2")
    monkeypatch.setenv("SSD_TRAINER_BATCH", "4")
    monkeypatch.setenv("SSD_TRAINER_AUGMENT", "1")
    trainer = importlib.import_module("trainer")
    hist = trainer.main(["--backbone", "mobilenet_v2"])
    assert len(hist["loss"]) == 1 and np.isfinite(hist["loss"]).all() and np.isfinite(hist["val_loss"]).all()
