"""LDS-DMA conv tiles over pre-split bf16 activation planes (csrc/ssd_convdma.hip, round 5): ``ssd_split_planes`` /
``ssd_join_planes`` / ``ssd_conv2d_planes`` through the C ABI.

* planes: the three-way split is EXACT (join(split(x)) == x bit for bit, non-finite and denormal inputs included as far as
  the split defines them), the one-plane form is the nearest-even bf16 rounding;
* every ``dma3_*`` tile against the NumPy conv oracle within the fp32 contract (1e-4; Keras Conv2D + BatchNorm + ReLU6 +
  residual call sites: models/ssd_mobilenet_v2.py:16-32, models/ssd_vgg16.py:52-91, models/header.py:60-61), on SAME /
  VALID / stride-2 / dilation-6 / 1x1 shapes with ragged M and Cout tails, with and without split-K -- and BITWISE equal to
  the ``mfma3_*`` tile of the same shape (same exact planes, same K walk, same six-product order: only the road into LDS
  differs);
* every ``dmab_*`` tile (bf16 storage) against a float64 contraction of the ROUNDED operands (2e-5 of the output scale);
* the plane output of the epilogue / of the split-K reduce is the split of the fp32 output it sits beside;
* planes are read inside their range only: NaN-poisoned gaps around every plane."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import net_oracle as no
from test_conv_gpu import run_conv, same, _np, _close, guarded
from test_bf16_gpu import bf16_round, conv_f64

pytestmark = pytest.mark.gpu

POISON = 4096          # bf16 elements of NaN around every plane


def make_planes(x, np_):
    """fp32 NHWC array (last axis = channels, a multiple of 32) -> (device int16 buffer, byte pointer of plane 0, plane
    stride in elements): slice-major planes written by ``ssd_split_planes`` into a NaN-poisoned buffer (bf16 NaN = 0x7fc0)."""
    import ssd_hip as h
    lib = h.lib()
    xd = guarded(x)
    n = int(np.prod(x.shape))
    stride = (n + POISON + 63) // 64 * 64
    buf = torch.full((POISON + np_ * stride + POISON,), 0x7fc0, dtype=torch.int16, device=xd.device)
    p0 = buf.data_ptr() + 2 * POISON
    assert p0 % 16 == 0
    rc = lib.ssd_split_planes(h.ptr(xd), n, int(x.shape[-1]), np_, h.vp(p0), stride, h.stream())
    if x.shape[-1] % 32:            # no slice-major planes of such a tensor: refused, the buffer stays poison
        assert rc == -1 and b"channels" in lib.ssd_last_error()
    else:
        h.check(rc, "split_planes")
    return buf, p0, stride


def join_planes(p0, n, channels, np_, stride):
    import ssd_hip as h
    out = torch.empty((n,), dtype=torch.float32, device=h.device())
    h.check(h.lib().ssd_join_planes(h.vp(p0), n, channels, np_, stride, h.ptr(out), h.stream()), "join_planes")
    return _np(out)


def run_conv_planes(x, w, np_, cfg, scale=None, shift=None, res=None, stride=1, dil=1, pads=(0, 0, 0, 0), act=0, split_k=1,
                    want_planes=True):
    import ssd_hip as h
    lib = h.lib()
    B, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    d = h.ConvDesc(B, H, W, Cin, Cout, kh, kw, stride, dil, pads[0], pads[2], pads[1], pads[3], act, int(res is not None))
    wd = h.to_dev(w)
    npk = lib.ssd_conv_packed_weight_floats(kh, kw, Cin, Cout)
    poisoned = torch.full((npk + 4096,), float("nan"), dtype=torch.float32, device=wd.device)
    packed = poisoned[:npk]
    h.check(lib.ssd_conv_pack_weights(h.ptr(wd), kh, kw, Cin, Cout, h.ptr(packed), h.stream()), "pack")
    Ho = lib.ssd_conv_out_size(H, kh, stride, dil, pads[0], pads[1])
    Wo = lib.ssd_conv_out_size(W, kw, stride, dil, pads[2], pads[3])
    keep, p0, pstride = make_planes(x, np_)
    sd = guarded(scale) if scale is not None else None
    hd = guarded(shift) if shift is not None else None
    rd = guarded(res) if res is not None else None
    out = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.float32, device=wd.device)
    n_out = B * Ho * Wo * Cout
    op = None
    ostride = 0
    if want_planes and Cout % 32 == 0:
        ostride = (n_out + 63) // 64 * 64
        op = torch.full((np_ * ostride + 64,), 0x7fc0, dtype=torch.int16, device=wd.device)
    ws = torch.empty(max(1, split_k * n_out), dtype=torch.float32, device=wd.device) if split_k > 1 else None
    rc = lib.ssd_conv2d_planes(ctypes.byref(d), h.vp(p0), np_, pstride, h.ptr(packed), h.ptr(sd), h.ptr(hd), h.ptr(rd),
                               h.ptr(out), 0, 0, h.ptr(op), ostride, cfg, split_k, h.ptr(ws), h.stream())
    joined = None
    if rc == 0 and op is not None:
        joined = join_planes(op.data_ptr(), n_out, Cout, np_, ostride).reshape(B, Ho, Wo, Cout)
    return rc, out, joined


def dma_configs(prefix):
    import ssd_hip as h
    lib = h.lib()
    return [(c, lib.ssd_conv_config_name(c).decode()) for c in range(lib.ssd_conv_num_configs())
            if lib.ssd_conv_config_name(c).startswith(prefix)]


def test_split_planes_is_exact_and_join_restores_it():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(8192) * np.exp(rng.uniform(-30, 30, 8192))).astype(np.float32)        # 128 pixels x 64 channels
    x[:8] = [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-30, 6.0]
    x[8:12] = np.float32([2.0 ** -120, -(2.0 ** -120), 1.17549435e-38, 65504.0])
    buf, p0, stride = make_planes(x.reshape(128, 64), 3)
    back = join_planes(p0, x.size, 64, 3, stride)
    big = np.abs(x) >= np.float32(2.0 ** -100)           # below ~2^-110 the l plane leaves bf16's normal range (documented)
    np.testing.assert_array_equal(back[big].view(np.uint32), x[big].view(np.uint32))
    assert np.abs(back[~big] - x[~big]).max() <= np.float32(2.0 ** -125)
    # the poison around the planes is untouched
    allb = _np(buf)
    assert (allb[:POISON] == 0x7fc0).all() and (allb[POISON + x.size:POISON + stride] == 0x7fc0).all()
    buf1, p1, stride1 = make_planes(x.reshape(128, 64), 1)
    np.testing.assert_array_equal(join_planes(p1, x.size, 64, 1, stride1), bf16_round(x))
    # slice-major: plane h of channel slice 1 starts P * 32 elements in -- pixel 5, channel 40 sits at (1 * 128 + 5) * 32 + 8
    h_plane = allb[POISON:POISON + x.size].view(np.uint16)
    assert h_plane[(128 + 5) * 32 + 8] == (x.reshape(128, 64)[5, 40].view(np.uint32) >> 16)


DMA_CASES = [
    # (B, H, Cin, Cout, k, stride, dil, padding)
    (2, 19, 96, 576, 1, 1, 1, "valid"),     # MBv2 expand: 1x1, K = 96 (three k-steps: ragged last tile in the bf16 form)
    (3, 10, 320, 1280, 1, 1, 1, "valid"),   # Conv_1
    (2, 19, 576, 100, 3, 1, 1, "same"),     # head level 1 (A (L + 4) = 100: ragged Cout)
    (2, 10, 256, 512, 3, 2, 1, "same"),     # extra1_2: stride-2 SAME (0,1)
    (2, 5, 128, 256, 3, 2, 1, "same"),      # extra2_2: (1,1)
    (1, 19, 64, 84, 3, 1, 1, "same"),       # head-like, Cout not a tile multiple
    (1, 19, 64, 64, 3, 1, 6, "same"),       # conv6-like dilation 6
    (2, 5, 128, 256, 3, 1, 1, "valid"),     # conv10_2-like VALID
    (70, 2, 256, 84, 3, 1, 1, "same"),      # tiny maps: a tile spans many images
    (1, 38, 64, 128, 3, 1, 1, "same"),      # VGG-like, M = 1444 (ragged M)
]


@pytest.mark.parametrize("case", DMA_CASES)
def test_conv2d_dma3_all_configs(case):
    import ssd_hip as h
    lib = h.lib()
    B, H, Cin, Cout, k, stride, dil, padding = case
    rng = np.random.default_rng(hash(case[:5]) % 1000)
    x = rng.standard_normal((B, H, H, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    pads = same(H, k, stride, dil) * 2 if padding == "same" else (0, 0, 0, 0)
    ref = no.relu6(no.conv2d(x, w, None, stride, dil, padding) * scale + shift)
    res = rng.standard_normal(ref.shape).astype(np.float32)
    mfma3 = {n[len("mfma3_"):]: c for c, n in dma_configs(b"mfma3_")}
    ran = 0
    for cfg, name in dma_configs(b"dma3_"):
        rc, out, joined = run_conv_planes(x, w, 3, cfg, scale, shift, res, stride, dil, pads, act=2)
        assert rc == 0, (name, lib.ssd_last_error())
        o = _np(out)
        _close(o, ref + res, 1e-4)
        if joined is not None:               # the plane output IS the fp32 output, split exactly
            np.testing.assert_array_equal(joined.view(np.uint32), o.view(np.uint32))
        twin = mfma3.get(name[len("dma3_"):])
        if twin is not None:                 # same planes, same K walk, same product order: same bits
            rc2, out2 = run_conv(x, w, scale, shift, res, stride, dil, pads, act=2, cfg=twin)
            if rc2 == 0:
                np.testing.assert_array_equal(_np(out2).view(np.uint32), o.view(np.uint32))
        ran += 1
    assert ran >= 8
    # split-K (deterministic slabs + reduce, which writes the planes too)
    cfg, name = dma_configs(b"dma3_")[1]
    for sk in (2, 3):
        rc, out, joined = run_conv_planes(x, w, 3, cfg, scale, shift, res, stride, dil, pads, act=2, split_k=sk)
        assert rc == 0, (name, sk, lib.ssd_last_error())
        _close(_np(out), ref + res, 1e-4)
        if joined is not None:
            np.testing.assert_array_equal(joined.view(np.uint32), _np(out).view(np.uint32))


@pytest.mark.parametrize("case", [c for c in DMA_CASES if c[2] % 64 == 0 or c[4] == 1])
def test_conv2d_dmab_all_configs(case):
    """bf16 storage: the tile multiplies the bf16-ROUNDED activation (the plane) with the bf16-rounded weights, fp32
    accumulation: against a float64 contraction of the rounded operands only the accumulation order differs."""
    import ssd_hip as h
    lib = h.lib()
    B, H, Cin, Cout, k, stride, dil, padding = case
    if dil != 1:
        pytest.skip("float64 helper has no dilation")
    rng = np.random.default_rng(hash(case[:5]) % 1000 + 1)
    x = rng.standard_normal((B, H, H, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    pads = same(H, k, stride) * 2 if padding == "same" else (0, 0, 0, 0)
    ref = conv_f64(bf16_round(x), bf16_round(w), stride, pads)
    scale_ = float(np.abs(ref).max())
    ran = 0
    for cfg, name in dma_configs(b"dmab_"):
        rc, out, joined = run_conv_planes(x, w, 1, cfg, None, None, None, stride, 1, pads)
        assert rc == 0, (name, lib.ssd_last_error())
        o = _np(out).astype(np.float64)
        assert np.abs(o - ref).max() <= 2e-5 * scale_, (name, np.abs(o - ref).max(), scale_)
        if joined is not None:
            np.testing.assert_array_equal(joined, bf16_round(_np(out)))
        ran += 1
    assert ran >= 8


def test_dma_tiles_refuse_what_they_cannot_run():
    import ssd_hip as h
    lib = h.lib()
    rng = np.random.default_rng(0)
    cfg, _ = dma_configs(b"dma3_")[0]
    x = rng.standard_normal((1, 8, 8, 24)).astype(np.float32)            # Cin % 32 != 0
    w = rng.standard_normal((3, 3, 24, 32)).astype(np.float32)
    rc, _, _ = run_conv_planes(x, w, 3, cfg, pads=(1, 1, 1, 1))
    assert rc == -3 and b"cannot run" in lib.ssd_last_error()
    x = rng.standard_normal((1, 8, 8, 32)).astype(np.float32)
    w = rng.standard_normal((3, 3, 32, 32)).astype(np.float32)
    rc, _, _ = run_conv_planes(x, w, 1, cfg, pads=(1, 1, 1, 1))          # one plane handed to a three-plane tile
    assert rc == -3
    rc, _ = run_conv(x, w, pads=(1, 1, 1, 1), cfg=cfg)                   # fp32 entry point: no planes
    assert rc == -3


@pytest.mark.parametrize("backbone,prec,B", [("mobilenet_v2", "fp32", 16), ("vgg16", "fp32", 2), ("mobilenet_v2", "bf16", 64), ("vgg16", "bf16", 8)])
def test_nets_on_lds_dma_tiles(backbone, prec, B):
    """Whole nets with the register-staged tiles of a kernel table replaced by their LDS-DMA twins (same tile shape,
    same split-K; ``set_tuning``): the producers' epilogues (conv / split-K reduce / max-pool / L2 normalisation / VGG stem /
    whole-image block kernel incl. block 13's expanded map and the combine launch) or a split pass write the bf16 planes.
    fp32 nets: network outputs BITWISE those of the ``mfma3_*`` table (exact planes, same K walk, same product order);
    bf16 nets: the planes ARE the mode's rounding of the activation, only the K-slice grouping differs (64 instead of 32
    channels per tap visit): outputs within the mode's own noise level.  Every live plane tensor is checked against the fp32
    activation it mirrors."""
    import ssd_hip as h
    import helpers
    lib = h.lib()
    get_model = __import__("models.ssd_%s" % backbone, fromlist=["get_model"]).get_model
    hp = helpers.hyper_params(backbone)
    w = helpers.synthetic_weights(backbone, hp)
    x = helpers.images(B, 300, seed=9)
    ref = get_model(hp, max_batch=B, precision=prec)
    ref.set_option("conv_dma", 0)
    ref.set_weights(w)
    d0, p0 = [_np(t) for t in ref(x)]
    names = {lib.ssd_conv_config_name(c).decode() for c in range(lib.ssd_conv_num_configs())}
    src, dst = ("mfma3_", "dma3_") if prec == "fp32" else ("bf16_", "dmab_")
    lines, swapped = [], 0
    for l in ref.get_tuning().splitlines():
        parts = l.split(" ")
        if len(parts) == 3 and parts[1].startswith(src) and dst + parts[1][len(src):] in names:
            parts[1] = dst + parts[1][len(src):]
            swapped += 1
        lines.append(" ".join(parts))
    assert swapped >= 2, "the reference table holds too few %s* lines to swap" % src
    m = get_model(hp, max_batch=B, precision=prec)
    m.set_weights(w)
    m.set_tuning("\n".join(lines) + "\n")
    d1, p1 = [_np(t) for t in m(x)]
    ran = [r for r in m.layers(B) if r["config"].startswith(dst) and r["flops"] > 0]
    # the planes are allocated for finalize's race and kept only where a CHOSEN tile reads them
    mem0, mem1 = ref.memory_summary(), m.memory_summary()
    assert mem0["planes"] == 0 and mem0["arena"] == mem1["arena"] > 0
    n_dma_inputs = len({r["name"] for r in m.layers(B) if r["config"].startswith(dst)})
    assert 0 < mem1["planes"] <= (3 if prec == "fp32" else 1) * mem1["arena"] // 2 + 512 * n_dma_inputs, (mem0, mem1)
    # (a swapped line whose layer has Cin % 32 != 0 -- or % 64 in the bf16 form -- falls back to the autotune: fine)
    # (MobileNetV2's dense convs outside the fused blocks are few: Conv_1, the extras, the heads)
    assert len(ran) >= (1 if backbone == "mobilenet_v2" else 3), "too few layers ran on %s* tiles: %s" % (dst, [r["config"] for r in m.layers(B)])
    if prec == "fp32":
        np.testing.assert_array_equal(d1.view(np.uint32), d0.view(np.uint32))
        np.testing.assert_array_equal(p1.view(np.uint32), p0.view(np.uint32))
    else:
        print("%s bf16, %d layers on dmab tiles: probs max |d| %.2e, deltas %.2e" % (backbone, len(ran), np.abs(p1 - p0).max(), np.abs(d1 - d0).max()))
        # (not accumulation-order noise alone: an fp32 output that moves in its last bit can land on the other side of a
        # bf16 rounding boundary of the NEXT layer's operand, 2^-9 relative, which these He-normal nets amplify -- measured
        # 1.6e-2 on VGG16, the size of the mode's own deviation from fp32; the tiles themselves are held to 2e-5 above)
        assert np.abs(p1 - p0).max() <= 6e-2 and np.abs(d1 - d0).max() <= 6e-2 * max(1.0, np.abs(d0).max())
    checked = 0
    for r in m.layers(B):
        try:
            pl, npl = m.fetch_planes(r["name"])
        except ValueError:
            continue
        if pl is None:
            continue
        act = m.fetch_activation(r["name"])
        assert npl == (3 if prec == "fp32" else 1)
        want = act if npl == 3 else bf16_round(act)
        np.testing.assert_array_equal(pl.view(np.uint32), want.view(np.uint32), err_msg="planes of %s" % r["name"])
        checked += 1
    assert checked >= 1, checked
    # the LDS-DMA option off: no planes, same results as the reference
    m.set_option("conv_dma", 0)
    m.set_tuning(None)
    d2, p2 = [_np(t) for t in m(x)]
    np.testing.assert_array_equal(p2, p0)
