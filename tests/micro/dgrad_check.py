"""Debug: backward-data of a conv computed the way csrc/ssd_train.hip does it (forward MFMA conv
kernel on dY with rotated/transposed weights, zero-inserted dY for stride 2, in-place accumulate
through the residual input) vs torch-CPU autograd.  Run on a GPU box."""
import ctypes, sys, os
import numpy as np, torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "tf-ssd_amd"), os.path.join(REPO, "tests")]
import ssd_hip as h
from oracle import net_oracle as no

lib = h.lib()


def ref_dx(x, w, dy, stride, pads):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    wt = torch.from_numpy(w).permute(3, 2, 0, 1).contiguous()
    pt, pb, pl, pr = pads
    y = F.conv2d(F.pad(xt, (pl, pr, pt, pb)), wt, stride=stride)
    (y * torch.from_numpy(dy).permute(0, 3, 1, 2)).sum().backward()
    return xt.grad.permute(0, 2, 3, 1).contiguous().numpy()


def gpu_dx(x_shape, w, dy, stride, pads, cpad=None, accumulate_into=None):
    B, H, W, Cin = x_shape
    kh, kw, _, Cout = w.shape
    cpad = cpad or Cout
    Ho, Wo = dy.shape[1], dy.shape[2]
    wt = np.zeros((kh, kw, cpad, Cin), np.float32)
    wt[:, :, :Cout, :] = np.transpose(w[::-1, ::-1], (0, 1, 3, 2))
    dyp = np.zeros((B, Ho, Wo, cpad), np.float32)
    dyp[..., :Cout] = dy
    if stride == 2:
        Hz, Wz = (Ho - 1) * 2 + 1, (Wo - 1) * 2 + 1
        z = np.zeros((B, Hz, Wz, cpad), np.float32)
        z[:, ::2, ::2] = dyp
        dyp, Ho, Wo = z, Hz, Wz
    pt, pl = (kh - 1) - pads[0], (kw - 1) - pads[2]
    pb, pr = H - Ho + (kh - 1) - pt, W - Wo + (kw - 1) - pl
    d = h.ConvDesc(B, Ho, Wo, cpad, Cin, kh, kw, 1, 1, pt, pl, pb, pr, 0, int(accumulate_into is not None))
    xd, wd = h.to_dev(dyp), h.to_dev(wt)
    packed = torch.empty((lib.ssd_conv_packed_weight_floats(kh, kw, cpad, Cin),), dtype=torch.float32, device=xd.device)
    h.check(lib.ssd_conv_pack_weights(h.ptr(wd), kh, kw, cpad, Cin, h.ptr(packed), h.stream()), "pack")
    out = h.to_dev(accumulate_into) if accumulate_into is not None else torch.full((B, H, W, Cin), float("nan"), dtype=torch.float32, device=xd.device)
    res = out if accumulate_into is not None else None
    h.check(lib.ssd_conv2d(ctypes.byref(d), h.ptr(xd), h.ptr(packed), None, None, h.ptr(res), h.ptr(out), 0, 0, h.stream()), "conv")
    return out.cpu().numpy()


rng = np.random.default_rng(0)
for name, (B, H, Cin, Cout, k, s, cpad, acc) in {
        "head3 5x5": (4, 5, 512, 150, 3, 1, 160, False), "head2 10x10": (4, 10, 1280, 150, 3, 1, 160, False),
        "head1 19x19": (4, 19, 576, 100, 3, 1, 128, False), "extra1_2 s2 10": (4, 10, 256, 512, 3, 2, None, False),
        "extra2_2 s2 5": (4, 5, 128, 256, 3, 2, None, False), "extra1_1 1x1 acc": (4, 10, 1280, 256, 1, 1, None, True),
        "project 1x1": (4, 38, 192, 32, 1, 1, None, False)}.items():
    x = rng.standard_normal((B, H, H, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    if k == 1:
        pads = (0, 0, 0, 0)
    else:
        _, a, b = no.same_pads(H, k, s, 1)
        pads = (a, b, a, b)
    Ho = lib.ssd_conv_out_size(H, k, s, 1, pads[0], pads[1])
    dy = rng.standard_normal((B, Ho, Ho, Cout)).astype(np.float32)
    ref = ref_dx(x, w, dy, s, pads)
    base = rng.standard_normal(ref.shape).astype(np.float32) if acc else None
    got = gpu_dx(x.shape, w, dy, s, pads, cpad, base)
    if acc:
        ref = ref + base
    err = np.abs(got - ref)
    print("%-18s max err %.3e (max ref %.3e) worst at %s  nan %d" % (name, err.max(), np.abs(ref).max(),
          np.unravel_index(np.nanargmax(err), err.shape), int(np.isnan(got).sum())))
