"""TEST INFRASTRUCTURE ONLY -- the training step as a torch-CPU autograd graph:
the NumPy graph restatements of oracle/net_oracle.py executed with differentiable torch ops in
TRAINING mode (BatchNorm on batch statistics, Keras momentum 0.999 / eps 1e-3 moving-average
update), the torch restatement of the loss (oracle/loss_oracle.py) and a NumPy Adam in TF's
ApplyAdam form.  Only tests/ may import this module.  PARITY UNPINNED (no TensorFlow here): what
Keras / TF do in ``fit`` is restated from source knowledge ([3P]): fused BatchNorm normalises with
the biased batch variance while the moving variance follows the Bessel-corrected one (the fused op's
second output; Keras' BatchNormalization keeps it: ``_bessels_correction_test_only = True``);
ReluGrad / Relu6Grad pass the gradient strictly inside (0, 6); Adam eps 1e-7."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import loss_oracle as lo
from oracle import net_oracle as no
from oracle import torch_cpu_graph as tg

BN_MOMENTUM = 0.999


class _MaskedAct(torch.autograd.Function):
    """clamp(x, lo, hi) whose backward uses an externally supplied pass-mask."""
    @staticmethod
    def forward(ctx, x, lo, hi, mask):
        ctx.save_for_backward(mask)
        return x.clamp(lo, hi) if hi is not None else x.clamp(min=lo)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None, None, None


def mobilenet_v2_act_names():
    """Names of the ReLU / ReLU6 outputs of the MobileNetV2-SSD graph in call order."""
    names = ["Conv1_relu", "expanded_conv_depthwise_relu"]
    for k in range(1, 17):
        names += ["block_%d_expand_relu" % k, "block_%d_depthwise_relu" % k]
    names.append("out_relu")
    for i in range(1, 5):
        names += ["extra%d_1" % i, "extra%d_2" % i]
    return names


def vgg16_act_names():
    """Names of the ReLU outputs of the VGG16-SSD graph in call order (models/ssd_vgg16.py:52-91)."""
    names = [n for n, _, _ in no._VGG] + ["conv6", "conv7"]
    for i in range(8, 12):
        names += ["conv%d_1" % i, "conv%d_2" % i]
    return names


def act_names(backbone):
    return mobilenet_v2_act_names() if backbone == "mobilenet_v2" else vgg16_act_names()


L2_REG = 5e-4       # models/ssd_vgg16.py:45: kernel_regularizer=l2(5e-4) on the backbone / extra convs


class TrainOps(tg.TorchOps):
    """tg.TorchOps with differentiable training-mode ops; tensors stay tensors."""
    moving = None       # list of (moving_mean tensor, new mean, moving_var tensor, new var)
    masks = None        # optional FIFO of NHWC pass-masks (one per ReLU / ReLU6 call, in call order)

    @staticmethod
    def _act(x, lo, hi):
        if TrainOps.masks is None:
            return F.relu6(x) if hi is not None else torch.relu(x)       # hardtanh / relu backward == TF
        m = torch.from_numpy(np.ascontiguousarray(TrainOps.masks.pop(0), dtype=np.float32)).permute(0, 3, 1, 2)
        assert m.shape == x.shape
        return _MaskedAct.apply(x, lo, hi, m)

    @staticmethod
    def output(x):
        # 4-D taps stay NCHW graph nodes with retained gradients (activation-gradient parity);
        # the [B,N,K] network outputs pass through
        if x.dim() == 4:
            x.retain_grad()
        return x

    @staticmethod
    def batch_norm(x, gamma, beta, mean, var, eps=no.BN_EPS):
        mu = x.mean(dim=(0, 2, 3))
        va = ((x - mu.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))           # biased
        m = x.numel() // x.shape[1]
        TrainOps.moving.append((mean, mu.detach(), var, va.detach() * np.float32(m / max(m - 1, 1))))
        xh = (x - mu.view(1, -1, 1, 1)) * torch.rsqrt(va + eps).view(1, -1, 1, 1)
        return xh * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)

    @staticmethod
    def relu6(x):
        return TrainOps._act(x, 0.0, 6.0)

    @staticmethod
    def relu(x):
        return TrainOps._act(x, 0.0, None)

    @staticmethod
    def softmax(x):
        return lo.keras_softmax(x)


def train_step(backbone, hyper_params, P, x, actual_deltas, actual_labels, neg_pos_ratio=3.0, loc_loss_alpha=1.0,
               final_mask=None, threads=None, act_masks=None):
    """One forward/backward.  P: dict name -> float32 array (Keras layouts).  Returns dict with
    per-image ``loc`` / ``conf``, ``deltas`` / ``probs`` (network outputs), ``grads`` {name: array}
    for every trainable parameter (d mean_b(loc_b + conf_b)), and ``moving`` {name: new value}.
    ``final_mask`` / ``act_masks`` synchronise the NON-DIFFERENTIABLE selections (hard-negative
    mask; ReLU / ReLU6 pass-masks, one NHWC array per activation in call order) with another
    implementation: a single fp32-noise flip of such a mask changes gradients by percents (the loss
    is only piecewise smooth), which would otherwise drown the comparison of the arithmetic."""
    if threads:
        torch.set_num_threads(threads)
    T = {}
    for name, value in P.items():
        t = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32))
        if not (name.endswith("moving_mean") or name.endswith("moving_variance")):
            t.requires_grad_(True)
        T[name] = t
    TrainOps.moving = []
    TrainOps.masks = list(act_masks) if act_masks is not None else None
    acts = {}
    deltas, probs = no.forward(backbone, hyper_params, T, x, acts, ops=TrainOps)
    assert not TrainOps.masks, "unused activation masks"
    TrainOps.masks = None
    yd = torch.from_numpy(np.asarray(actual_deltas, np.float32))
    yl = torch.from_numpy(np.asarray(actual_labels, np.float32))
    loc, conf = lo.torch_loss(yd, yl, deltas, probs, neg_pos_ratio, loc_loss_alpha, final_mask)
    total = (loc + conf).mean()
    if backbone == "vgg16":        # Keras adds the layers' regularisation losses to the objective
        for name, t in T.items():
            if name.endswith("/kernel") and not name[0].isdigit():
                total = total + L2_REG * (t * t).sum()
    total.backward()
    ids = {id(t): n for n, t in T.items()}
    moving = {}
    for mm, mu, mv, va in TrainOps.moving:
        moving[ids[id(mm)]] = (mm - (mm - mu) * np.float32(1.0 - BN_MOMENTUM)).numpy()
        moving[ids[id(mv)]] = (mv - (mv - va) * np.float32(1.0 - BN_MOMENTUM)).numpy()
    grads = {n: t.grad.numpy() for n, t in T.items() if t.requires_grad and t.grad is not None}
    act_grads = {n: t.grad.permute(0, 2, 3, 1).contiguous().numpy() for n, t in acts.items()
                 if isinstance(t, torch.Tensor) and t.dim() == 4 and t.grad is not None}
    act_vals = {n: t.detach().permute(0, 2, 3, 1).contiguous().numpy() for n, t in acts.items()
                if isinstance(t, torch.Tensor) and t.dim() == 4}
    return {"loc": loc.detach().numpy(), "conf": conf.detach().numpy(), "deltas": deltas.detach().numpy(),
            "probs": probs.detach().numpy(), "grads": grads, "moving": moving, "act_grads": act_grads, "acts": act_vals}


def adam_step(var, m, v, g, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """[3P] TF ApplyAdam (Keras Adam, non-amsgrad), float32 state, step t >= 1."""
    f = np.float32
    alpha = f(lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    m = (m + (g - m) * f(1.0 - f(b1))).astype(f)
    v = (v + (g * g - v) * f(1.0 - f(b2))).astype(f)
    var = (var - alpha * m / (np.sqrt(v) + f(eps))).astype(f)
    return var, m, v
