"""TEST INFRASTRUCTURE ONLY -- the training step as a torch-CPU autograd graph:
the NumPy graph restatements of oracle/net_oracle.py executed with differentiable torch ops in
TRAINING mode (BatchNorm on batch statistics, Keras momentum 0.999 / eps 1e-3 moving-average
update), the torch restatement of the loss (oracle/loss_oracle.py) and a NumPy Adam in TF's
ApplyAdam form.  Only tests/ may import this module.  PARITY UNPINNED (no TensorFlow here): what
Keras / TF do in ``fit`` is restated from source knowledge ([3P]): fused BatchNorm normalises with
the biased batch variance and Keras removes Bessel's correction before the moving-average update;
ReluGrad / Relu6Grad pass the gradient strictly inside (0, 6); Adam eps 1e-7."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import loss_oracle as lo
from oracle import net_oracle as no
from oracle import torch_cpu_graph as tg

BN_MOMENTUM = 0.999


class TrainOps(tg.TorchOps):
    """tg.TorchOps with differentiable training-mode ops; tensors stay tensors."""
    moving = None       # list of (moving_mean tensor, new mean, moving_var tensor, new var)

    @staticmethod
    def output(x):
        return x.permute(0, 2, 3, 1) if x.dim() == 4 else x

    @staticmethod
    def batch_norm(x, gamma, beta, mean, var, eps=no.BN_EPS):
        mu = x.mean(dim=(0, 2, 3))
        va = ((x - mu.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))           # biased
        TrainOps.moving.append((mean, mu.detach(), var, va.detach()))
        xh = (x - mu.view(1, -1, 1, 1)) * torch.rsqrt(va + eps).view(1, -1, 1, 1)
        return xh * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)

    @staticmethod
    def relu6(x):
        return F.relu6(x)           # hardtanh backward: zero at and beyond 0 / 6 (== TF Relu6Grad)

    @staticmethod
    def softmax(x):
        return lo.keras_softmax(x)


def train_step(backbone, hyper_params, P, x, actual_deltas, actual_labels, neg_pos_ratio=3.0, loc_loss_alpha=1.0,
               final_mask=None, threads=None):
    """One forward/backward.  P: dict name -> float32 array (Keras layouts).  Returns dict with
    per-image ``loc`` / ``conf``, ``deltas`` / ``probs`` (network outputs), ``grads`` {name: array}
    for every trainable parameter (d mean_b(loc_b + conf_b)), and ``moving`` {name: new value}."""
    if threads:
        torch.set_num_threads(threads)
    T = {}
    for name, value in P.items():
        t = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32))
        if not (name.endswith("moving_mean") or name.endswith("moving_variance")):
            t.requires_grad_(True)
        T[name] = t
    TrainOps.moving = []
    deltas, probs = no.forward(backbone, hyper_params, T, x, ops=TrainOps)
    yd = torch.from_numpy(np.asarray(actual_deltas, np.float32))
    yl = torch.from_numpy(np.asarray(actual_labels, np.float32))
    loc, conf = lo.torch_loss(yd, yl, deltas, probs, neg_pos_ratio, loc_loss_alpha, final_mask)
    (loc + conf).mean().backward()
    ids = {id(t): n for n, t in T.items()}
    moving = {}
    for mm, mu, mv, va in TrainOps.moving:
        moving[ids[id(mm)]] = (mm - (mm - mu) * np.float32(1.0 - BN_MOMENTUM)).numpy()
        moving[ids[id(mv)]] = (mv - (mv - va) * np.float32(1.0 - BN_MOMENTUM)).numpy()
    grads = {n: t.grad.numpy() for n, t in T.items() if t.requires_grad and t.grad is not None}
    return {"loc": loc.detach().numpy(), "conf": conf.detach().numpy(), "deltas": deltas.detach().numpy(),
            "probs": probs.detach().numpy(), "grads": grads, "moving": moving}


def adam_step(var, m, v, g, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """[3P] TF ApplyAdam (Keras Adam, non-amsgrad), float32 state, step t >= 1."""
    f = np.float32
    alpha = f(lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    m = (m + (g - m) * f(1.0 - f(b1))).astype(f)
    v = (v + (g * g - v) * f(1.0 - f(b2))).astype(f)
    var = (var - alpha * m / (np.sqrt(v) + f(eps))).astype(f)
    return var, m, v
