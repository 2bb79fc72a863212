"""torch-CPU (oneDNN, all host cores) op set for oracle/net_oracle.py's graphs.

TEST INFRASTRUCTURE / CPU BASELINE ONLY.  Two uses: (1) an independent implementation the
NumPy conv restatement is cross-checked against (tests/test_oracle_net.py); (2) the
multi-threaded "port" CPU baseline bench.py times beside the MI355X numbers -- the literal
TF-CPU reference path cannot be run here (TensorFlow is not installable: BASELINE.md 3).
Activations are NHWC NumPy-compatible memory viewed as channels_last NCHW tensors (no copies).
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import net_oracle as no


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class TorchOps(object):
    @staticmethod
    def input(x):
        return _t(x).permute(0, 3, 1, 2)            # NHWC memory == channels_last NCHW view

    @staticmethod
    def hw(x):
        return x.shape[2], x.shape[3]

    @staticmethod
    def output(x):
        if x.dim() == 4:
            return x.permute(0, 2, 3, 1).contiguous().numpy()
        return x.contiguous().numpy()

    @staticmethod
    def _pads(x, kh, kw, stride, dilation, padding):
        H, W = x.shape[2], x.shape[3]
        if padding == "same":
            _, pt, pb = no.same_pads(H, kh, stride, dilation)
            _, pl, pr = no.same_pads(W, kw, stride, dilation)
        elif padding == "valid":
            pt = pb = pl = pr = 0
        else:
            pt, pb, pl, pr = padding
        return pt, pb, pl, pr

    @staticmethod
    def conv2d(x, w, bias=None, stride=1, dilation=1, padding="same"):
        wt = _t(w).permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last)   # HWIO -> OIHW
        pt, pb, pl, pr = TorchOps._pads(x, w.shape[0], w.shape[1], stride, dilation, padding)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, wt, None if bias is None else _t(bias), stride=stride, dilation=dilation)

    @staticmethod
    def depthwise_conv2d(x, w, stride=1, padding="same"):
        C = w.shape[2]
        wt = _t(w).permute(2, 3, 0, 1).contiguous()                                      # [3,3,C,1] -> [C,1,3,3]
        pt, pb, pl, pr = TorchOps._pads(x, 3, 3, stride, 1, padding)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, wt, None, stride=stride, groups=C)

    @staticmethod
    def batch_norm(x, gamma, beta, mean, var, eps=no.BN_EPS):
        inv = _t(gamma) / torch.sqrt(_t(var) + eps)
        return x * inv.view(1, -1, 1, 1) + (_t(beta) - _t(mean) * inv).view(1, -1, 1, 1)

    @staticmethod
    def relu(x):
        return torch.relu(x)

    @staticmethod
    def relu6(x):
        return torch.clamp(x, 0.0, 6.0)

    @staticmethod
    def max_pool(x, k, stride, padding="same"):
        H, W = x.shape[2], x.shape[3]
        _, pt, pb = no.same_pads(H, k, stride)
        _, pl, pr = no.same_pads(W, k, stride)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
        return F.max_pool2d(x, k, stride)

    @staticmethod
    def l2_normalize_scale(x, gamma):
        sq = (x * x).sum(1, keepdim=True)
        return x * torch.rsqrt(torch.clamp(sq, min=1e-12)) * _t(gamma).view(1, -1, 1, 1)

    @staticmethod
    def softmax(x):
        return torch.softmax(x, -1)

    @staticmethod
    def add(a, b):
        return a + b

    @staticmethod
    def reshape(a, shape):
        if a.dim() == 4:
            a = a.permute(0, 2, 3, 1)      # back to NHWC order before flattening (models/header.py:39)
        return a.reshape(shape)

    @staticmethod
    def concat(parts, axis):
        return torch.cat(parts, axis)


def forward(backbone, hyper_params, P, x, threads=None):
    if threads:
        torch.set_num_threads(threads)
    with torch.no_grad():
        return no.forward(backbone, hyper_params, P, x, ops=TorchOps)
