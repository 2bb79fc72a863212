"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's training loss
(``ssd_loss.py:3-65``), NumPy float32 op for op, plus a torch-CPU autograd
restatement of the same graph used as the gradient oracle.  Only tests/ and bench.py's
cpu_baseline leg may import this module; the product (tf-ssd_amd/ssd_loss.py) never does.

PARITY UNPINNED: TensorFlow cannot be imported here and the reference ships no vectors, so the
[3P] pieces are restated from source knowledge of the pinned versions (TF 2.0.0,
environment.yml:70-73):
* ``tf.losses.Huber(reduction=NONE)`` -> ``huber_loss``: q = min(|e|, d); 0.5 q^2 + d (|e| - q)
  per coordinate, NO mean over the last axis before TF 2.2 (rank 3 -> the reference sums, :20-22;
  the TF >= 2.2 branch, mean * 4, is the same value up to rounding);
* ``tf.losses.CategoricalCrossentropy(reduction=NONE)`` on probabilities: the model's softmax is
  Keras' ``softmax`` for ndim > 2 (exp / reduce_sum, a RealDiv op, not a Softmax op), so Keras
  takes the probability branch: p / sum(p), clip to [1e-7, 1 - 1e-7], -sum(y * log(p));
* ``tf.argsort(direction="DESCENDING")`` orders equal keys by ascending index (top_k);
* ``tf.cast(float, int32)`` truncates.
"""
import numpy as np

F32 = np.float32
EPS = F32(1e-7)


def huber_sum(actual_deltas, pred_deltas, delta=1.0):
    """ssd_loss.py:18-23 (TF 2.0 huber_loss, then reduce_sum over the 4 coordinates)."""
    y, p = np.asarray(actual_deltas, F32), np.asarray(pred_deltas, F32)
    err = p - y
    a = np.abs(err)
    q = np.minimum(a, F32(delta))
    lin = a - q
    per = F32(0.5) * (q * q) + F32(delta) * lin
    out = per[..., 0].copy()
    for k in range(1, per.shape[-1]):
        out = out + per[..., k]
    return out.astype(F32)


def loc_loss_fn(actual_deltas, pred_deltas, loc_loss_alpha=1.0):
    """ssd_loss.py:8-33 -> [B]."""
    y = np.asarray(actual_deltas, F32)
    loss_all = huber_sum(y, pred_deltas)                             # :18-23
    pos_mask = np.any(y != F32(0), axis=2).astype(F32)               # :25-26
    total_pos = pos_mask.sum(axis=1, dtype=F32)                      # :27
    loc = (pos_mask * loss_all).sum(axis=-1, dtype=F32)              # :29
    total_pos = np.where(total_pos == 0, F32(1), total_pos)          # :30
    return (loc / total_pos * F32(loc_loss_alpha)).astype(F32)       # :31-33


def cross_entropy(actual_labels, pred_labels):
    """[3P] Keras categorical_crossentropy on probabilities (see module docstring) -> [B,N]."""
    y, p = np.asarray(actual_labels, F32), np.asarray(pred_labels, F32)
    s = p[..., 0].copy()
    for c in range(1, p.shape[-1]):
        s = s + p[..., c]
    out = np.clip(p / s[..., None], EPS, F32(1) - EPS)
    t = y * np.log(out)
    acc = t[..., 0].copy()
    for c in range(1, t.shape[-1]):
        acc = acc + t[..., c]
    return (-acc).astype(F32)


def hard_negative_mask(ce, y0, total_neg):
    """ssd_loss.py:54-57: rank of (ce * y0) in descending order (ties: lower index) < total_neg."""
    masked = (np.asarray(ce, F32) * np.asarray(y0, F32)).astype(F32)
    B, N = masked.shape
    neg = np.zeros((B, N), F32)
    for b in range(B):
        order = np.lexsort((np.arange(N), -masked[b].astype(np.float64)))      # descending, ties by index
        rank = np.empty(N, np.int64)
        rank[order] = np.arange(N)
        neg[b] = (rank < int(total_neg[b])).astype(F32)
    return neg


def conf_loss_fn(actual_labels, pred_labels, neg_pos_ratio=3.0, return_parts=False, ce=None):
    """ssd_loss.py:35-65 -> [B] (``ce``: use these per-anchor losses instead of recomputing)."""
    y = np.asarray(actual_labels, F32)
    ce = cross_entropy(y, pred_labels) if ce is None else np.asarray(ce, F32)      # :45-46
    pos_mask = np.any(y[..., 1:] != F32(0), axis=2).astype(F32)                     # :48-49
    total_pos = pos_mask.sum(axis=1, dtype=F32)                                     # :50
    total_neg = (total_pos * F32(neg_pos_ratio)).astype(np.int32)                   # :52
    neg_mask = hard_negative_mask(ce, y[..., 0], total_neg)                         # :54-58
    final_mask = pos_mask + neg_mask                                                # :60
    conf = (final_mask * ce).sum(axis=-1, dtype=F32)                                # :61
    total_pos = np.where(total_pos == 0, F32(1), total_pos)                         # :62
    out = (conf / total_pos).astype(F32)                                            # :63
    if return_parts:
        return out, ce, final_mask
    return out


def torch_loss(yd, yl, pd, probs, neg_pos_ratio=3.0, loc_loss_alpha=1.0, final_mask=None):
    """The two per-image loss terms on torch tensors (differentiable in pd / probs).  The
    hard-negative mask is non-differentiable (argsort) and computed in NumPy from the detached
    per-anchor losses unless ``final_mask`` [B,N] (pos + neg) is supplied."""
    import torch
    err = pd - yd
    a = err.abs()
    q = torch.minimum(a, torch.tensor(1.0))
    hub = (0.5 * q * q + 1.0 * (a - q)).sum(-1)
    pos = (yd != 0).any(-1).float()
    tp = pos.sum(1)
    loc = (pos * hub).sum(-1) / torch.where(tp == 0, torch.ones_like(tp), tp) * loc_loss_alpha
    out = torch.clamp(probs / probs.sum(-1, keepdim=True), 1e-7, 1 - 1e-7)
    ce = -(yl * torch.log(out)).sum(-1)
    cpos = (yl[..., 1:] != 0).any(-1).float()
    ctp = cpos.sum(1)
    if final_mask is None:
        tneg = (ctp * neg_pos_ratio).to(torch.int32)
        neg = torch.from_numpy(hard_negative_mask(ce.detach().numpy(), yl[..., 0].numpy(), tneg.numpy()))
        fm = cpos + neg
    else:
        fm = torch.as_tensor(np.asarray(final_mask, F32))
    conf = (fm * ce).sum(-1) / torch.where(ctp == 0, torch.ones_like(ctp), ctp)
    return loc, conf


def keras_softmax(z):
    """[3P] Keras ``softmax`` for ndim > 2 (TF 2.0/2.1): exp(x - max) / sum."""
    import torch
    e = torch.exp(z - z.max(-1, keepdim=True).values)
    return e / e.sum(-1, keepdim=True)


def torch_loss_and_grads(actual_deltas, actual_labels, pred_deltas, logits, neg_pos_ratio=3.0,
                         loc_loss_alpha=1.0):
    """Gradient oracle: the same graph in torch-CPU fp32 ops with autograd.  The model's softmax
    is part of the graph, so the returned gradients are w.r.t. pred_deltas and the LOGITS.
    Objective: Keras batch mean of loc_loss + conf_loss (compile(loss=[loc, conf]), reduction
    SUM_OVER_BATCH_SIZE).  Returns (loc [B], conf [B], probs, d/d pred_deltas, d/d logits)."""
    import torch
    yd = torch.from_numpy(np.asarray(actual_deltas, F32))
    yl = torch.from_numpy(np.asarray(actual_labels, F32))
    pd = torch.from_numpy(np.asarray(pred_deltas, F32)).requires_grad_(True)
    z = torch.from_numpy(np.asarray(logits, F32)).requires_grad_(True)
    probs = keras_softmax(z)
    loc, conf = torch_loss(yd, yl, pd, probs, neg_pos_ratio, loc_loss_alpha)
    (loc + conf).mean().backward()
    return (loc.detach().numpy(), conf.detach().numpy(), probs.detach().numpy(), pd.grad.numpy(), z.grad.numpy())
