"""Drop-in for the reference's ``ssd_loss.py``: ``CustomLoss`` with
``loc_loss_fn`` / ``conf_loss_fn``.  Both terms are evaluated by ONE HIP kernel per image
(``ssd_loss`` in include/ssd_hip.h, csrc/ssd_loss.hip): Huber over the positives, categorical
cross-entropy on renormalised + clipped probabilities, 3:1 hard-negative mining through a
radix select of the loss rank, per-image normalisation.  No torch math on this path."""
import torch

import ssd_hip as _h


class CustomLoss(object):
    """reference ssd_loss.py:3-65."""

    def __init__(self, neg_pos_ratio, loc_loss_alpha):
        self.neg_pos_ratio = float(neg_pos_ratio)
        self.loc_loss_alpha = float(loc_loss_alpha)
        self.last_final_mask = None        # diagnostics: the reference's final_mask of the last conf call
        self.last_cross_entropy = None

    def _run(self, yd, pd, yl, pl, want_aux=False, want_grads=False, grad_scale=1.0):
        ref = pd if pd is not None else pl
        B, N = ref.shape[0], ref.shape[1]
        L = pl.shape[2] if pl is not None else 1
        dev = ref.device
        lib = _h.lib()
        loc = torch.empty((B,), dtype=torch.float32, device=dev) if pd is not None else None
        conf = torch.empty((B,), dtype=torch.float32, device=dev) if pl is not None else None
        ce = torch.empty((B, N), dtype=torch.float32, device=dev) if (want_aux and pl is not None) else None
        mask = torch.empty((B, N), dtype=torch.float32, device=dev) if (want_aux and pl is not None) else None
        gd = torch.empty_like(pd) if (want_grads and pd is not None) else None
        gz = torch.empty_like(pl) if (want_grads and pl is not None) else None
        ws = _h.workspace(lib.ssd_loss_workspace_bytes(B, N))
        _h.check(lib.ssd_loss(_h.ptr(yd), _h.ptr(pd), _h.ptr(yl), _h.ptr(pl), B, N, L, self.neg_pos_ratio,
                              self.loc_loss_alpha, _h.ptr(loc), _h.ptr(conf), _h.ptr(ce), _h.ptr(mask), _h.ptr(gd),
                              _h.ptr(gz), float(grad_scale), _h.ptr(ws), ws.numel(), _h.stream()), "ssd_loss")
        return loc, conf, ce, mask, gd, gz

    @staticmethod
    def _pair(actual, pred, last):
        a, p = _h.to_dev(actual), _h.to_dev(pred)
        if a.dim() != 3 or a.shape != p.shape or (last and a.shape[2] != last):
            raise ValueError("bad shapes %s / %s" % (tuple(a.shape), tuple(p.shape)))
        return a, p

    def loc_loss_fn(self, actual_deltas, pred_deltas):
        """reference ssd_loss.py:8-33 -> loc_loss [B]."""
        yd, pd = self._pair(actual_deltas, pred_deltas, 4)
        return self._run(yd, pd, None, None)[0]

    def conf_loss_fn(self, actual_labels, pred_labels):
        """reference ssd_loss.py:35-65 -> conf_loss [B]."""
        yl, pl = self._pair(actual_labels, pred_labels, 0)
        _, conf, ce, mask, _, _ = self._run(None, None, yl, pl, want_aux=True)
        self.last_cross_entropy, self.last_final_mask = ce, mask
        return conf

    def loss_and_grads(self, actual_deltas, actual_labels, pred_deltas, pred_labels, grad_scale=None):
        """Both terms + the gradients of the Keras objective mean_b(loc_b + conf_b) w.r.t. the
        predicted deltas and the softmax LOGITS (what the training step back-propagates)."""
        yd, pd = self._pair(actual_deltas, pred_deltas, 4)
        yl, pl = self._pair(actual_labels, pred_labels, 0)
        gs = 1.0 / pd.shape[0] if grad_scale is None else grad_scale
        loc, conf, _, _, gd, gz = self._run(yd, pd, yl, pl, want_grads=True, grad_scale=gs)
        return loc, conf, gd, gz
