"""Drop-in for the reference's ``ssd_loss.py`` -- ROW N1 ("next"), NOT at the parity bar yet.

Forward evaluation of the two SSD loss terms written with torch device ops so that
``trainer.py`` can report loss values; the HIP loss/backward kernels, Adam and the RCCL
gradient all-reduce of the training step are not built in this round."""
import torch


class CustomLoss(object):
    def __init__(self, neg_pos_ratio, loc_loss_alpha):
        self.neg_pos_ratio = float(neg_pos_ratio)
        self.loc_loss_alpha = float(loc_loss_alpha)

    def loc_loss_fn(self, actual_deltas, pred_deltas):
        """reference ssd_loss.py:8-33: Huber(delta=1) summed over the 4 coords, positives only
        (any non-zero target delta), normalised by max(#positives, 1) per sample."""
        err = pred_deltas - actual_deltas
        a = err.abs()
        huber = torch.where(a <= 1.0, 0.5 * err * err, a - 0.5).sum(-1)
        pos = (actual_deltas != 0).any(-1).float()
        total_pos = pos.sum(1)
        loss = (pos * huber).sum(-1) / torch.where(total_pos == 0, torch.ones_like(total_pos), total_pos)
        return loss * self.loc_loss_alpha

    def conf_loss_fn(self, actual_labels, pred_labels):
        """reference ssd_loss.py:35-65: categorical cross-entropy on probabilities (Keras
        renormalises and clips to [1e-7, 1-1e-7]) with 3:1 hard-negative mining by loss rank."""
        p = pred_labels / pred_labels.sum(-1, keepdim=True)
        p = p.clamp(1e-7, 1 - 1e-7)
        ce = -(actual_labels * p.log()).sum(-1)
        pos = (actual_labels[..., 1:] != 0).any(-1).float()
        total_pos = pos.sum(1)
        total_neg = (total_pos * self.neg_pos_ratio).to(torch.int32)
        masked = ce * actual_labels[..., 0]
        order = torch.argsort(masked, dim=-1, descending=True, stable=True)
        rank = torch.argsort(order, dim=-1, stable=True)
        neg = (rank < total_neg.unsqueeze(1)).float()
        loss = ((pos + neg) * ce).sum(-1) / torch.where(total_pos == 0, torch.ones_like(total_pos), total_pos)
        return loss
