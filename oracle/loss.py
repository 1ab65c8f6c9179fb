"""Oracle: training loss (experiments/utils.py:7-24).  TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn.functional as F


def calc_loss(pred, true, smoothing=True):
    true = true.reshape(-1)
    if not smoothing:
        return F.cross_entropy(pred, true, reduction='mean')
    eps, n_class = 0.2, pred.size(1)
    target = torch.full_like(pred, eps / (n_class - 1))
    target.scatter_(1, true.view(-1, 1), 1 - eps)
    return -(target * F.log_softmax(pred, dim=1)).sum(1).mean()
