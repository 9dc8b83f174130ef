"""Survival losses with the reference's class names (utils/loss_utils.py:114-139 in the reference);
`train_loop_survival` dispatches on these classes with isinstance (utils/core_utils.py:202-208).
The loss value and its gradients come from the HIP kernels (mmf_nll_surv / mmf_cox_surv)."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def nll_loss(hazards, S, Y, c, alpha=0.4, eps=1e-7):
    """utils/loss_utils.py:22-39.  S=None recomputes the survival curve from the hazards."""
    if S is None:
        S = torch.cumprod(1 - hazards, dim=1)
    return ops.nll_surv(hazards, S, Y, c, alpha=alpha, eps=eps)


class NLLSurvLoss(object):
    def __init__(self, alpha=0.15):
        self.alpha = alpha

    def __call__(self, hazards, S, Y, c, alpha=None):
        return nll_loss(hazards, S, Y, c, alpha=self.alpha if alpha is None else alpha)


class CoxSurvLoss(object):
    def __call__(self, risks, times, c, **kwargs):
        if not torch.is_tensor(times):
            times = torch.as_tensor(np.asarray(times), dtype=torch.float64)
        return ops.cox_surv(risks, times, c)


def ranking_loss(risks, times, c, phi, reduction):
    """utils/loss_utils.py:58-101 (pairwise loop over the batch) as one HIP launch."""
    return ops.ranking_loss(risks, times, c, phi=phi, reduction=reduction)


class RankingSurvLoss(object):
    """utils/loss_utils.py:141-149."""

    def __init__(self, phi="sigmoid", reduction="mean"):
        super().__init__()
        self.phi = phi
        self.reduction = reduction

    def __call__(self, risks, times, c):
        return ranking_loss(risks, times, c, self.phi, self.reduction)


class RankingNLLSurvLoss(object):
    """utils/loss_utils.py:151-164: ranking over the BIN LABELS (`ranking_loss(risks, Y, c, ...)`, :160) + nll_ratio * nll."""

    def __init__(self, phi="sigmoid", reduction="mean", alpha=0.15, nll_ratio=0.5):
        self.alpha = alpha
        self.phi = phi
        self.reduction = reduction
        self.nll_ratio = nll_ratio

    def __call__(self, hazards, risks, S, Y, c, alpha=None):
        ranking_ls = ranking_loss(risks, Y, c, self.phi, self.reduction)
        nll_ls = nll_loss(hazards, S, Y, c, alpha=self.alpha if alpha is None else alpha)
        return ranking_ls + nll_ls * self.nll_ratio
