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
