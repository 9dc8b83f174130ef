"""Pathology attention-MIL head; drop-in for models/model_attention_mil_path.py of the reference
(same class names, ctor signatures :13 / :46, forward(**kwargs) contract :50-72, state_dict keys)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..utils.utils import initialize_weights
from .model_modules import AMIL_SIZES, amil_stack, amil_stack_head, amil_stack_nll_step, make_amil_stack


class MIL_Attention_fc_path(nn.Module):
    """Parameter container: `attention_net_WSI` (the stack) and `classifier`; the arithmetic lives in the HIP kernels."""

    def __init__(self, gate_path=True, dropout=True, model_size_wsi: str = "small", n_classes=4):
        super().__init__()
        self.size_dict_WSI = {name: list(dims) for name, dims in AMIL_SIZES.items()}
        self.attention_net_WSI = make_amil_stack(model_size_wsi, gated=gate_path, att_dropout=dropout)
        self.classifier = nn.Linear(AMIL_SIZES[model_size_wsi][1], n_classes)
        initialize_weights(self)

    def relocate(self):
        self.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))

    def forward(self, h, return_features=False, attention_only=False):
        pass            # abstract in the reference too (:42-43)


class MIL_Attention_fc_surv_path(MIL_Attention_fc_path):
    def __init__(self, gate_path=True, model_size_wsi: str = "small", dropout=False, n_classes=4):
        super().__init__(gate_path=gate_path, model_size_wsi=model_size_wsi, dropout=dropout, n_classes=n_classes)

    def forward(self, **kwargs):
        bag = kwargs["path_features"]
        want_embedding, want_scores = kwargs.get("return_features"), kwargs.get("attention_only")
        if want_embedding or want_scores:
            M, A_raw = amil_stack(self.attention_net_WSI, bag, self.training)
            return M if want_embedding else A_raw
        # (hazards, S, Y_hat, A_raw): stack + classifier + hazard head as one autograd node
        return amil_stack_head(self.attention_net_WSI, self.classifier, bag, self.training)

    def nll_step(self, path_features, label, c, alpha=0.0, loss_scale=1.0, grad_out=None, accumulate=None):
        """Extension of the reference surface (the training loop mirror uses it, utils/core_utils.py): forward +
        NLLSurvLoss(alpha) + backward of one bag in ONE C-ABI call -- what `hazards, S, Y_hat, A_raw = model(...)`,
        `loss = loss_fn(hazards=hazards, S=S, Y=label, c=c)`, `(loss * loss_scale).backward()` compute together
        (models/model_attention_mil_path.py:50-72 + utils/loss_utils.py:22-39 + autograd), with the same dropout draw.
        Gradients land in the parameters' .grad (accumulated) -- or in `grad_out`, tensors in self.parameters() order,
        overwritten unless `accumulate`; returns (hazards, S, Y_hat, A_raw, loss, risk)."""
        if any(not p.requires_grad for p in self.parameters()):
            raise RuntimeError("nll_step needs every parameter of the head to require grad")
        return amil_stack_nll_step(self.attention_net_WSI, self.classifier, path_features, self.training, label, c,
                                   alpha, loss_scale, grad_out, accumulate)
