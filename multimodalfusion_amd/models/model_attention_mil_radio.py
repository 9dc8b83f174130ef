"""Radiology attention-MIL head; drop-in for models/model_attention_mil_radio.py of the reference
(ctor signatures :14-15 / :67-68, forward(**kwargs) :73-115, state_dict keys)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.utils import initialize_weights
from .model_modules import AMIL_SIZES, amil_stack, amil_stack_head, make_amil_stack


class MIL_Attention_fc_radio(nn.Module):
    """Parameter container: `reduce_dim` (several modalities), `attention_net_radio`, `classifier`."""

    def __init__(self, radio_fusion="concat", gate_radio=True, dropout=True, model_size_radio: str = "small",
                 n_classes=4, modalities=["T1", "T2", "T1Gd", "FLAIR"]):
        super().__init__()
        self.radio_fusion, self.n_classes, self.modalities = radio_fusion, n_classes, modalities
        self.size_dict_radio = {name: list(dims) for name, dims in AMIL_SIZES.items()}
        feat, hidden, _ = AMIL_SIZES[model_size_radio]
        n_mod = len(modalities)
        if n_mod > 1:                           # created before the stack, as in the reference (:27-32): same RNG order
            if radio_fusion == "concat":
                self.reduce_dim = nn.Linear(feat * n_mod, feat)
            elif radio_fusion == "tensor":
                # unusable in the reference as well (its forward reads an attribute that is never set,
                # model_attention_mil_radio.py:84)
                raise NotImplementedError("radio_fusion='tensor' is unusable in the reference and not provided")
        self.attention_net_radio = make_amil_stack(model_size_radio, gated=gate_radio, att_dropout=dropout)
        self.classifier = nn.Linear(hidden, n_classes)
        initialize_weights(self)

    def relocate(self):
        self.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))

    def forward(self, h, return_features=False, attention_only=False):
        pass            # abstract in the reference too (:64-65)


class MIL_Attention_fc_surv_radio(MIL_Attention_fc_radio):
    def __init__(self, radio_fusion="concat", gate_radio=True, dropout=True, model_size_radio="small", n_classes=4,
                 modalities=["T1", "T2", "T1Gd", "FLAIR"]):
        model_size_radio = "small"              # the reference overrides the argument (:70)
        super().__init__(radio_fusion=radio_fusion, gate_radio=gate_radio, dropout=dropout,
                         model_size_radio=model_size_radio, n_classes=n_classes, modalities=modalities)

    def forward(self, **kwargs):
        bags = [kwargs[m] for m in self.modalities]
        # several modalities: cat(axis=1) + reduce_dim without materialising the concatenation (:80-82)
        x = ops.linear_cat(bags, self.reduce_dim.weight, self.reduce_dim.bias) if len(bags) > 1 else bags[0]
        flags = [kwargs.get(k) for k in ("attention_only", "return_features", "return_attention")]
        if any(flags):
            M, A_raw = amil_stack(self.attention_net_radio, x, self.training)
            return M if (flags[1] and not flags[0]) else A_raw      # attention_only wins, then features (:91-113)
        return amil_stack_head(self.attention_net_radio, self.classifier, x, self.training)
