"""Radiology attention-MIL head; drop-in for models/model_attention_mil_radio.py of the reference
(ctor signatures :14-15 / :67-68, forward(**kwargs) :73-115, state_dict keys)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.utils import initialize_weights
from .model_modules import Attn_Net, Attn_Net_Gated, amil_stack, amil_stack_head


class MIL_Attention_fc_radio(nn.Module):
    def __init__(self, radio_fusion="concat", gate_radio=True, dropout=True, model_size_radio: str = "small",
                 n_classes=4, modalities=["T1", "T2", "T1Gd", "FLAIR"]):
        super().__init__()
        self.radio_fusion = radio_fusion
        self.n_classes = n_classes
        self.size_dict_radio = {"small": [1024, 256, 256], "big": [1024, 512, 384]}
        self.modalities = modalities
        size_radio = self.size_dict_radio[model_size_radio]
        if len(self.modalities) > 1:
            if self.radio_fusion == "tensor":
                # unreachable in the reference as well (forward uses an undefined attribute,
                # model_attention_mil_radio.py:84)
                raise NotImplementedError("radio_fusion='tensor' is unusable in the reference and not provided")
            elif self.radio_fusion == "concat":
                self.reduce_dim = nn.Linear(size_radio[0] * len(self.modalities), size_radio[0])
        fc_radio = [nn.Linear(size_radio[0], size_radio[1]), nn.ReLU(), nn.Dropout(0.25)]
        if gate_radio:
            attention_net_radio = Attn_Net_Gated(L=size_radio[1], D=size_radio[2], dropout=dropout, n_classes=1)
        else:
            attention_net_radio = Attn_Net(L=size_radio[1], D=size_radio[2], dropout=dropout, n_classes=1)
        fc_radio.append(attention_net_radio)
        self.attention_net_radio = nn.Sequential(*fc_radio)
        self.classifier = nn.Linear(size_radio[1], n_classes)
        initialize_weights(self)

    def relocate(self):
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if len(self.modalities) > 1 and self.radio_fusion == "concat":
            self.reduce_dim = self.reduce_dim.to(device)
        self.attention_net_radio = self.attention_net_radio.to(device)
        self.classifier = self.classifier.to(device)

    def forward(self, h, return_features=False, attention_only=False):
        pass


class MIL_Attention_fc_surv_radio(MIL_Attention_fc_radio):
    def __init__(self, radio_fusion="concat", gate_radio=True, dropout=True, model_size_radio: str = "small",
                 n_classes=4, modalities=["T1", "T2", "T1Gd", "FLAIR"]):
        # the reference forces model_size_radio='small' here (model_attention_mil_radio.py:70)
        super().__init__(radio_fusion=radio_fusion, gate_radio=gate_radio, model_size_radio="small",
                         dropout=dropout, n_classes=n_classes, modalities=modalities)

    def forward(self, **kwargs):
        h = [kwargs[m] for m in self.modalities]
        if len(self.modalities) > 1:
            h = ops.linear_cat(h, self.reduce_dim.weight, self.reduce_dim.bias)   # cat(axis=1) + reduce_dim
        else:
            h = h[0]
        if kwargs.get("attention_only") or kwargs.get("return_features") or kwargs.get("return_attention"):
            M, A_raw = amil_stack(self.attention_net_radio, h, self.training)
            if kwargs.get("attention_only"):
                return A_raw
            return M if kwargs.get("return_features") else A_raw
        return amil_stack_head(self.attention_net_radio, self.classifier, h, self.training)
