"""Pathology attention-MIL head; drop-in for models/model_attention_mil_path.py of the reference
(same class names, ctor signatures :13 / :46, forward(**kwargs) contract :50-72, state_dict keys)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.utils import initialize_weights
from .model_modules import Attn_Net, Attn_Net_Gated, amil_stack, amil_stack_head


class MIL_Attention_fc_path(nn.Module):
    def __init__(self, gate_path=True, dropout=True, model_size_wsi: str = "small", n_classes=4):
        super().__init__()
        self.size_dict_WSI = {"small": [1024, 256, 256], "big": [1024, 512, 384]}
        size_WSI = self.size_dict_WSI[model_size_wsi]
        fc_WSI = [nn.Linear(size_WSI[0], size_WSI[1]), nn.ReLU(), nn.Dropout(0.25)]
        if gate_path:
            attention_net_WSI = Attn_Net_Gated(L=size_WSI[1], D=size_WSI[2], dropout=dropout, n_classes=1)
        else:
            attention_net_WSI = Attn_Net(L=size_WSI[1], D=size_WSI[2], dropout=dropout, n_classes=1)
        fc_WSI.append(attention_net_WSI)
        self.attention_net_WSI = nn.Sequential(*fc_WSI)
        self.classifier = nn.Linear(size_WSI[1], n_classes)
        initialize_weights(self)

    def relocate(self):
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.attention_net_WSI = self.attention_net_WSI.to(device)
        self.classifier = self.classifier.to(device)

    def forward(self, h, return_features=False, attention_only=False):
        pass


class MIL_Attention_fc_surv_path(MIL_Attention_fc_path):
    def __init__(self, gate_path=True, model_size_wsi: str = "small", dropout=False, n_classes=4):
        super().__init__(gate_path=gate_path, model_size_wsi=model_size_wsi, dropout=dropout, n_classes=n_classes)

    def forward(self, **kwargs):
        h = kwargs["path_features"]
        if kwargs.get("return_features") or kwargs.get("attention_only"):
            M, A_raw = amil_stack(self.attention_net_WSI, h, self.training)
            return M if kwargs.get("return_features") else A_raw
        return amil_stack_head(self.attention_net_WSI, self.classifier, h, self.training)
