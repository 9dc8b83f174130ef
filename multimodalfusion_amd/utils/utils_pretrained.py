"""Helpers of the stage-2 pipeline with the reference's names (utils/utils_pretrained.py)."""
from __future__ import annotations

import torch.nn as nn


def initialize_weights(module):
    """utils/utils_pretrained.py:145-154: xavier-normal Linear weights, zero biases, BatchNorm1d weight 1 / bias 0,
    visiting `module.modules()` in order (so the same torch seed gives the reference's initial state)."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight)
            m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm1d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
