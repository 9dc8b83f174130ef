"""Building blocks with the reference's names, constructor signatures and submodule trees
(models/model_modules.py:64-110), so state_dict keys and same-seed initialisation match.
They are parameter containers: the arithmetic of a whole attention stack runs in the fused
HIP kernels called by the heads (ops.amil_pool)."""
from __future__ import annotations

import torch.nn as nn


def SNN_Block(dim1, dim2, dropout=0.25):
    """models/model_modules.py:64-68."""
    return nn.Sequential(nn.Linear(dim1, dim2), nn.SELU(), nn.AlphaDropout(p=dropout, inplace=False))


class Attn_Net(nn.Module):
    """models/model_modules.py:70-85: module = [Linear(L,D), Tanh, (Dropout), Linear(D,1)]."""

    def __init__(self, L=1024, D=256, dropout=False, n_classes=1):
        super().__init__()
        mods = [nn.Linear(L, D), nn.Tanh()]
        if dropout:
            mods.append(nn.Dropout(0.25))
        mods.append(nn.Linear(D, n_classes))
        self.module = nn.Sequential(*mods)
        self.att_dropout = bool(dropout)

    def stack_params(self):
        a, c = self.module[0], self.module[-1]
        return a.weight, a.bias, None, None, c.weight, c.bias

    def forward(self, x):
        raise NotImplementedError(
            "Attn_Net runs fused inside the attention-MIL heads (ops.amil_pool); call the head's forward")


class Attn_Net_Gated(nn.Module):
    """models/model_modules.py:87-110: attention_a = [Linear, Tanh, (Dropout)], attention_b = [Linear, Sigmoid,
    (Dropout)], attention_c = Linear(D, 1)."""

    def __init__(self, L=1024, D=256, dropout=False, n_classes=1):
        super().__init__()
        a = [nn.Linear(L, D), nn.Tanh()]
        b = [nn.Linear(L, D), nn.Sigmoid()]
        if dropout:
            a.append(nn.Dropout(0.25))
            b.append(nn.Dropout(0.25))
        self.attention_a = nn.Sequential(*a)
        self.attention_b = nn.Sequential(*b)
        self.attention_c = nn.Linear(D, n_classes)
        self.att_dropout = bool(dropout)

    def stack_params(self):
        a, b, c = self.attention_a[0], self.attention_b[0], self.attention_c
        return a.weight, a.bias, b.weight, b.bias, c.weight, c.bias

    def forward(self, x):
        raise NotImplementedError(
            "Attn_Net_Gated runs fused inside the attention-MIL heads (ops.amil_pool); call the head's forward")


def amil_stack(seq, x, training):
    """Run Sequential(Linear, ReLU, Dropout(0.25), Attn_Net*) + softmax pooling on the GPU.
    Returns (M [1 x H], A_raw [1 x N]).  Dropout probabilities follow nn.Module.training exactly as
    the reference's nn.Dropout layers do (the 0.25 after the ReLU is always there in train mode;
    the two inside the attention net only when it was built with dropout=True)."""
    from .. import ops
    lin, att = seq[0], seq[3]
    gated = isinstance(att, Attn_Net_Gated)
    Wa, ba, Wb, bb, Wc, bc = att.stack_params()
    p_h = seq[2].p if training else 0.0
    p_att = 0.25 if (training and att.att_dropout) else 0.0
    seed = ops.next_dropout_seed() if training else 0
    return ops.amil_pool(x, lin.weight, lin.bias, Wa, ba, Wb, bb, Wc, bc, gated, p_h, p_att, seed)
