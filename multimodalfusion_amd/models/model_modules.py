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


def snn_stack(fc_omic, x, training, seed=None):
    """Sequential of SNN_Blocks (Linear + SELU + AlphaDropout) on the GPU; block i uses dropout site i."""
    from .. import ops
    if training and seed is None:
        seed = ops.next_dropout_seed()
    f = x
    for i, blk in enumerate(fc_omic):
        lin, adrop = blk[0], blk[2]
        f = ops.dense(f, lin.weight, lin.bias, act="selu", drop_kind="alpha" if training else "none",
                      drop_p=adrop.p if training else 0.0, seed=seed or 0, site=i)
    return f


class XlinearFusion(nn.Module):
    """Gated Kronecker ("tensor") fusion; drop-in for models/model_modules.py:128-178 of the reference
    (same ctor signature and submodule tree: reduce.{i}.{0,1,2}.0, encoder1.0, encoder2.0).

    Deviations from the reference, both of which make it runnable rather than change its maths:
    the appended ones are created on the input's device (the reference hard-codes
    torch.cuda.FloatTensor, :164); use_bilinear and gate=0 are not provided (the reference indexes past
    the ModuleList for gate=0, :145-148 vs :161-163)."""

    def __init__(self, skip=1, use_bilinear=0, gate=1, dim=256, scale_dim=16, num_modalities=4,
                 mmhid1=256, mmhid2=256, dropout_rate=0.25):
        super().__init__()
        if use_bilinear or not gate:
            raise NotImplementedError("XlinearFusion: only gate=1, use_bilinear=0 (the configuration the heads use)")
        self.skip = skip
        self.use_bilinear = use_bilinear
        self.gate = gate
        self.num_modalities = num_modalities
        self.dropout_rate = dropout_rate
        dim_og, dim = dim, dim // scale_dim
        skip_dim = dim_og * self.num_modalities if skip else 0
        reduce = []
        for _ in range(self.num_modalities):
            linear_h = nn.Sequential(nn.Linear(dim_og, dim), nn.ReLU())
            linear_z = nn.Sequential(nn.Linear(dim_og * self.num_modalities, dim))
            linear_o = nn.Sequential(nn.Linear(dim, dim), nn.ReLU(), nn.Dropout(p=dropout_rate))
            reduce.append(nn.ModuleList([linear_h, linear_z, linear_o]))
        self.reduce = nn.ModuleList(reduce)
        self.post_fusion_dropout = nn.Dropout(p=dropout_rate)
        self.encoder1 = nn.Sequential(nn.Linear((dim + 1) ** num_modalities, mmhid1), nn.ReLU(),
                                      nn.Dropout(p=dropout_rate))
        self.encoder2 = nn.Sequential(nn.Linear(mmhid1 + skip_dim, mmhid2), nn.ReLU(), nn.Dropout(p=dropout_rate))

    def forward(self, v_list: list, seed=None):
        """Dropout sites under one seed: o_i -> i, post-fusion -> 8, encoder1 -> 9, encoder2 -> 10."""
        import torch
        from .. import ops
        tr = self.training
        if tr and seed is None:
            seed = ops.next_dropout_seed()
        seed = seed or 0
        p = self.dropout_rate if tr else 0.0
        if self.skip:          # the configuration the heads use: the whole block as one autograd node
            weights = []
            for i in range(len(v_list)):
                for lin in (self.reduce[i][0][0], self.reduce[i][1][0], self.reduce[i][2][0]):
                    weights += [lin.weight, lin.bias]
            weights += [self.encoder1[0].weight, self.encoder1[0].bias, self.encoder2[0].weight, self.encoder2[0].bias]
            return ops.xfusion(list(v_list), weights, p=p, seed=seed)
        kind = "dropout" if tr else "none"
        v_cat = torch.cat(v_list, dim=1)
        o_list = []
        for i, v in enumerate(v_list):
            lh, lz, lo = self.reduce[i][0][0], self.reduce[i][1][0], self.reduce[i][2][0]
            h = ops.dense(v, lh.weight, lh.bias, act="relu")
            z = ops.dense(v_cat, lz.weight, lz.bias)
            o = ops.dense(ops.gate_mul(z, h), lo.weight, lo.bias, act="relu", drop_kind=kind, drop_p=p, seed=seed, site=i)
            o_list.append(o)
        out = ops.kron_ones(o_list, drop_p=p, seed=seed, site=8)
        e1, e2 = self.encoder1[0], self.encoder2[0]
        out = ops.dense(out, e1.weight, e1.bias, act="relu", drop_kind=kind, drop_p=p, seed=seed, site=9)
        if self.skip:
            out = torch.cat([out] + list(v_list), dim=1)
        out = ops.dense(out, e2.weight, e2.bias, act="relu", drop_kind=kind, drop_p=p, seed=seed, site=10)
        return out


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
