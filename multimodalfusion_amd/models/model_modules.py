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
        """(A [N x n_classes], x), as the reference (models/model_modules.py:84-85).  Inside a head the scorer runs fused
        with the projection and the pooling (ops.amil_head); called on its own it runs the same K-gate / K-dh / K-tn
        kernels through mmf_attn_net_forward / _backward."""
        from .. import ops
        Wa, ba, Wb, bb, Wc, bc = self.stack_params()
        if Wc.shape[0] != 1:
            raise NotImplementedError("the HIP attention scorer supports n_classes = 1 (every use in the reference)")
        p_att = 0.25 if (self.training and self.att_dropout) else 0.0
        seed = ops.next_dropout_seed() if p_att > 0 else 0
        return ops.attn_net(x, Wa, ba, Wb, bb, Wc, bc, False, p_att, seed), x


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
        """(A [N x n_classes], x), as the reference (models/model_modules.py:105-110).  Inside a head the scorer runs fused
        with the projection and the pooling (ops.amil_head); called on its own it runs the same K-gate / K-dh / K-tn
        kernels through mmf_attn_net_forward / _backward."""
        from .. import ops
        Wa, ba, Wb, bb, Wc, bc = self.stack_params()
        if Wc.shape[0] != 1:
            raise NotImplementedError("the HIP attention scorer supports n_classes = 1 (every use in the reference)")
        p_att = 0.25 if (self.training and self.att_dropout) else 0.0
        seed = ops.next_dropout_seed() if p_att > 0 else 0
        return ops.attn_net(x, Wa, ba, Wb, bb, Wc, bc, True, p_att, seed), x


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


def _drop_args(module_training, p, seed):
    from .. import ops
    if not module_training:
        return 0.0, 0
    return p, (ops.next_dropout_seed() if seed is None else seed)


class Highway(nn.Module):
    """models/model_modules.py:5-27 of the reference (same ctor, same submodule tree: nonlinear.{i}, linear.{i}, gate.{i},
    bn1, bn2, dropout1).  bn1 + dropout1 and bn2 are one launch each, every layer is three dense launches + one mix.
    Only f = relu is provided (the only value the reference passes)."""

    def __init__(self, size, num_layers, f=None):
        super().__init__()
        self.num_layers = num_layers
        self.nonlinear = nn.ModuleList([nn.Linear(size, size) for _ in range(num_layers)])
        self.linear = nn.ModuleList([nn.Linear(size, size) for _ in range(num_layers)])
        self.gate = nn.ModuleList([nn.Linear(size, size) for _ in range(num_layers)])
        self.f = f
        self.bn1 = nn.BatchNorm1d(size)
        self.bn2 = nn.BatchNorm1d(size)
        self.dropout1 = nn.Dropout(0.7)

    def forward(self, x, seed=None):
        from .. import ops
        p, seed = _drop_args(self.training, self.dropout1.p, seed)
        x = ops.batchnorm(x, self.bn1, drop_p=p, seed=seed, site=0)
        for layer in range(self.num_layers):
            zg = ops.dense(x, self.gate[layer].weight, self.gate[layer].bias)
            zn = ops.dense(x, self.nonlinear[layer].weight, self.nonlinear[layer].bias)
            zl = ops.dense(x, self.linear[layer].weight, self.linear[layer].bias)
            x = ops.highway_mix(zg, zn, zl)
        return ops.batchnorm(x, self.bn2)


class ResidualBlock(nn.Module):
    """models/model_modules.py:29-49: fc1-bn1-relu-fc2-bn2, += residual, relu (bn2 + add + relu is one launch)."""

    def __init__(self, size):
        super().__init__()
        self.fc1 = nn.Linear(size, size)
        self.bn1 = nn.BatchNorm1d(size)
        self.relu = nn.ReLU(inplace=True)
        self.fc2 = nn.Linear(size, size)
        self.bn2 = nn.BatchNorm1d(size)

    def forward(self, x):
        from .. import ops
        out = ops.batchnorm(ops.dense(x, self.fc1.weight, self.fc1.bias), self.bn1, act="relu")
        return ops.batchnorm(ops.dense(out, self.fc2.weight, self.fc2.bias), self.bn2, res=x, act="relu")


class Residual(nn.Module):
    """models/model_modules.py:51-58."""

    def __init__(self, size, n_layer):
        super().__init__()
        self.n_layer = n_layer
        self.blocks = nn.ModuleList([ResidualBlock(size) for _ in range(n_layer)])

    def forward(self, x):
        for i in range(self.n_layer):
            x = self.blocks[i](x)
        return x


def fcnn_block(seq, x, seed, site):
    """Sequential(Linear, BatchNorm1d, ReLU, Dropout(p)[, Linear]) of the stage-2 models on the GPU:
    dense, then BN + ReLU + dropout in one launch, then the optional last dense."""
    from .. import ops
    lin, bn, drop = seq[0], seq[1], seq[3]
    p = drop.p if seq.training else 0.0
    h = ops.batchnorm(ops.dense(x, lin.weight, lin.bias), bn, act="relu", drop_p=p, seed=seed, site=site)
    if len(seq) > 4:
        h = ops.dense(h, seq[4].weight, seq[4].bias)
    return h


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
        B = v_list[0].shape[0]
        sdim = self.reduce[0][0][0].weight.shape[0]
        # the configuration the heads use (skip, B = 1): the whole block as one autograd node; its single-workgroup
        # gating kernel holds m * B * sdim values in LDS, so larger batches (stage 2: B = 32) take the composable ops
        if self.skip and len(v_list) * B * sdim <= 384:
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


# feature dim, hidden dim, attention dim of the two stack sizes (model_attention_mil_path.py:16, ..._radio.py:20)
AMIL_SIZES = {"small": (1024, 256, 256), "big": (1024, 512, 384)}


def make_amil_stack(size: str, gated: bool, att_dropout: bool) -> nn.Sequential:
    """The attention stack every head owns: projection (Linear + ReLU + Dropout(0.25)) followed by the gated or plain
    attention scorer.  Built in the reference's order (projection first, scorer second), so a given torch seed
    initialises it identically and the state_dict keys are `<name>.0.*` and `<name>.3.*`."""
    feat, hidden, att = AMIL_SIZES[size]
    scorer = Attn_Net_Gated if gated else Attn_Net
    return nn.Sequential(nn.Linear(feat, hidden), nn.ReLU(), nn.Dropout(0.25),
                         scorer(L=hidden, D=att, dropout=att_dropout, n_classes=1))


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


def amil_stack_nll_step(seq, classifier, x, training, Y, c, alpha, loss_scale=1.0, grad_out=None, accumulate=None,
                        dx_out=None):
    """The whole training step of one bag -- stack, classifier / hazard head, nll_surv and the backward -- as ONE C-ABI
    call (ops.amil_nll_step), gradients of loss * loss_scale ADDED to the parameters' .grad exactly as
    `(loss * loss_scale).backward()` would (parameters whose .grad is None get a fresh buffer, as autograd does; when
    all of them are None the kernels write instead of accumulate and nothing is zero-filled).
    grad_out: instead of .grad, a list of gradient tensors in the order of [*seq.parameters(), *classifier.parameters()]
    (pipeline.BagsInFlight hands in views of a stream's gradient slot) with `accumulate` said explicitly.
    dx_out: an [N x L] fp32 tensor that receives the gradient with respect to the bag (overwritten), for a head whose bag
    is itself computed (the radio head's reduce_dim).
    Returns (hazards, S, Y_hat, A_raw, loss, risk), all detached."""
    import torch
    from .. import ops
    lin, att = seq[0], seq[3]
    gated = isinstance(att, Attn_Net_Gated)
    Wa, ba, Wb, bb, Wc, bc = att.stack_params()
    params = [lin.weight, lin.bias, Wa, ba, Wb, bb, Wc, bc, classifier.weight, classifier.bias]
    live = [p for p in params if p is not None]
    if grad_out is not None:
        it = iter(grad_out)
        grads = [None if p is None else next(it) for p in params]
        accumulate = bool(accumulate)
    else:
        missing = [p for p in live if p.grad is None]
        accumulate = len(missing) < len(live)
        if missing:
            n = sum(p.numel() for p in missing)
            flat = (torch.zeros if accumulate else torch.empty)(n, dtype=torch.float32, device=x.device)
            off = 0
            for p in missing:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        grads = [None if p is None else p.grad for p in params]
    p_h = seq[2].p if training else 0.0
    p_att = 0.25 if (training and att.att_dropout) else 0.0
    seed = ops.next_dropout_seed() if training else 0
    with torch.no_grad():
        return ops.amil_nll_step(x, (lin.weight, lin.bias, Wa, ba, Wb, bb, Wc, bc), classifier.weight, classifier.bias,
                                 gated, Y, c, alpha, grads, loss_scale=loss_scale, accumulate=accumulate,
                                 p_h=p_h, p_att=p_att, seed=seed, dx=dx_out)


def amil_stack_head(seq, classifier, x, training):
    """amil_stack followed by the classifier / hazard head as one autograd node -> (hazards, S, Y_hat, A_raw)."""
    from .. import ops
    lin, att = seq[0], seq[3]
    gated = isinstance(att, Attn_Net_Gated)
    Wa, ba, Wb, bb, Wc, bc = att.stack_params()
    p_h = seq[2].p if training else 0.0
    p_att = 0.25 if (training and att.att_dropout) else 0.0
    seed = ops.next_dropout_seed() if training else 0
    return ops.amil_head(x, lin.weight, lin.bias, Wa, ba, Wb, bb, Wc, bc, classifier.weight, classifier.bias,
                         gated, p_h, p_att, seed)
