"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Torch-CPU restatement ("port") of the reference hot path, functional style over
a plain state-dict, same ATen op sequence as the reference so that it is both
the parity oracle (run in fp64 or fp32; gradients by autograd) and the
`cpu_baseline` timed by bench.py (kind = "port").

Each function cites the reference lines it restates (paths relative to
/root/reference).  Dropout is expressed through explicit, already-scaled masks
(mask == keep/(1-p)); `None` means eval mode.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def to_torch(sd, dtype=torch.float64, requires_grad=True):
    out = {}
    for k, v in sd.items():
        t = torch.as_tensor(np.asarray(v)).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _mul_mask(t, masks, key):
    if isinstance(masks, str):          # "torch": draw the mask with torch's own CPU RNG, as nn.Dropout does
        return F.dropout(t, 0.25, True) # (used only when timing the cpu_baseline, never for parity)
    if masks is not None and masks.get(key) is not None:
        return t * masks[key]
    return t


# -- models/model_modules.py:70-85 (Attn_Net) and :87-110 (Attn_Net_Gated) ------------
def attn_net(sd, prefix, h, gated, dropout, masks=None):
    if gated:
        a = torch.tanh(_lin(sd, f"{prefix}.attention_a.0", h))
        b = torch.sigmoid(_lin(sd, f"{prefix}.attention_b.0", h))
        if dropout:
            a = _mul_mask(a, masks, "a")
            b = _mul_mask(b, masks, "b")
        A = _lin(sd, f"{prefix}.attention_c", a.mul(b))
    else:
        a = torch.tanh(_lin(sd, f"{prefix}.module.0", h))
        last = 2
        if dropout:
            a = _mul_mask(a, masks, "a")
            last = 3
        A = _lin(sd, f"{prefix}.module.{last}", a)
    return A, h


# -- Sequential(Linear, ReLU, Dropout(0.25), Attn_Net*) + softmax pooling -------------
# models/model_attention_mil_path.py:20-29,52-56
def amil_pool(sd, prefix, x, gated, dropout, masks=None):
    h = torch.relu(_lin(sd, f"{prefix}.0", x))
    h = _mul_mask(h, masks, "h")              # the Dropout(0.25) that is always present
    A, h = attn_net(sd, f"{prefix}.3", h, gated, dropout, masks)
    A = torch.transpose(A, 1, 0)
    A_raw = A
    A = F.softmax(A, dim=1)
    M = torch.mm(A, h)
    return M, A_raw


# -- classifier + hazards (model_attention_mil_path.py:58-61) -------------------------
def surv_head(logits):
    Y_hat = torch.topk(logits, 1, dim=1)[1]
    hazards = torch.sigmoid(logits)
    S = torch.cumprod(1 - hazards, dim=1)
    return hazards, S, Y_hat


def path_forward(sd, x, gated=True, dropout=False, masks=None):
    """models/model_attention_mil_path.py:50-72 -> (hazards, S, Y_hat, A_raw, M)."""
    M, A_raw = amil_pool(sd, "attention_net_WSI", x, gated, dropout, masks)
    logits = _lin(sd, "classifier", M)
    hazards, S, Y_hat = surv_head(logits)
    return hazards, S, Y_hat, A_raw, M


def radio_forward(sd, xs, gated=True, dropout=True, masks=None):
    """models/model_attention_mil_radio.py:73-115; xs = list of [n x 1024] in modality order."""
    if len(xs) > 1:
        h = torch.cat(list(xs), dim=1)
        h = _lin(sd, "reduce_dim", h)
    else:
        h = xs[0]
    M, A_raw = amil_pool(sd, "attention_net_radio", h, gated, dropout, masks)
    logits = _lin(sd, "classifier", M)
    hazards, S, Y_hat = surv_head(logits)
    return hazards, S, Y_hat, A_raw, M


# -- models/model_modules.py:64-68 (SNN_Block) ---------------------------------------
_SELU_ALPHA = 1.6732632423543772848170429916717
_SELU_SCALE = 1.0507009873554804934193349852946


def alpha_dropout_apply(x, keep, p=0.25):
    """torch AlphaDropout (train): y = a*(x*keep + alpha'*(1-keep)) + b (model_modules.py:68)."""
    alpha_p = -_SELU_ALPHA * _SELU_SCALE
    a = 1.0 / math.sqrt((alpha_p * alpha_p * p + 1.0) * (1.0 - p))
    b = -a * alpha_p * p
    return a * (x * keep + alpha_p * (1.0 - keep)) + b


def snn_stack(sd, prefix, x, n_blocks=2, keeps=None, p=0.25):
    f = x
    for i in range(n_blocks):
        f = F.selu(_lin(sd, f"{prefix}.{i}.0", f))
        if keeps is not None and keeps[i] is not None:
            f = alpha_dropout_apply(f, keeps[i], p)
    return f


def maxnet_forward(sd, x, nll=True, keeps=None):
    """models/model_genomic.py:53-72 (incl. the unsqueeze(0) / cumprod(dim=1) quirk)."""
    feats = snn_stack(sd, "fc_omic", x, 2, keeps)
    if nll:
        logits = _lin(sd, "classifier", feats).unsqueeze(0)
        Y_hat = torch.topk(logits, 1, dim=1)[1]
        hazards = torch.sigmoid(logits)
        S = torch.cumprod(1 - hazards, dim=1)
        return hazards, S, Y_hat, feats
    risk = _lin(sd, "classifier", feats).squeeze()
    return risk, None, None, feats


# -- models/model_modules.py:156-178 (XlinearFusion.forward, gate=1, skip=1) ----------
def xfusion(sd, prefix, v_list, masks=None):
    v_cat = torch.cat(v_list, dim=1)
    o_list = []
    for i, v in enumerate(v_list):
        h = torch.relu(_lin(sd, f"{prefix}.reduce.{i}.0.0", v))
        z = _lin(sd, f"{prefix}.reduce.{i}.1.0", v_cat)
        o = torch.relu(_lin(sd, f"{prefix}.reduce.{i}.2.0", torch.sigmoid(z) * h))
        o = _mul_mask(o, masks, f"o{i}")
        o = torch.cat((o, torch.ones(o.shape[0], 1, dtype=o.dtype)), 1)
        o_list.append(o)
    o_fusion = o_list[0]
    for o in o_list[1:]:
        o_fusion = torch.bmm(o_fusion.unsqueeze(2), o.unsqueeze(1)).flatten(start_dim=1)
    out = _mul_mask(o_fusion, masks, "post")
    out = torch.relu(_lin(sd, f"{prefix}.encoder1.0", out))
    out = _mul_mask(out, masks, "enc1")
    for v in v_list:                       # skip = 1
        out = torch.cat((out, v), dim=1)
    out = torch.relu(_lin(sd, f"{prefix}.encoder2.0", out))
    out = _mul_mask(out, masks, "enc2")
    return out


def mm_forward(sd, radio_xs, path_x, omic_x, fusion="concat", gate_path=True, gate_radio=True,
               dropout=False, mode="radio_path_omic", masks=None, path_override=None):
    """models/model_mm_attention_mil.py:128-200 (radio_fusion='concat').

    omic_x is 1-D [G] (the forward does X.unsqueeze(0), :165).  masks is a dict of
    dicts: {'radio':{h,a,b}, 'path':{h,a,b}, 'omic_keeps':[k0,k1], 'mm':{...}, 'cls': mask}.
    path_override = (M_path, A_raw_path): the pathology branch computed elsewhere (oracle/bf16_port.py supplies the
    bf16-rounded branch as a leaf so that its hand-derived backward can take over from d(M_path)).
    """
    masks = masks or {}
    A_raw = {}
    vs = {}
    if "radio" in mode:
        h = torch.cat(list(radio_xs), dim=1) if len(radio_xs) > 1 else radio_xs[0]
        if len(radio_xs) > 1:
            h = _lin(sd, "reduce_dim", h)
        M_radio, A = amil_pool(sd, "attention_net_radio", h, gate_radio, dropout, masks.get("radio"))
        A_raw["radiology"] = A
        vs["radio"] = M_radio
    if "path" in mode:
        if path_override is not None:
            M_path, A = path_override
        else:
            M_path, A = amil_pool(sd, "attention_net_WSI", path_x, gate_path, dropout, masks.get("path"))
        A_raw["pathology"] = A
        vs["path"] = M_path
    if "omic" in mode:
        vs["omic"] = snn_stack(sd, "fc_omic", omic_x.unsqueeze(0), 2, masks.get("omic_keeps"))
    # list order per :168-187
    has = lambda k: k in mode
    if has("radio") and has("path") and not has("omic"):
        order = ["radio", "path"]
    elif has("radio") and has("omic") and not has("path"):
        order = ["radio", "omic"]
    elif has("omic") and has("path") and not has("radio"):
        order = ["omic", "path"]
    else:
        order = ["radio", "path", "omic"]
    v_list = [vs[k] for k in order]
    if fusion == "tensor":
        MM = xfusion(sd, "mm", v_list, masks.get("mm"))
        c = torch.relu(_lin(sd, "classifier.0", MM))
        c = _mul_mask(c, masks, "cls")
        logits = _lin(sd, "classifier.3", c)
    else:
        MM = torch.cat(v_list, dim=1)
        logits = _lin(sd, "classifier", MM)
    hazards, S, Y_hat = surv_head(logits)
    return hazards, S, Y_hat, A_raw, MM


# -- utils/loss_utils.py:22-39 --------------------------------------------------------
def nll_loss(hazards, S, Y, c, alpha=0.4, eps=1e-7):
    bsz = len(Y)
    Y = Y.view(bsz, 1)
    c = c.view(bsz, 1).to(hazards.dtype)
    if S is None:
        S = torch.cumprod(1 - hazards, dim=1)
    S_pad = torch.cat([torch.ones_like(c), S], 1)
    unc = -(1 - c) * (torch.log(torch.gather(S_pad, 1, Y).clamp(min=eps))
                      + torch.log(torch.gather(hazards, 1, Y).clamp(min=eps)))
    cen = -c * torch.log(torch.gather(S_pad, 1, Y + 1).clamp(min=eps))
    neg_l = cen + unc
    loss = (1 - alpha) * neg_l + alpha * unc
    return loss.mean()


# -- utils/loss_utils.py:124-139 (vectorised risk-set matrix; same values as the loop) --
def cox_loss(risks, times, c):
    t = torch.as_tensor(np.asarray(times))
    R = (t.view(1, -1) >= t.view(-1, 1)).to(risks.dtype)   # R[i,j] = times[j] >= times[i]
    theta = risks.reshape(-1)
    exp_theta = torch.exp(theta)
    return -torch.mean((theta - torch.log(torch.sum(exp_theta * R, dim=1))) * (1 - c.to(risks.dtype)))


# -- utils/utils.py:249-257 -----------------------------------------------------------
def l1_reg_all(sd):
    tot = None
    for w in sd.values():
        s = torch.abs(w).sum()
        tot = s if tot is None else tot + s
    return tot


def grads_of(loss, sd):
    names = [k for k, v in sd.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(sd[k])) for k, g in zip(names, gs)}


def path_step(sd_np, x_np, y, c, alpha, gated=True, dropout=False, masks_np=None,
              dtype=torch.float64):
    """One forward + nll_surv + backward of the path head; returns plain numpy results."""
    sd = to_torch(sd_np, dtype)
    x = torch.as_tensor(x_np).to(dtype)
    masks = None
    if masks_np is not None:
        masks = {k: torch.as_tensor(v).to(dtype) for k, v in masks_np.items()}
    hz, S, Yh, A_raw, M = path_forward(sd, x, gated, dropout, masks)
    loss = nll_loss(hz, S, torch.tensor([int(y)]), torch.tensor([float(c)]), alpha=alpha)
    g = grads_of(loss, sd)
    out = dict(hazards=hz, S=S, Y_hat=Yh, A_raw=A_raw, M=M, loss=loss)
    out = {k: v.detach().numpy() for k, v in out.items()}
    out["grads"] = {k: v.detach().numpy() for k, v in g.items()}
    return out
