"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Torch-CPU restatement of the reference's stage-2 path (embedding-level fusion models and their batched losses),
functional over a plain state-dict, gradients by autograd.  Cited lines are in /root/reference.
Dropout is an explicit, already scaled mask per site (None = eval).  BatchNorm1d uses torch's own functional form
(training: batch statistics, running buffers updated in place on the copies held in `sd`).
Pinned by tests/golden/stage2.npz (generated from the imported reference by oracle/gen_golden_stage2.py).
"""
from __future__ import annotations

from itertools import combinations

import numpy as np
import torch
import torch.nn.functional as F

from . import inputs as gen
from . import torch_port as tp

CASES = [
    # name, family, kind, train_type, mode, B, K, n_layers, train, loss
    ("uni_nll_fcnn_train", "nll", "uni", "fcnn", "path", 16, 4, 1, True, ("nll", 0.15)),
    ("uni_nll_highway_train", "nll", "uni", "highway", "radio", 32, 4, 2, True, ("rank_nll", "sigmoid", "mean", 0.15, 0.5)),
    ("uni_nll_highway_eval", "nll", "uni", "highway", "omic", 8, 8, 1, False, ("nll", 0.0)),
    ("uni_cox_fcnn_train", "cox", "uni", "fcnn", "radio", 32, 4, 1, True, ("cox",)),
    ("uni_cox_highway_train", "cox", "uni", "highway", "path", 24, 4, 1, True, ("rank", "sigmoid", "mean")),
    ("uni_cox_residual_train", "cox", "uni", "residual", "omic", 16, 4, 2, True, ("rank", "relu", "sum")),
    ("mm_nll_early_fcnn_train", "nll", "mm", "early-fcnn", "radio_path_omic", 32, 4, 1, True, ("nll", 0.15)),
    ("mm_nll_late_fcnn_train", "nll", "mm", "late-fcnn", "radio_path_omic", 32, 4, 1, True, ("rank_nll", "sigmoid", "mean", 0.15, 0.5)),
    ("mm_nll_early_highway_train", "nll", "mm", "early-highway", "path_omic", 16, 4, 1, True, ("nll", 0.4)),
    ("mm_nll_late_highway_train", "nll", "mm", "late-highway", "radio_path", 16, 4, 1, True, ("nll", 0.15)),
    ("mm_nll_kronecker_train", "nll", "mm", "kronecker", "radio_path_omic", 32, 4, 1, True, ("nll", 0.15)),
    ("mm_nll_kronecker_eval", "nll", "mm", "kronecker", "radio_omic", 5, 4, 1, False, ("nll", 0.15)),
    ("mm_cox_late_fcnn_train", "cox", "mm", "late-fcnn", "radio_path_omic", 32, 4, 1, True, ("cox",)),
    ("mm_cox_early_highway_train", "cox", "mm", "early-highway", "radio_path_omic", 16, 4, 1, True, ("rank", "sigmoid", "mean")),
    ("mm_cox_kronecker_train", "cox", "mm", "kronecker", "radio_path", 32, 4, 1, True, ("cox",)),
]


def case_meta():
    meta = {}
    for i, (name, family, kind, tt, mode, B, K, nl, train, loss) in enumerate(CASES):
        meta[name] = dict(family=family, kind=kind, train_type=tt, mode=mode, B=B, K=K, n_layers=nl, train=train,
                          loss=list(loss), seed=700 + i, x_seed=7700 + i, mask_seed=8800 + 3 * i)
    return meta


# ---- deterministic inputs (shared by the generator and the tests) -----------------------------------------------
def state_dict_for(shapes: dict, seed: int):
    """Weights for a stage-2 model from the repo generator, keyed and shaped like `shapes` (name -> shape, in
    state_dict order): Linear weights N(0, 1/sqrt(fan_in)), biases N(0, 0.05), BatchNorm weight 1 + N(0, 0.1),
    bias N(0, 0.1), running_mean N(0, 0.1), running_var in [0.5, 1.5]."""
    sd = {}
    for i, (k, shp) in enumerate(shapes.items()):
        shp = tuple(shp)
        if k.endswith("num_batches_tracked"):
            sd[k] = np.zeros(shp, dtype=np.int64)
        elif k.endswith("running_var"):
            sd[k] = (0.5 + gen.uniform01(seed, int(np.prod(shp)), stream=i).reshape(shp)).astype(np.float32)
        elif k.endswith("running_mean"):
            sd[k] = gen.normal(seed, shp, stream=i, std=0.1)
        elif len(shp) == 2:
            sd[k] = gen.normal(seed, shp, stream=i, std=1.0 / np.sqrt(shp[1]))
        elif "bn" in k.split(".")[-2] or _is_bn_key(k, shapes):
            sd[k] = (1.0 if k.endswith("weight") else 0.0) + gen.normal(seed, shp, stream=i, std=0.1)
        else:
            sd[k] = gen.normal(seed, shp, stream=i, std=0.05)
    return sd


def _is_bn_key(k, shapes):
    return (k.rsplit(".", 1)[0] + ".running_mean") in shapes


def batch_for(m):
    B, K = m["B"], m["K"]
    hs = [gen.normal(m["x_seed"], (B, 256), stream=s) for s in range(3)]
    u = gen.uniform01(m["x_seed"], 3 * B, stream=9)
    Y = np.minimum((u[:B] * K).astype(np.int64), K - 1)
    c = (u[B:2 * B] < 0.4).astype(np.float32)
    t = np.round(u[2 * B:] * 40.0) / 2.0            # event times with ties
    return hs, Y, c, t


# ---- building blocks ----------------------------------------------------------------------------------------------
def _bn(sd, prefix, x, train):
    """nn.BatchNorm1d: models/model_modules.py:13-14; torch defaults eps 1e-5, momentum 0.1."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=train, momentum=0.1, eps=1e-5)


def _mask(x, m):
    return x if m is None else x * m


def highway(sd, prefix, x, n_layers, train, mask):
    """models/model_modules.py:17-27."""
    x = _mask(_bn(sd, prefix + ".bn1", x, train), mask)
    for layer in range(n_layers):
        gate = torch.sigmoid(tp._lin(sd, f"{prefix}.gate.{layer}", x))
        nonlinear = torch.relu(tp._lin(sd, f"{prefix}.nonlinear.{layer}", x))
        linear = tp._lin(sd, f"{prefix}.linear.{layer}", x)
        x = gate * nonlinear + (1 - gate) * linear
    return _bn(sd, prefix + ".bn2", x, train)


def residual(sd, prefix, x, n_layers, train):
    """models/model_modules.py:29-58."""
    for i in range(n_layers):
        p = f"{prefix}.blocks.{i}"
        out = torch.relu(_bn(sd, p + ".bn1", tp._lin(sd, p + ".fc1", x), train))
        out = _bn(sd, p + ".bn2", tp._lin(sd, p + ".fc2", out), train)
        x = torch.relu(out + x)
    return x


def fcnn(sd, prefix, x, train, mask, last):
    """Sequential(Linear, BatchNorm1d, ReLU, Dropout(0.7)[, Linear]): models/nll_models_pretrained.py:82-90."""
    h = _mask(torch.relu(_bn(sd, prefix + ".1", tp._lin(sd, prefix + ".0", x), train)), mask)
    return tp._lin(sd, prefix + ".4", h) if last else h


def pick(mode, r, p, o):
    """cat / v_list order of models/nll_models_pretrained.py:153-160,164-171,180-187."""
    R, P, O = "radio" in mode, "path" in mode, "omic" in mode
    if R and P and not O:
        return [r, p]
    if R and O and not P:
        return [r, o]
    if O and P and not R:
        return [o, p]
    return [r, p, o]


def heads(logits):
    """models/nll_models_pretrained.py:58-62."""
    hazards = torch.sigmoid(logits)
    S = torch.cumprod(1 - hazards, dim=1)
    return -torch.sum(S, dim=1), hazards, S


def forward(sd, m, hs, masks):
    """(risk, hazards, S) of the case's model.  masks: dict site-name -> scaled mask (see masks_for)."""
    hr, hp, ho = hs
    train, tt, fam, nl = m["train"], m["train_type"], m["family"], m["n_layers"]
    g = (lambda k: masks.get(k)) if masks else (lambda k: None)
    if m["kind"] == "uni":
        h = {"path": hp, "radio": hr, "omic": ho}[m["mode"]]
        if fam == "nll":
            if tt == "fcnn":
                logits = _mask(tp._lin(sd, "classifier.0", h), g("d0"))
            else:
                logits = tp._lin(sd, "classifier", highway(sd, "highway", h, nl, train, g("d0")))
            return heads(logits)
        if tt == "fcnn":
            risk = fcnn(sd, "classifier", h, train, g("d0"), True)
        elif tt == "highway":
            risk = tp._lin(sd, "classifier", highway(sd, "highway", h, nl, train, g("d0")))
        else:
            risk = tp._lin(sd, "classifier", residual(sd, "residual", h, nl, train))
        return risk.squeeze(), None, None
    # multimodal
    if "late" in tt:
        if tt == "late-fcnn":
            last = fam == "cox"
            r = fcnn(sd, "layer_MRI", hr, train, g("d0"), last)
            p = fcnn(sd, "layer_WSI", hp, train, g("d1"), last)
            o = fcnn(sd, "layer_omic", ho, train, g("d2"), last)
        else:
            r = highway(sd, "highway_radio", hr, nl, train, g("h0"))
            p = highway(sd, "highway_path", hp, nl, train, g("h1"))
            o = highway(sd, "highway_omic", ho, nl, train, g("h2"))
        mm = torch.cat(pick(m["mode"], r, p, o), dim=1)
        cls = "classifier.0" if tt == "late-fcnn" else "classifier"
        out = tp._lin(sd, cls, mm)
        return (out.squeeze(), None, None) if fam == "cox" else heads(out)
    if "early" in tt:
        mm = torch.cat(pick(m["mode"], hr, hp, ho), dim=1)
        if tt == "early-fcnn":
            out = fcnn(sd, "classifier", mm, train, g("d0"), True)
        else:
            out = tp._lin(sd, "classifier", highway(sd, "highway", mm, nl, train, g("d0")))
        return (out, None, None) if fam == "cox" else heads(out)
    xm = {k: masks.get(k) for k in ("o0", "o1", "o2", "post", "enc1", "enc2")} if masks else None
    out = tp._lin(sd, "classifier", tp.xfusion(sd, "xfusion", pick(m["mode"], hr, hp, ho), xm))
    return (out, None, None) if fam == "cox" else heads(out)


def masks_for(m, dtype=np.float64):
    """Scaled keep masks, by dropout site, as the device draws them for mask_seed (p = 0.7 everywhere in stage 2).
    Sites: d0/d1/d2 = sites 0/1/2 of mask_seed; h0/h1/h2 = site 0 of mask_seed + 0/1/2 (late-highway: one Highway per
    modality); o*/post/enc* = XlinearFusion sites i / 8 / 9 / 10."""
    if not m["train"]:
        return None
    B, K, s, tt = m["B"], m["K"], m["mask_seed"], m["train_type"]
    nmod = sum(k in m["mode"] for k in ("radio", "path", "omic"))
    mk = lambda seed, site, cols: gen.drop_scale_mask(seed, site, B, cols, 0.7, dtype)
    if m["kind"] == "uni":
        if tt == "fcnn":
            return {"d0": mk(s, 0, K if m["family"] == "nll" else 128)}
        return {"d0": mk(s, 0, 256)} if tt == "highway" else {}
    if tt == "late-fcnn":
        return {f"d{i}": mk(s, i, 128) for i in range(3)}
    if tt == "late-highway":
        return {f"h{i}": mk(s + i, 0, 256) for i in range(3)}
    if tt == "early-fcnn":
        return {"d0": mk(s, 0, 128)}
    if tt == "early-highway":
        return {"d0": mk(s, 0, 256 * nmod)}
    out = {f"o{i}": mk(s, i, 16) for i in range(nmod)}
    out.update(post=mk(s, 8, 17 ** nmod), enc1=mk(s, 9, 256), enc2=mk(s, 10, 256))
    return out


# ---- losses -------------------------------------------------------------------------------------------------------
def ranking_loss(risks, times, c, phi, reduction):
    """utils/loss_utils.py:58-101, the same pair rule, same order."""
    B = len(times)
    events = 1 - c
    more, less = [], []
    for a, b in combinations(range(B), 2):
        if times[a] < times[b] and events[a]:
            more.append(risks[a]); less.append(risks[b])
        elif times[b] < times[a] and events[b]:
            more.append(risks[b]); less.append(risks[a])
    if not less:
        return torch.zeros(1, dtype=risks.dtype, requires_grad=True)
    r = torch.stack(more) - torch.stack(less)
    v = torch.sigmoid(r) if phi == "sigmoid" else torch.relu(r)
    return -v.mean() if reduction == "mean" else -v.sum()


def loss_of(m, risk, hazards, S, Y, c, t):
    spec = m["loss"]
    Yt, ct = torch.as_tensor(Y), torch.as_tensor(c).to(risk.dtype)
    if spec[0] == "nll":
        return tp.nll_loss(hazards, S, Yt, ct, alpha=spec[1])
    if spec[0] == "cox":
        return tp.cox_loss(risk, torch.as_tensor(t), ct)
    if spec[0] == "rank":
        return ranking_loss(risk.reshape(-1), torch.as_tensor(t), ct, spec[1], spec[2])
    rk = ranking_loss(risk.reshape(-1), Yt, ct, spec[1], spec[2])             # RankingNLLSurvLoss ranks over the labels
    return rk + tp.nll_loss(hazards, S, Yt, ct, alpha=spec[3]) * spec[4]


def run_case(m, shapes, dtype=torch.float64):
    sd_np = state_dict_for(shapes, m["seed"])
    sd = {k: torch.as_tensor(v).to(dtype if v.dtype != np.int64 else torch.int64).clone() for k, v in sd_np.items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    hs, Y, c, t = batch_for(m)
    mk = masks_for(m)
    masks = {k: torch.as_tensor(v).to(dtype) for k, v in mk.items()} if mk else None
    risk, hazards, S = forward(sd, m, [torch.as_tensor(h).to(dtype) for h in hs], masks)
    loss = loss_of(m, risk, hazards, S, Y, c, t)
    params = {k: v for k, v in sd.items() if v.requires_grad}
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    out = dict(risk=risk.detach().numpy(), loss=float(loss.detach().reshape(-1)[0]),
               grads={k: (g.detach().numpy() if g is not None else np.zeros(tuple(params[k].shape))) for k, g in zip(params, grads)},
               buffers={k: v.detach().numpy() for k, v in sd.items() if "running" in k})
    if hazards is not None:
        out.update(hazards=hazards.detach().numpy(), S=S.detach().numpy())
    return out
