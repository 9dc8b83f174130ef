"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Rebuilds the inputs of a golden case from its `meta` record and runs the torch-CPU
port on them.  Shared by the CPU tests (oracle vs fixtures) and the GPU parity
tests (HIP path vs oracle on the same inputs).
"""
from __future__ import annotations

import numpy as np
import torch

from . import inputs as gen
from . import torch_port as tp

MODS = ["T1", "T2", "T1Gd", "FLAIR"]


def _np(d):
    return {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in d.items()}


def amil_masks(seed, N, H, D, gated, dropout, dtype=np.float64):
    """Train-mode masks of one AMIL stack: site 0 = h, 1 = a, 2 = b (scaled by 1/(1-p))."""
    m = {"h": gen.drop_scale_mask(seed, 0, N, H, 0.25, dtype)}
    if dropout:
        m["a"] = gen.drop_scale_mask(seed, 1, N, D, 0.25, dtype)
        if gated:
            m["b"] = gen.drop_scale_mask(seed, 2, N, D, 0.25, dtype)
    return m


def shape_bag(x, kind):
    """Other input distributions than the generator's N(0, 1), derived from it deterministically (fp32 in, fp32 out):
      relu       max(x, 0): non-negative, half the entries zero -- what pooled post-ReLU ResNet features look like;
      lognormal  10 ** clip(2.5 x - 1, -6, 4): magnitudes from 1e-6 to 1e4 in one bag, heavy right tail;
      edge       N(0, 1) with a sprinkle of fp32 denormals (1e-40), exact zeros, and -- in feature column 0 only, whose
                 first-layer weights the test zeroes -- values beyond the largest bf16 (3.39e38 .. FLT_MAX)."""
    if not kind:
        return x
    x = np.asarray(x, np.float32)
    if kind == "relu":
        return np.maximum(x, np.float32(0))
    if kind == "lognormal":
        return np.power(np.float32(10), np.clip(np.float32(2.5) * x - np.float32(1), -6, 4)).astype(np.float32)
    if kind == "edge":
        y = x.copy()
        flat = y.reshape(-1)
        flat[::97] = np.float32(1e-40)
        flat[5::389] = np.float32(-2e-41)
        flat[11::211] = np.float32(0)
        y[:, 0] = np.where(np.arange(y.shape[0]) % 2 == 0, np.float32(3.4028234e38), np.float32(-3.39e38))
        return y
    raise ValueError(kind)


def path_inputs(m):
    sd = gen.path_state_dict(seed=m["seed"], gated=m["gated"], size=m["size"], n_classes=m["K"],
                             dropout=m["dropout"], bias_std=m["bias_std"])
    x = shape_bag(gen.bag(m["x_seed"], m["N"]), m.get("x_kind"))
    masks = None
    if m["train"]:
        H, D = gen.SIZE_DICT[m["size"]][1:]
        masks = amil_masks(m["mask_seed"], m["N"], H, D, m["gated"], m["dropout"])
    return sd, x, masks


def run_path(m, dtype=torch.float64):
    sd, x, masks = path_inputs(m)
    return tp.path_step(sd, x, m["y"], m["c"], m["alpha"], gated=m["gated"], dropout=m["dropout"],
                        masks_np=masks, dtype=dtype)


def radio_inputs(m):
    sd = gen.radio_state_dict(seed=m["seed"], gated=m["gated"], n_classes=m["K"], dropout=m["dropout"],
                              n_mod=m["n_mod"], bias_std=m["bias_std"])
    xs = [gen.bag(m["x_seed"], m["n"], stream=7 * i) for i in range(m["n_mod"])]
    masks = amil_masks(m["mask_seed"], m["n"], 256, 256, m["gated"], m["dropout"]) if m["train"] else None
    return sd, xs, masks


def run_radio(m, dtype=torch.float64):
    sd_np, xs, masks = radio_inputs(m)
    sd = tp.to_torch(sd_np, dtype)
    tm = {k: torch.as_tensor(v).to(dtype) for k, v in masks.items()} if masks else None
    hz, S, Yh, A_raw, M = tp.radio_forward(sd, [torch.as_tensor(x).to(dtype) for x in xs],
                                           m["gated"], m["dropout"], tm)
    loss = tp.nll_loss(hz, S, torch.tensor([m["y"]]), torch.tensor([float(m["c"])]), alpha=m["alpha"])
    g = tp.grads_of(loss, sd)
    out = _np(dict(hazards=hz, S=S, Y_hat=Yh, A_raw=A_raw, M=M, loss=loss))
    out["grads"] = _np(g)
    return out


def omic_inputs(m):
    from .gen_golden import omic_batch
    sd = gen.maxnet_state_dict(seed=m["seed"], input_dim=m["G"], nll=m["nll"], n_classes=m["K"],
                               bias_std=m["bias_std"])
    x, t, c = omic_batch(m["x_seed"], m["B"], m["G"])
    keeps = None
    if m["train"]:
        keeps = [gen.keep_mask(m["mask_seed"], i, m["B"], 256, 0.25).astype(np.float64) for i in range(2)]
    return sd, x, t, c, keeps


def run_omic(m, dtype=torch.float64):
    sd_np, x, t, c, keeps = omic_inputs(m)
    sd = tp.to_torch(sd_np, dtype)
    tk = [torch.as_tensor(k).to(dtype) for k in keeps] if keeps else None
    xt = torch.as_tensor(x).to(dtype)
    if m["nll"]:
        hz, S, Yh, feats = tp.maxnet_forward(sd, xt, True, tk)
        loss = tp.nll_loss(hz[0], S[0], torch.tensor([m["y"]]), torch.tensor([float(c[0])]), alpha=m["alpha"])
        out = _np(dict(hazards=hz, S=S, Y_hat=Yh, M=feats, loss=loss))
    else:
        risk, _, _, feats = tp.maxnet_forward(sd, xt, False, tk)
        loss = tp.cox_loss(risk, t, torch.as_tensor(c).to(dtype))
        out = _np(dict(hazards=risk.reshape(-1), M=feats, loss=loss))
    out["grads"] = _np(tp.grads_of(loss, sd))
    return out


def mm_inputs(m):
    sd = gen.mm_state_dict(seed=m["seed"], input_dim=m["G"], fusion=m["fusion"], gate_path=m["gate_path"],
                           gate_radio=m["gate_radio"], dropout=False, n_classes=m["K"], mode=m["mode"],
                           n_mod=4, bias_std=m["bias_std"])
    xs = [gen.bag(m["x_seed"], max(m["nr"], 1), stream=7 * i) for i in range(4)]
    xp = gen.bag(m["x_seed"], max(m["Np"], 1), stream=100)
    xo = gen.normal(m["x_seed"], (m["G"],), stream=200)
    return sd, xs, xp, xo


def run_mm(m, dtype=torch.float64):
    sd_np, xs, xp, xo = mm_inputs(m)
    sd = tp.to_torch(sd_np, dtype)
    T = lambda a: torch.as_tensor(a).to(dtype)
    hz, S, Yh, A_raw, MM = tp.mm_forward(sd, [T(x) for x in xs], T(xp), T(xo), fusion=m["fusion"],
                                         gate_path=m["gate_path"], gate_radio=m["gate_radio"],
                                         dropout=False, mode=m["mode"])
    loss = tp.nll_loss(hz, S, torch.tensor([m["y"]]), torch.tensor([float(m["c"])]), alpha=m["alpha"])
    g = tp.grads_of(loss, sd)
    out = _np(dict(hazards=hz, S=S, Y_hat=Yh, loss=loss))
    out["A_raw"] = _np(A_raw)
    out["grads"] = _np(g)
    return out


def run_trajectory(meta, dtype=torch.float64):
    """Re-creation of utils/core_utils.py:200-247 + Adam (utils/utils.py:144-146), dropout disabled."""
    K, gc, lam = meta["K"], meta["gc"], meta["lambda_reg"]
    sd_np = gen.path_state_dict(seed=meta["seed"], gated=True, size="small", n_classes=K, bias_std=0.05)
    sd = tp.to_torch(sd_np, dtype)
    params = list(sd.values())
    opt = torch.optim.Adam(params, lr=meta["lr"], weight_decay=meta["reg"])
    losses, risks, steps = [], [], []
    for bi, b in enumerate(meta["bags"]):
        x = torch.as_tensor(gen.bag(b["x_seed"], b["n"])).to(dtype)
        hz, S, Yh, A_raw, M = tp.path_forward(sd, x, True, False, None)
        risks.append(float(-S.detach().sum()))
        loss = tp.nll_loss(hz, S, torch.tensor([b["y"]]), torch.tensor([float(b["c"])]), alpha=meta["alpha"])
        losses.append(float(loss.detach()))
        loss = loss / gc + tp.l1_reg_all(sd) * lam
        loss.backward()
        if (bi + 1) % gc == 0:
            opt.step()
            opt.zero_grad()
            steps.append({k: v.detach().numpy().copy() for k, v in sd.items()})
    return dict(losses=np.array(losses), risks=np.array(risks), steps=steps)
