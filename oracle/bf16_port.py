"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the attention-MIL stack with the rounding points of the bf16-storage
kernels (multimodalfusion_amd/csrc/mmf_bf16.h), forward AND a hand-derived backward
(SURVEY.md Appendix A.4), because autograd cannot express "round the saved activation,
not its gradient".

The reference has no bf16 mode (models/model_attention_mil_path.py:50-72 runs in fp32),
so this file is pinned in two steps:
  * with rounding switched off (`rnd=None`) the manual forward/backward must reproduce the
    golden fixtures generated from the imported reference (tests/test_oracle_bf16.py) --
    that pins the formulas;
  * with rounding on, it differs from those fixtures only by bf16 quantisation
    (same test, bf16-sized tolerance) -- that pins the rounding points as harmless.
Math in float64; `rb()` rounds to nearest-even bf16 exactly where the kernels do.
"""
from __future__ import annotations

import numpy as np
import torch


def rb(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (nearest even) and come back to float64."""
    return t.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def _t(v):
    return torch.as_tensor(np.asarray(v)).to(torch.float64)


def amil_bf16(sd, prefix, x, gated, dropout, masks=None, rnd=rb):
    """Forward of Sequential(Linear, ReLU, Dropout, Attn_Net*) + softmax pooling
    (models/model_attention_mil_path.py:20-29,52-56; models/model_modules.py:70-110).

    x: [N x L] (already bf16-representable when rnd is rb).  masks: scaled keep masks
    {"h","a","b"} or None (eval).  Returns (M [1 x H], A_raw [1 x N], saved) where `saved`
    is what backward needs."""
    r = rnd if rnd is not None else (lambda t: t)
    g = lambda k: _t(sd[k])
    W1, b1 = g(f"{prefix}.0.weight"), g(f"{prefix}.0.bias")
    att = f"{prefix}.3"
    if gated:
        Wa, ba = g(f"{att}.attention_a.0.weight"), g(f"{att}.attention_a.0.bias")
        Wb, bb = g(f"{att}.attention_b.0.weight"), g(f"{att}.attention_b.0.bias")
        Wc, bc = g(f"{att}.attention_c.weight"), g(f"{att}.attention_c.bias")
    else:
        last = 3 if dropout else 2
        Wa, ba = g(f"{att}.module.0.weight"), g(f"{att}.module.0.bias")
        Wb = bb = None
        Wc, bc = g(f"{att}.module.{last}.weight"), g(f"{att}.module.{last}.bias")
    x = _t(x)
    one = torch.ones((), dtype=torch.float64)
    m_h = _t(masks["h"]) if masks and masks.get("h") is not None else one
    m_a = _t(masks["a"]) if masks and dropout and masks.get("a") is not None else one
    m_b = _t(masks["b"]) if masks and dropout and gated and masks.get("b") is not None else one

    W1q, Waq = r(W1), r(Wa)
    Wbq = r(Wb) if gated else None
    h = r(torch.relu(x @ W1q.T + b1) * m_h)                  # saved bf16
    a = r(torch.tanh(h @ Waq.T + ba))                        # saved bf16; the scores use the same rounded values
    b = r(torch.sigmoid(h @ Wbq.T + bb)) if gated else None
    ab = (a * m_a) * (b * m_b) if gated else a * m_a
    s = ab @ Wc.T + bc                                       # [N x 1]
    A_raw = s.T
    p = torch.softmax(A_raw, dim=1)                          # [1 x N]
    M = p @ h
    saved = dict(x=x, h=h, a=a, b=b, p=p, M=M, Wc=Wc, Waq=Waq, Wbq=Wbq,
                 m_h=m_h, m_a=m_a, m_b=m_b, gated=gated, rnd=r)
    return M, A_raw, saved


def amil_bf16_backward(saved, dM, gA=None):
    """dM [1 x H], gA [1 x N] or None -> dict of parameter gradients (names as mmf_amil_grads)."""
    r = saved["rnd"]
    x, h, a, b, p, M = (saved[k] for k in ("x", "h", "a", "b", "p", "M"))
    Wc, Waq, Wbq, gated = saved["Wc"], saved["Waq"], saved["Wbq"], saved["gated"]
    m_h, m_a, m_b = saved["m_h"], saved["m_a"], saved["m_b"]
    dM = _t(dM).reshape(1, -1)
    gvec = h @ dM.T                                          # [N x 1]
    dmm = (dM * M).sum()
    ds = p.T * (gvec - dmm)
    if gA is not None:
        ds = ds + _t(gA).reshape(-1, 1)
    out = {"dbc": ds.sum().reshape(1)}
    wc = Wc.reshape(1, -1)
    if gated:
        a_d, b_d = a * m_a, b * m_b
        dPa = r(ds * wc * b_d * m_a * (1 - a * a))
        dPb = r(ds * wc * a_d * m_b * b * (1 - b))
        out["dWc"] = (ds * a_d * b_d).sum(0, keepdim=True)
        out["dWb"], out["dbb"] = dPb.T @ h, dPb.sum(0)
        dh = dPa @ Waq + dPb @ Wbq
    else:
        a_d = a * m_a
        dPa = r(ds * wc * m_a * (1 - a * a))
        out["dWc"] = (ds * a_d).sum(0, keepdim=True)
        dh = dPa @ Waq
    out["dWa"], out["dba"] = dPa.T @ h, dPa.sum(0)
    dh = dh + p.T * dM
    scale_h = m_h.max() if m_h.ndim else m_h                 # 1/(1-p) in train mode, 1 in eval
    du = r(torch.where(h > 0, dh * scale_h, torch.zeros_like(dh)))
    out["dW1"], out["db1"] = du.T @ x, du.sum(0)
    return out


def path_step_bf16(sd, x, y, c, alpha, gated=True, dropout=False, masks=None, rnd=rb):
    """One bag through the path head (models/model_attention_mil_path.py:50-72) + nll_surv
    (utils/loss_utils.py:22-39): the attention stack by the functions above, the tiny fp32 tail
    (classifier, hazards, loss) by autograd in float64."""
    from . import torch_port as tp
    M, A_raw, saved = amil_bf16(sd, "attention_net_WSI", x, gated, dropout, masks, rnd)
    Mleaf = M.detach().clone().requires_grad_(True)
    Wk = _t(sd["classifier.weight"]).requires_grad_(True)
    bk = _t(sd["classifier.bias"]).requires_grad_(True)
    logits = Mleaf @ Wk.T + bk
    hazards, S, Y_hat = tp.surv_head(logits)
    Y = torch.tensor([[y]], dtype=torch.int64)
    cc = torch.tensor([[float(c)]], dtype=torch.float64)
    loss = tp.nll_loss(hazards, S, Y, cc, alpha=alpha)
    loss.backward()
    grads = amil_bf16_backward(saved, Mleaf.grad)
    names = {"dW1": "attention_net_WSI.0.weight", "db1": "attention_net_WSI.0.bias"}
    att = "attention_net_WSI.3"
    if gated:
        names.update({"dWa": f"{att}.attention_a.0.weight", "dba": f"{att}.attention_a.0.bias",
                      "dWb": f"{att}.attention_b.0.weight", "dbb": f"{att}.attention_b.0.bias",
                      "dWc": f"{att}.attention_c.weight", "dbc": f"{att}.attention_c.bias"})
    else:
        last = 3 if dropout else 2
        names.update({"dWa": f"{att}.module.0.weight", "dba": f"{att}.module.0.bias",
                      "dWc": f"{att}.module.{last}.weight", "dbc": f"{att}.module.{last}.bias"})
    g = {names[k]: v.reshape(np.asarray(sd[names[k]]).shape).numpy() for k, v in grads.items()}
    g["classifier.weight"] = Wk.grad.numpy()
    g["classifier.bias"] = bk.grad.numpy()
    return dict(hazards=hazards.detach().numpy(), S=S.detach().numpy(), Y_hat=Y_hat.numpy(),
                A_raw=A_raw.numpy(), M=M.numpy(), loss=float(loss), grads=g)


def _amil_grad_names(prefix, gated, dropout):
    att = f"{prefix}.3"
    names = {"dW1": f"{prefix}.0.weight", "db1": f"{prefix}.0.bias"}
    if gated:
        names.update({"dWa": f"{att}.attention_a.0.weight", "dba": f"{att}.attention_a.0.bias",
                      "dWb": f"{att}.attention_b.0.weight", "dbb": f"{att}.attention_b.0.bias",
                      "dWc": f"{att}.attention_c.weight", "dbc": f"{att}.attention_c.bias"})
    else:
        last = 3 if dropout else 2
        names.update({"dWa": f"{att}.module.0.weight", "dba": f"{att}.module.0.bias",
                      "dWc": f"{att}.module.{last}.weight", "dbc": f"{att}.module.{last}.bias"})
    return names


def mm_step_bf16(sd_np, radio_xs, path_x, omic_x, y, c, alpha, fusion="concat", gate_path=True, gate_radio=True,
                 mode="radio_path_omic", rnd=rb):
    """The multimodal head (models/model_mm_attention_mil.py:128-200, eval mode) with the PATHOLOGY branch in bf16
    storage -- BASELINE config 5: the path bag and its saved activations are rounded where the bf16 kernels round them
    (amil_bf16 above), the radiology stack, the omic SNN, the fusion and the classifier are the fp32 path of the
    reference (oracle/torch_port.py, float64 here).  Gradients: autograd up to d(M_path), then the hand-derived
    backward of the rounded branch."""
    from . import torch_port as tp
    sd = tp.to_torch(sd_np, torch.float64)
    T = lambda a: torch.as_tensor(np.asarray(a)).to(torch.float64)
    M, A_raw_p, saved = amil_bf16(sd_np, "attention_net_WSI", path_x, gate_path, False, None, rnd)
    Mleaf = M.detach().clone().requires_grad_(True)
    hz, S, Yh, A_raw, MM = tp.mm_forward(sd, [T(x) for x in radio_xs], None, T(omic_x), fusion=fusion,
                                         gate_path=gate_path, gate_radio=gate_radio, dropout=False, mode=mode,
                                         path_override=(Mleaf, A_raw_p))
    loss = tp.nll_loss(hz, S, torch.tensor([int(y)]), torch.tensor([float(c)]), alpha=alpha)
    names = [k for k in sd if not k.startswith("attention_net_WSI.")]
    gs = torch.autograd.grad(loss, [sd[k] for k in names] + [Mleaf], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(sd[k])).numpy() for k, g in zip(names, gs[:-1])}
    pg = amil_bf16_backward(saved, gs[-1])
    for k, name in _amil_grad_names("attention_net_WSI", gate_path, False).items():
        grads[name] = pg[k].reshape(np.asarray(sd_np[name]).shape).numpy()
    return dict(hazards=hz.detach().numpy(), S=S.detach().numpy(), Y_hat=Yh.numpy(), loss=float(loss.detach()),
                A_raw={k: v.detach().numpy() for k, v in A_raw.items()}, grads=grads)
