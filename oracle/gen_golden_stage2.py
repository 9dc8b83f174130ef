"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/stage2.npz by importing the reference's stage-2 modules
(models/nll_models_pretrained.py, models/coxranking_models_pretrained.py, utils/loss_utils.py) on CPU:

    python -m oracle.gen_golden_stage2          # build container only; the reference never travels

Shims (recorded in the fixture meta): the empty torchvision stub; `torch.cuda.FloatTensor` -> CPU constructor for the
Kronecker fusion (models/model_modules.py:164); `utils.loss_utils.device` is whatever the reference picks on this
CPU-only box; train mode: nn.Dropout.forward replaced by a multiply with the known mask of each site, in call order.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

from . import stage2_port as s2
from .gen_golden import OUT, REF, _MaskQueue, summarize


def _import():
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv); sys.modules.setdefault("torchvision.transforms", tvt)
    sys.path.insert(0, REF)
    import models.coxranking_models_pretrained as cm
    import models.nll_models_pretrained as nm
    import utils.loss_utils as lu
    return nm, cm, lu


def _mask_order(m, masks):
    """Masks in the order the reference's forward calls nn.Dropout."""
    if not masks:
        return []
    tt = m["train_type"]
    if m["kind"] == "mm" and tt == "late-fcnn":
        return [masks["d0"], masks["d1"], masks["d2"]]            # layer_MRI, layer_WSI, layer_omic (:146-148)
    if m["kind"] == "mm" and tt == "late-highway":
        return [masks["h0"], masks["h1"], masks["h2"]]
    if tt == "kronecker":
        nmod = sum(k in m["mode"] for k in ("radio", "path", "omic"))
        return [masks[f"o{i}"] for i in range(nmod)] + [masks["post"], masks["enc1"], masks["enc2"]]
    return [masks["d0"]] if "d0" in masks else []


def main():
    nm, cm, lu = _import()
    orig_ft = torch.cuda.FloatTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    out, meta = {}, s2.case_meta()
    try:
        for name, m in meta.items():
            mod = nm if m["family"] == "nll" else cm
            kw = dict(n_classes=m["K"], mode=m["mode"], train_type=m["train_type"], n_layers=m["n_layers"])
            for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
                torch.manual_seed(0)
                model = (mod.unimonal_pretrained if m["kind"] == "uni" else mod.multimodal_pretrained)(**kw)
                shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
                if tag == "f64":
                    meta[name]["shapes"] = {k: list(v) for k, v in shapes.items()}
                sd_np = s2.state_dict_for(shapes, m["seed"])
                model.load_state_dict({k: torch.as_tensor(v) for k, v in sd_np.items()}, strict=True)
                model = model.to(dt)
                model.train() if m["train"] else model.eval()
                hs, Y, c, t = s2.batch_for(m)
                hr, hp, ho = [torch.as_tensor(h).to(dt) for h in hs]
                masks = s2.masks_for(m)
                with _MaskQueue() as mq:
                    mq.queue = [torch.as_tensor(x) for x in _mask_order(m, masks)]
                    risk, hazards, S = model(h_radio=hr, h_path=hp, h_omic=ho)
                    assert not mq.queue, (name, len(mq.queue))
                Yt, ct = torch.as_tensor(Y), torch.as_tensor(c).to(dt)
                spec = m["loss"]
                if spec[0] == "nll":
                    loss = lu.NLLSurvLoss(alpha=spec[1])(hazards=hazards, S=S, Y=Yt, c=ct)
                elif spec[0] == "cox":
                    loss = lu.CoxSurvLoss()(risks=risk, times=torch.tensor(t), c=ct)
                elif spec[0] == "rank":
                    loss = lu.RankingSurvLoss(phi=spec[1], reduction=spec[2])(risks=risk.reshape(-1), times=torch.tensor(t), c=ct)
                else:
                    loss = lu.RankingNLLSurvLoss(phi=spec[1], reduction=spec[2], alpha=spec[3], nll_ratio=spec[4])(
                        hazards=hazards, risks=risk, S=S, Y=Yt, c=ct)
                loss.backward()
                key = f"{name}/{tag}"
                out[key + "/risk"] = risk.detach().double().numpy()
                out[key + "/loss"] = np.float64(loss.item())
                if hazards is not None:
                    out[key + "/hazards"] = hazards.detach().double().numpy()
                    out[key + "/S"] = S.detach().double().numpy()
                for k, p in model.named_parameters():
                    g = p.grad if p.grad is not None else torch.zeros_like(p)
                    summarize(f"{key}/grad/{k}", g.detach().double().numpy(), out)
                for k, b in model.named_buffers():
                    if "running" in k:
                        summarize(f"{key}/buf/{k}", b.detach().double().numpy(), out)
        # same-seed initial state (utils/utils_pretrained.py:145-154), two representative configs
        for tag, (mod, cls, kw) in {"init_nll_mm_late_highway": (nm, "multimodal_pretrained", dict(train_type="late-highway", mode="radio_path_omic", n_layers=2)),
                                    "init_cox_uni_fcnn": (cm, "unimonal_pretrained", dict(train_type="fcnn", mode="path"))}.items():
            torch.manual_seed(1234)
            mdl = getattr(mod, cls)(**kw)
            for k, v in mdl.state_dict().items():
                summarize(f"{tag}/{k}", v.detach().double().numpy(), out, full_below=0)
    finally:
        torch.cuda.FloatTensor = orig_ft
    out["meta"] = np.array(json.dumps(dict(
        cases=meta, shims=["torchvision stub", "torch.cuda.FloatTensor -> torch.FloatTensor", "nn.Dropout.forward -> known mask"],
        generator="oracle/gen_golden_stage2.py")))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "stage2.npz"), **out)
    print("wrote", os.path.join(OUT, "stage2.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
