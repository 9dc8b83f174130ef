"""Omic self-normalising network head; drop-in for models/model_genomic.py of the reference
(MaxNet_base ctor :13-39, MaxNet.forward :53-72, state_dict keys fc_omic.{i}.0.*, classifier.*)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.utils import init_max_weights
from .model_modules import SNN_Block, snn_stack


OMIC_SIZES = {"small": (256, 256), "big": (1024, 256)}         # hidden widths of the SNN (model_genomic.py:17)


class MaxNet_base(nn.Module):
    """Parameter container: `fc_omic` (SNN blocks: Linear + SELU + AlphaDropout) and `classifier`."""

    def __init__(self, input_dim: int, model_size_omic: str = "small", bag_loss=None, n_classes: int = 4):
        super().__init__()
        self.n_classes, self.bag_loss = n_classes, bag_loss
        self.size_dict_omic = {name: list(w) for name, w in OMIC_SIZES.items()}
        widths = (input_dim,) + OMIC_SIZES[model_size_omic]
        # block 0 keeps SNN_Block's default dropout, the others pass 0.25 explicitly, as the reference does (:22-24)
        self.fc_omic = nn.Sequential(*[SNN_Block(dim1=a, dim2=b) if i == 0 else SNN_Block(dim1=a, dim2=b, dropout=0.25)
                                       for i, (a, b) in enumerate(zip(widths[:-1], widths[1:]))])
        # discrete-hazard heads emit n_classes logits, Cox / ranking heads one risk score (:33-36)
        self.classifier = nn.Linear(widths[-1], n_classes if "nll" in bag_loss else 1)
        init_max_weights(self)

    def relocate(self):
        self.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))

    def forward(self, **kwargs):
        pass            # abstract in the reference too


class MaxNet(MaxNet_base):
    def __init__(self, input_dim: int, model_size_omic: str = "small", bag_loss=None, n_classes: int = 4):
        super().__init__(input_dim, model_size_omic, bag_loss, n_classes)

    def cox_step_ok(self, x):
        """True when cox_step can take this batch: the `small` net with a Cox head, B <= 256, input_dim <= 256."""
        return ("nll" not in self.bag_loss and torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and 1 <= x.shape[0] <= 256
                and x.shape[1] <= 256 and self.fc_omic[0][0].weight.shape[0] == 256 and self.fc_omic[1][0].weight.shape[0] == 256
                and len(self.fc_omic) == 2 and self.classifier.weight.shape[0] == 1
                and all(p.requires_grad for p in self.parameters()))

    def _times_to_device(self, t, device):
        """Event times as the loader delivers them (a host array) -> float64 on the device through a small ring of PINNED
        staging buffers and an asynchronous copy: no pageable copy, no host synchronisation per step (the reference's loop
        does `torch.tensor(event_time)` + a blocking .to(device) every step, utils/core_utils.py:204).  A ring slot is
        reused only after the copy that last read it has completed (an event per slot, normally long done)."""
        n = int(t.size)
        ring = self.__dict__.get("_mmf_times_ring")
        if ring is None or ring["cap"] < n or ring["device"] != device:
            cap = max(256, n)
            ring = dict(cap=cap, device=device, i=0, bufs=[torch.empty(cap, dtype=torch.float64).pin_memory() for _ in range(4)],
                        evs=[None] * 4)
            self.__dict__["_mmf_times_ring"] = ring          # not a parameter / buffer: stays out of state_dict
        k = ring["i"] = (ring["i"] + 1) % 4
        if ring["evs"][k] is not None:
            ring["evs"][k].synchronize()
        buf = ring["bufs"][k][:n]
        buf.numpy()[:] = t.reshape(-1)
        out = buf.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ring["evs"][k] = ev
        return out

    def cox_step(self, genomic_features, times, c, loss_scale=1.0, grad_out=None, accumulate=None):
        """Extension of the reference surface (the training-loop mirror uses it): `risk = model(genomic_features=x)[0]`,
        `loss = CoxSurvLoss()(risks=risk, times=times, c=c)`, `(loss * loss_scale).backward()` as ONE launch
        (ops.maxnet_cox_step), with the same dropout draw.  `times`: anything CoxSurvLoss accepts; a float64 CUDA tensor
        saves the per-step host-to-device copy.  Gradients land in the parameters' .grad (accumulated) -- or in `grad_out`
        (tensors in self.parameters() order; overwritten unless `accumulate`).  Returns (risk [B], loss), detached."""
        import numpy as np
        x = genomic_features
        params = [self.fc_omic[0][0].weight, self.fc_omic[0][0].bias, self.fc_omic[1][0].weight, self.fc_omic[1][0].bias,
                  self.classifier.weight, self.classifier.bias]
        if grad_out is not None:
            grads, accumulate = list(grad_out), bool(accumulate)
        else:
            missing = [p for p in params if p.grad is None]
            accumulate = len(missing) < len(params)
            for p in missing:
                p.grad = (torch.zeros_like if accumulate else torch.empty_like)(p)
            grads = [p.grad for p in params]
        if not (torch.is_tensor(times) and times.is_cuda and times.dtype == torch.float64):
            times = self._times_to_device(np.asarray(times.cpu() if torch.is_tensor(times) else times, dtype=np.float64), x.device)
        tr = self.training
        seed = ops.next_dropout_seed() if tr else 0
        with torch.no_grad():
            return ops.maxnet_cox_step(x, *params, times, c, grads, loss_scale=loss_scale, accumulate=accumulate,
                                       p_drop=self.fc_omic[0][2].p if tr else 0.0, seed=seed)

    def forward(self, **kwargs):
        x = kwargs["genomic_features"]
        if kwargs.get("return_features"):
            return snn_stack(self.fc_omic, x, self.training)
        # SNN blocks + classifier as one autograd node (block i uses dropout site i, as snn_stack does)
        tr = self.training
        layers = [(blk[0].weight, blk[0].bias, "selu", "alpha" if tr else "none", blk[2].p if tr else 0.0, i)
                  for i, blk in enumerate(self.fc_omic)]
        seed = ops.next_dropout_seed() if tr else 0
        if "nll" in self.bag_loss:
            # The reference unsqueezes to [1 x B x K] and then takes topk / cumprod over dim=1, i.e. over the
            # BATCH axis (model_genomic.py:63-69).  Reproduced as is: hazards come from the HIP dense kernel,
            # the two degenerate axis ops are plain tensor ops on a [1 x B x K] view.
            layers.append((self.classifier.weight, self.classifier.bias, "sigmoid", "none", 0.0, 0))
            hazards = ops.mlp(x, layers, seed).unsqueeze(0)
            Y_hat = torch.topk(hazards.detach(), 1, dim=1)[1]   # sigmoid is monotone: same indices as topk(logits)
            S = torch.cumprod(1 - hazards, dim=1)
            return hazards, S, Y_hat, None
        layers.append((self.classifier.weight, self.classifier.bias, "none", "none", 0.0, 0))
        risk = ops.mlp(x, layers, seed).squeeze()
        return risk, None, None, None
