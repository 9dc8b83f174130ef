"""Stage-2 models on the exported 256-d embeddings, discrete-hazard (nll) heads: drop-in for
models/nll_models_pretrained.py:13-197 of the reference (same class names -- including the reference's spelling
`unimonal_pretrained` --, constructor signatures, submodule trees and therefore state_dict keys).
All arithmetic runs in the HIP kernels behind include/mmf_amil.h (dense, batchnorm, highway mix, Kronecker fusion,
hazards)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.utils_pretrained import initialize_weights
from .model_modules import Highway, XlinearFusion, fcnn_block


def _seed(training):
    return ops.next_dropout_seed() if training else 0


class unimonal_pretrained(nn.Module):
    """models/nll_models_pretrained.py:13-62."""

    def __init__(self, dropout=True, n_classes=4, mode=None, train_type=None, bag_loss=None, n_layers=1):
        super().__init__()
        self.n_classes = n_classes
        self.train_type = train_type
        self.bag_loss = bag_loss
        self.mode = mode
        if self.train_type == "fcnn":
            self.classifier = nn.Sequential(*[nn.Linear(256, n_classes), nn.Dropout(0.7)])
        elif self.train_type == "highway":
            self.highway = Highway(256, n_layers)
            self.classifier = nn.Linear(256, n_classes)
        initialize_weights(self)

    def relocate(self):
        self.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))

    def forward(self, **kwargs):
        h = kwargs[{"path": "h_path", "radio": "h_radio", "omic": "h_omic"}[self.mode]]
        if self.train_type == "fcnn":
            lin, drop = self.classifier[0], self.classifier[1]
            p = drop.p if self.training else 0.0
            logits = ops.dense(h, lin.weight, lin.bias, drop_kind="dropout" if p > 0 else "none", drop_p=p,
                               seed=_seed(self.training), site=0)
        elif self.train_type == "highway":
            logits = ops.dense(self.highway(h), self.classifier.weight, self.classifier.bias)
        else:
            raise NotImplementedError(f"train_type {self.train_type!r}")     # 'residual' is commented out in the reference (:27-29)
        risk, hazards, S, _ = ops.hazards_from_logits(logits)
        return risk, hazards, S


def _pick(mode, h_radio, h_path, h_omic):
    """The modality order of the reference's cat / v_list (nll_models_pretrained.py:153-160, 164-171, 180-187):
    [radio, path] | [radio, omic] | [omic, path] | [radio, path, omic]."""
    r, p, o = "radio" in mode, "path" in mode, "omic" in mode
    if r and p and not o:
        return [h_radio, h_path]
    if r and o and not p:
        return [h_radio, h_omic]
    if o and p and not r:
        return [h_omic, h_path]
    if r and p and o:
        return [h_radio, h_path, h_omic]
    raise NotImplementedError(f"mode {mode!r}")


class multimodal_pretrained(nn.Module):
    """models/nll_models_pretrained.py:66-197."""

    def __init__(self, input_dim: int = 37, dropout=True, n_classes=4, mode="radio_path_omic", train_type=None,
                 bag_loss=None, n_layers=1):
        super().__init__()
        self.n_classes = n_classes
        self.mode = mode
        self.train_type = train_type
        self.bag_loss = bag_loss
        num_modalities = sum(k in mode for k in ("radio", "path", "omic"))
        if train_type == "early-fcnn":
            self.classifier = nn.Sequential(*[nn.Linear(num_modalities * 256, 128), nn.BatchNorm1d(128), nn.ReLU(),
                                              nn.Dropout(0.7), nn.Linear(128, n_classes)])
        elif train_type == "late-fcnn":
            self.layer_WSI = nn.Sequential(*[nn.Linear(256, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.7)])
            self.layer_MRI = nn.Sequential(*[nn.Linear(256, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.7)])
            self.layer_omic = nn.Sequential(*[nn.Linear(256, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.7)])
            self.classifier = nn.Sequential(*[nn.Linear(num_modalities * 128, n_classes)])
        elif train_type == "early-highway":
            self.highway = Highway(num_modalities * 256, n_layers)
            self.classifier = nn.Linear(num_modalities * 256, n_classes)
        elif train_type == "late-highway":
            self.highway_radio = Highway(256, n_layers)
            self.highway_path = Highway(256, n_layers)
            self.highway_omic = Highway(256, n_layers)
            self.classifier = nn.Linear(num_modalities * 256, n_classes)
        elif train_type == "kronecker":
            self.xfusion = XlinearFusion(num_modalities=num_modalities, dropout_rate=0.7)
            self.classifier = nn.Linear(256, n_classes)
        initialize_weights(self)

    def relocate(self):
        self.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))

    def _late_layers(self, h_radio, h_path, h_omic, seed):
        """All three branches are evaluated, as the reference does (:144-151), even if the mode drops one."""
        if self.train_type == "late-fcnn":
            return (fcnn_block(self.layer_MRI, h_radio, seed or 0, 0), fcnn_block(self.layer_WSI, h_path, seed or 0, 1),
                    fcnn_block(self.layer_omic, h_omic, seed or 0, 2))
        return (self.highway_radio(h_radio, seed=None if seed is None else seed),
                self.highway_path(h_path, seed=None if seed is None else seed + 1),
                self.highway_omic(h_omic, seed=None if seed is None else seed + 2))

    def _logits(self, h_radio, h_path, h_omic):
        seed = _seed(self.training) if self.training else None
        if "late" in self.train_type:
            r, p, o = self._late_layers(h_radio, h_path, h_omic, seed)
            mm = torch.cat(_pick(self.mode, r, p, o), dim=1)
            cls = self.classifier[0] if isinstance(self.classifier, nn.Sequential) else self.classifier
            return ops.dense(mm, cls.weight, cls.bias)
        if "early" in self.train_type:
            mm = torch.cat(_pick(self.mode, h_radio, h_path, h_omic), dim=1)
            if self.train_type == "early-fcnn":
                return fcnn_block(self.classifier, mm, seed or 0, 0)
            return ops.dense(self.highway(mm, seed=seed), self.classifier.weight, self.classifier.bias)
        if self.train_type == "kronecker":
            mm = self.xfusion(v_list=_pick(self.mode, h_radio, h_path, h_omic), seed=seed)
            return ops.dense(mm, self.classifier.weight, self.classifier.bias)
        raise NotImplementedError(f"train_type {self.train_type!r}")

    def forward(self, h_radio, h_path, h_omic):
        logits = self._logits(h_radio, h_path, h_omic)
        risk, hazards, S, _ = ops.hazards_from_logits(logits)
        return risk, hazards, S
