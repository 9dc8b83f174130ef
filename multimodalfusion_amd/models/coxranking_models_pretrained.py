"""Stage-2 models with a scalar risk head (Cox / ranking losses): drop-in for
models/coxranking_models_pretrained.py:14-200 of the reference (same class names, constructor signatures, submodule
trees / state_dict keys).  forward returns (risk, None, None) as the reference does."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.utils_pretrained import initialize_weights
from .model_modules import Highway, Residual, XlinearFusion, fcnn_block
from .nll_models_pretrained import _pick, _seed


class unimonal_pretrained(nn.Module):
    """models/coxranking_models_pretrained.py:14-58."""

    def __init__(self, dropout=True, n_classes=4, mode="radio", train_type=None, bag_loss=None, n_layers=1):
        super().__init__()
        self.n_classes = n_classes
        self.train_type = train_type
        self.bag_loss = bag_loss
        self.mode = mode
        self.n_layers = n_layers
        if self.train_type == "fcnn":
            self.classifier = nn.Sequential(*[nn.Linear(256, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.7),
                                              nn.Linear(128, 1)])
        elif self.train_type == "highway":
            self.highway = Highway(256, n_layers)
            self.classifier = nn.Linear(256, 1)
        elif self.train_type == "residual":
            self.residual = Residual(256, n_layers)
            self.classifier = nn.Linear(256, 1)
        initialize_weights(self)

    def relocate(self):
        self.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))

    def forward(self, **kwargs):
        h = kwargs[{"path": "h_path", "radio": "h_radio", "omic": "h_omic"}[self.mode]]
        if self.train_type == "fcnn":
            risk = fcnn_block(self.classifier, h, _seed(self.training), 0)
        elif self.train_type == "highway":
            risk = ops.dense(self.highway(h), self.classifier.weight, self.classifier.bias)
        elif self.train_type == "residual":
            risk = ops.dense(self.residual(h), self.classifier.weight, self.classifier.bias)
        else:
            raise NotImplementedError(f"train_type {self.train_type!r}")
        return risk.squeeze(), None, None


class multimodal_pretrained(nn.Module):
    """models/coxranking_models_pretrained.py:62-200."""

    def __init__(self, dropout=True, n_classes=4, mode="radio_path_omic", train_type=None, bag_loss=None, n_layers=1):
        super().__init__()
        self.n_classes = n_classes
        self.mode = mode
        self.train_type = train_type
        self.bag_loss = bag_loss
        self.n_layers = n_layers
        num_modalities = sum(k in mode for k in ("radio", "path", "omic"))
        blk = lambda: nn.Sequential(*[nn.Linear(256, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.7), nn.Linear(128, 1)])
        if train_type == "late-fcnn":
            self.layer_WSI = blk()
            self.layer_MRI = blk()
            self.layer_omic = blk()
            self.classifier = nn.Sequential(*[nn.Linear(num_modalities, 1)])
        elif train_type == "early-fcnn":
            self.classifier = nn.Sequential(*[nn.Linear(num_modalities * 256, 128), nn.BatchNorm1d(128), nn.ReLU(),
                                              nn.Dropout(0.7), nn.Linear(128, 1)])
        elif train_type == "early-highway":
            self.highway = Highway(num_modalities * 256, n_layers)
            self.classifier = nn.Linear(num_modalities * 256, 1)
        elif train_type == "late-highway":
            self.highway_radio = Highway(256, n_layers)
            self.highway_path = Highway(256, n_layers)
            self.highway_omic = Highway(256, n_layers)
            self.classifier = nn.Linear(num_modalities * 256, 1)
        elif train_type == "kronecker":
            self.xfusion = XlinearFusion(num_modalities=num_modalities, dropout_rate=0.7)
            self.classifier = nn.Linear(256, 1)
        initialize_weights(self)

    def relocate(self):
        self.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))

    def forward(self, h_radio, h_path, h_omic):
        seed = _seed(self.training) if self.training else None
        if "late" in self.train_type:
            if self.train_type == "late-fcnn":
                r = fcnn_block(self.layer_MRI, h_radio, seed or 0, 0)
                p = fcnn_block(self.layer_WSI, h_path, seed or 0, 1)
                o = fcnn_block(self.layer_omic, h_omic, seed or 0, 2)
            else:
                r = self.highway_radio(h_radio, seed=seed)
                p = self.highway_path(h_path, seed=None if seed is None else seed + 1)
                o = self.highway_omic(h_omic, seed=None if seed is None else seed + 2)
            mm = torch.cat(_pick(self.mode, r, p, o), dim=1)
            cls = self.classifier[0] if isinstance(self.classifier, nn.Sequential) else self.classifier
            risk = ops.dense(mm, cls.weight, cls.bias).squeeze()
        elif "early" in self.train_type:
            mm = torch.cat(_pick(self.mode, h_radio, h_path, h_omic), dim=1)
            if self.train_type == "early-fcnn":
                risk = fcnn_block(self.classifier, mm, seed or 0, 0)
            else:
                risk = ops.dense(self.highway(mm, seed=seed), self.classifier.weight, self.classifier.bias)
        elif self.train_type == "kronecker":
            mm = self.xfusion(v_list=_pick(self.mode, h_radio, h_path, h_omic), seed=seed)
            risk = ops.dense(mm, self.classifier.weight, self.classifier.bias)
        else:
            raise NotImplementedError(f"train_type {self.train_type!r}")
        return risk, None, None
