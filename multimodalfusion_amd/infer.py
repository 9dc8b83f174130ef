"""Forward-only consumers of the path, on the MI355X kernels.

The reference runs three kinds of inference through the same heads:
  * embedding export for stage 2 -- `model(..., return_features=True)` per subject under `torch.no_grad()`
    (pre_trained_feature.py:116-162);
  * per-patient inference -- hazards, risk = -sum(S), raw attention scores (utils/heatmap_utils.py:249-275);
  * attention scoring of patch batches for heat-maps -- `_, _, _, A = model(path_features=features)` per batch of 512
    patch embeddings (utils/heatmap_utils.py:111-150).
Everything below is host plumbing around the drop-in heads; under `torch.no_grad()` in eval mode the heads take the
forward-only C-ABI entry points (include/mmf_amil.h: mmf_amil[_bf16]_infer), which save nothing for a backward.
File I/O (.pt / .h5), WSI handling and the image feature extractor stay with the caller: they are outside the path.
"""
from __future__ import annotations

import numpy as np
import torch

from .models import (MaxNet, MIL_Attention_fc_surv_path, MIL_Attention_fc_surv_radio, MM_MIL_Attention_fc_surv)


def _dev(model):
    return next(model.parameters()).device


def extract_features(model, **inputs) -> torch.Tensor:
    """One subject's pooled embedding, as pre_trained_feature.py:128,144,160 computes it:
    `model(**inputs, return_features=True)` in eval mode without autograd.  Returns a CPU tensor ([1 x 256] for the
    path / radio heads, [B x 256] for the omic head)."""
    dev = _dev(model)
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            kw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inputs.items()}
            if isinstance(model, MaxNet) and "genomic_features" in kw:
                kw["genomic_features"] = kw["genomic_features"].float()          # pre_trained_feature.py:158
            feat = model(**kw, return_features=True)
        return feat.detach().cpu()
    finally:
        model.train(was_training)


def extract_features_for_subjects(models: dict, subjects, skip=lambda subject_id, modality: False):
    """The export loop of pre_trained_feature.py:116-162 over an iterable of
    (subject_id, radio_features: dict, path_features, genomic_features) tuples.  `models` maps
    'path' / 'radio' / 'omic' to loaded heads (any subset).  A modality whose tensor is the reference's
    "missing" sentinel (`torch.zeros((1, 1))`, pre_trained_feature.py:122,135,153) is skipped.
    Yields (subject_id, modality, features_cpu)."""
    sentinel = torch.zeros((1, 1))

    def missing(t):
        return tuple(t.shape) == (1, 1) and torch.equal(t.detach().float().cpu(), sentinel)

    for subject_id, radio_features, path_features, genomic_features in subjects:
        if "path" in models and path_features is not None and not missing(path_features) and not skip(subject_id, "path"):
            yield subject_id, "path", extract_features(models["path"], path_features=path_features)
        if ("radio" in models and radio_features and not all(missing(r) for r in radio_features.values())
                and not skip(subject_id, "radio")):
            yield subject_id, "radio", extract_features(models["radio"], **radio_features)
        if "omic" in models and genomic_features is not None and not missing(genomic_features) and not skip(subject_id, "omic"):
            yield subject_id, "omic", extract_features(models["omic"], genomic_features=genomic_features)


def infer_patient(model, features, bins=None, label=None, verbose=False):
    """utils/heatmap_utils.py:249-275: returns (Y_hat_model, risk, A_final) with risk = -sum(S) and A_final the raw
    (pre-softmax) attention scores as an [N x 1] numpy array.  `features` is the bag tensor for a path head and the
    dict of modality bags for a radio head, exactly as the reference passes them."""
    dev = _dev(model)
    with torch.no_grad():
        if isinstance(model, MIL_Attention_fc_surv_path):
            hazards, survival, Y_hat_model, A = model(path_features=features.to(dev))
        elif isinstance(model, MIL_Attention_fc_surv_radio):
            hazards, survival, Y_hat_model, A = model(**{k: v.to(dev) for k, v in features.items()})
        else:
            raise NotImplementedError            # as the reference (heatmap_utils.py:268-269)
        risk = -torch.sum(survival, dim=1).cpu().numpy()
        Y_hat = int(np.digitize(risk, np.array(bins))[0] - 1) if bins is not None else None
        A_final = A.view(-1, 1).cpu().numpy()
    Y_hat_model = Y_hat_model.cpu().numpy()[0][0]
    if verbose:
        print("Y_hat: {}, Y: {}, risk: {}, hazards: {}".format(
            Y_hat, label, risk, ["{:.4f}".format(p) for p in hazards.cpu().flatten()]))
    return Y_hat_model, risk, A_final


def score_patch_batches(model, feature_batches, ref_scores=None):
    """Attention scoring of utils/heatmap_utils.py:129-141: for every batch of patch embeddings ([n x 1024], n <= 512
    in the reference) the raw attention scores of `model(path_features=features)`; optionally mapped to percentiles
    of `ref_scores` (score2percentile, heatmap_utils.py:32-34).  Batches are independent bags of the SAME head, so
    they are simply run back to back on the forward-only kernels; yields one [n x 1] float32 numpy array per batch."""
    dev = _dev(model)
    was_training = model.training
    model.eval()
    try:
        ref_sorted = np.sort(np.asarray(ref_scores).reshape(-1)) if ref_scores is not None else None
        for features in feature_batches:
            with torch.no_grad():
                A = model(path_features=features.to(dev), attention_only=True)
            A = A.view(-1, 1).cpu().numpy()
            if ref_sorted is not None:
                # scipy.stats.percentileofscore(ref, score) with the default kind='rank', vectorised
                lo = np.searchsorted(ref_sorted, A[:, 0], side="left")
                hi = np.searchsorted(ref_sorted, A[:, 0], side="right")
                A = ((lo + hi + (hi > lo)) * 50.0 / len(ref_sorted)).reshape(-1, 1).astype(A.dtype)
            yield A
    finally:
        model.train(was_training)
