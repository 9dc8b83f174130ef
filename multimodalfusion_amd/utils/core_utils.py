"""Training / validation loops with the reference's call surface (utils/core_utils.py:173-264 and :267-355):
same signature, same per-bag order of operations (forward, loss, l1 regulariser added AFTER the /gc division,
backward, optimizer step every `gc` bags), same logged quantities.  The forward, loss and backward run in the
HIP kernels through the drop-in modules; this file is host control flow only.

Differences, all outside the arithmetic:
  * the c-index is computed with a small numpy restatement of sksurv's concordance_index_censored (sksurv is
    not a dependency of this package);
  * optional one-bag-per-GPU data parallelism (`dp=True` under torch.distributed): each rank takes every
    world_size-th bag and the flat gradient buffer is all-reduced once per optimizer step (see dp.py);
  * host synchronisation (`loss.item()`) is deferred to the end of the epoch instead of every bag.
"""
from __future__ import annotations

import numpy as np
import torch

from ..dp import FlatGradBuffer
from ..optim import FlatAdam
from .loss_utils import CoxSurvLoss, NLLSurvLoss


def concordance_index_censored(event_indicator, event_time, estimate, tied_tol=1e-8):
    """Harrell's c for right-censored data (sksurv.metrics.concordance_index_censored semantics):
    a pair (i, j) is comparable when the shorter time is an observed event; risk ties count 1/2."""
    e = np.asarray(event_indicator, dtype=bool)
    t = np.asarray(event_time, dtype=np.float64)
    r = np.asarray(estimate, dtype=np.float64)
    n = len(t)
    conc = disc = tied = 0.0
    for i in range(n):
        if not e[i]:
            continue
        later = (t > t[i]) | ((t == t[i]) & ~e)     # j outlived i (or was censored at the same time)
        later[i] = False
        d = r[i] - r[later]
        conc += float((d > tied_tol).sum())
        disc += float((d < -tied_tol).sum())
        tied += float((np.abs(d) <= tied_tol).sum())
    comparable = conc + disc + tied
    cindex = (conc + 0.5 * tied) / comparable if comparable > 0 else float("nan")
    return cindex, conc, disc, tied, 0


def _to_device(radio_features, path_features, genomic_features, label, c, device):
    feats = {i: r.to(device, non_blocking=True) for i, r in radio_features.items()}
    feats["path_features"] = path_features.to(device, non_blocking=True)
    feats["genomic_features"] = genomic_features.to(device, non_blocking=True).float()
    return feats, label.to(device), c.to(device)


def _is_sentinel(t):
    """The dataset marks a missing modality with zeros((1, 1)) (utils/core_utils.py:185-192 of the reference).
    Shape is checked first so that real bags (possibly already on the GPU via feed.DevicePrefetcher) cost no sync."""
    return tuple(t.shape) == (1, 1) and not bool(t.any())


def _skip(mode, radio_features, path_features, genomic_features):
    if "omic" in mode and _is_sentinel(genomic_features):
        return True
    if "path" in mode and _is_sentinel(path_features):
        return True
    if "radio" in mode and all(_is_sentinel(r) for r in radio_features.values()):
        return True
    return False


def train_loop_survival(epoch, model, loader, optimizer, n_classes, mode, writer=None, loss_fn=None, reg_fn=None,
                        lambda_reg=0., gc=16, t_bin=None, dp=False, grad_buffer=None, inflight=1):
    """utils/core_utils.py:173-264.  Extras (all off by default): `dp` = one bag per rank with one all-reduce per
    optimizer step; a `FlatAdam` optimizer = fused L1 + Adam tail; `inflight` > 1 (needs FlatAdam) = the bags of an
    accumulation window run round-robin on that many HIP streams, each into its own gradient slot
    (pipeline.BagsInFlight)."""
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model.train()
    pipe = None
    if inflight > 1:
        if not isinstance(optimizer, FlatAdam):
            raise ValueError("inflight > 1 needs the FlatAdam optimizer (flat gradient buffer)")
        from ..pipeline import BagsInFlight
        pipe = BagsInFlight(model, inflight, device)
    world, rank = 1, 0
    if dp and torch.distributed.is_available() and torch.distributed.is_initialized():
        world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        if grad_buffer is None:
            grad_buffer = FlatGradBuffer(model)
    losses, regs, all_risk, all_c, all_t = [], [], [], [], []
    seen = 0
    for batch_idx, (radio_features, path_features, genomic_features, label, event_time, c) in enumerate(loader):
        if _skip(mode, radio_features, path_features, genomic_features):
            continue
        if world > 1 and batch_idx % world != rank:
            continue
        feats, label, c = _to_device(radio_features, path_features, genomic_features, label, c, device)

        def forward_loss():
            hazards, S, Y_hat, _ = model(**feats)
            if isinstance(loss_fn, CoxSurvLoss):
                return hazards, loss_fn(risks=hazards, times=torch.as_tensor(np.asarray(event_time)), c=c)
            if isinstance(loss_fn, NLLSurvLoss):
                return -torch.sum(S, dim=1), loss_fn(hazards=hazards, S=S, Y=label, c=c)
            raise NotImplementedError(type(loss_fn))

        fused_tail = isinstance(optimizer, FlatAdam)
        if pipe is not None:
            box = {}

            def bag():
                box["risk"], box["loss"] = forward_loss()
                return box["loss"] / (gc * world)

            pipe.run(bag)
            risk, loss = box["risk"], box["loss"]
            loss_reg = optimizer.l1_value() if (reg_fn is not None and lambda_reg) else 0
            losses.append(loss.detach())
            regs.append(loss_reg.detach() if torch.is_tensor(loss_reg) else torch.tensor(float(loss_reg), device=device))
            all_risk.append(risk.detach().reshape(-1))
            all_c.append(c.detach().reshape(-1))
            all_t.append(np.asarray(event_time).reshape(-1))
            seen += 1
            if seen % gc == 0:
                optimizer.flat_g.copy_(pipe.reduce(all_reduce=world > 1))
                optimizer.lambda_l1 = lambda_reg if reg_fn is not None else 0.0
                optimizer.step(l1_micro_batches=gc * world)
                pipe.release()
            continue
        risk, loss = forward_loss()
        if fused_tail:
            # fused per-step tail: the L1 term never enters autograd; its gradient (lambda * sign(W) per micro-batch)
            # is added inside the Adam kernel, its value is a device scalar for logging only
            loss_reg = optimizer.l1_value() if (reg_fn is not None and lambda_reg) else 0
        else:
            loss_reg = 0 if reg_fn is None else reg_fn(model) * lambda_reg
        losses.append(loss.detach())
        regs.append(loss_reg.detach() if torch.is_tensor(loss_reg) else torch.tensor(float(loss_reg), device=device))
        all_risk.append(risk.detach().reshape(-1))
        all_c.append(c.detach().reshape(-1))
        all_t.append(np.asarray(event_time).reshape(-1))
        # the reference: loss = loss / gc + loss_reg ; backward ; step every gc bags (core_utils.py:242-247)
        if fused_tail:
            (loss / (gc * world)).backward()
        else:
            (loss / (gc * world) + loss_reg).backward()
        seen += 1
        if seen % gc == 0:
            if fused_tail:
                if world > 1:
                    optimizer.all_reduce()
                optimizer.lambda_l1 = lambda_reg if reg_fn is not None else 0.0
                optimizer.step(l1_micro_batches=gc * world)
                optimizer.zero_grad()
                continue
            if grad_buffer is not None and world > 1:
                grad_buffer.all_reduce()
            optimizer.step()
            if grad_buffer is not None:
                grad_buffer.zero()
            else:
                optimizer.zero_grad()
    n = max(len(losses), 1)
    loss_vals = torch.stack(losses).float().cpu().numpy() if losses else np.zeros(0)
    reg_vals = torch.stack(regs).float().cpu().numpy() if regs else np.zeros(0)
    train_loss_surv = float(loss_vals.sum()) / n
    train_loss = float((loss_vals + reg_vals).sum()) / n
    risks = torch.cat(all_risk).cpu().numpy() if all_risk else np.zeros(0)
    cens = torch.cat(all_c).cpu().numpy() if all_c else np.zeros(0)
    times = np.concatenate(all_t) if all_t else np.zeros(0)
    c_index = concordance_index_censored((1 - cens).astype(bool), times, risks, tied_tol=1e-08)[0]
    print('Epoch: {}, train_loss_surv: {:.4f}, train_loss: {:.4f}, train_c_index: {:.4f}'.format(
        epoch, train_loss_surv, train_loss, c_index))
    if writer:
        writer.add_scalar('train/loss_surv', train_loss_surv, epoch)
        writer.add_scalar('train/loss', train_loss, epoch)
        writer.add_scalar('train/c_index', c_index, epoch)
    return dict(loss_surv=train_loss_surv, loss=train_loss, c_index=c_index, losses=loss_vals, risks=risks)


def validate_survival(cur, epoch, model, loader, n_classes, mode, early_stopping=None, writer=None, loss_fn=None,
                      reg_fn=None, lambda_reg=0., results_dir=None, t_bin=None):
    """utils/core_utils.py:267-355: eval-mode forward + loss + c-index (early stopping hook kept)."""
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model.eval()
    losses, regs, all_risk, all_c, all_t = [], [], [], [], []
    with torch.no_grad():
        for (radio_features, path_features, genomic_features, label, event_time, c) in loader:
            if _skip(mode, radio_features, path_features, genomic_features):
                continue
            feats, label, c = _to_device(radio_features, path_features, genomic_features, label, c, device)
            hazards, S, Y_hat, _ = model(**feats)
            if isinstance(loss_fn, CoxSurvLoss):
                risk = hazards
                loss = loss_fn(risks=risk, times=torch.as_tensor(np.asarray(event_time)), c=c)
            else:
                risk = -torch.sum(S, dim=1)
                loss = loss_fn(hazards=hazards, S=S, Y=label, c=c, alpha=0)
            loss_reg = 0 if reg_fn is None else reg_fn(model) * lambda_reg
            losses.append(loss)
            regs.append(loss_reg if torch.is_tensor(loss_reg) else torch.tensor(float(loss_reg), device=device))
            all_risk.append(risk.reshape(-1))
            all_c.append(c.reshape(-1))
            all_t.append(np.asarray(event_time).reshape(-1))
    n = max(len(losses), 1)
    loss_vals = torch.stack(losses).float().cpu().numpy()
    reg_vals = torch.stack(regs).float().cpu().numpy()
    val_loss_surv = float(loss_vals.sum()) / n
    val_loss = float((loss_vals + reg_vals).sum()) / n
    risks = torch.cat(all_risk).cpu().numpy()
    cens = torch.cat(all_c).cpu().numpy()
    times = np.concatenate(all_t)
    c_index = concordance_index_censored((1 - cens).astype(bool), times, risks, tied_tol=1e-08)[0]
    if writer:
        writer.add_scalar('val/loss_surv', val_loss_surv, epoch)
        writer.add_scalar('val/loss', val_loss, epoch)
        writer.add_scalar('val/c-index', c_index, epoch)
    if early_stopping is not None:
        early_stopping(epoch, val_loss_surv, model)
        if getattr(early_stopping, "early_stop", False):
            print("Early stopping")
            return True
    return False


def summary_survival(model, loader, n_classes, mode, t_bin=None, loss_fn=None):
    """utils/core_utils.py:358-430: eval-mode pass over a loader -> (patient_results, c_index).
    risk = the head's scalar output for Cox / ranking losses, -sum(S) for the discrete-hazard losses; subjects whose
    required modality is the "missing" sentinel are skipped exactly as in the reference (:379-386).  Subject ids are
    read from `loader.dataset.slides_radio_data['subject_id']` when the loader has one (the reference requires it),
    else the running index is used.  One device -> host copy at the end instead of one per subject."""
    from .loss_utils import RankingSurvLoss
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model.eval()
    ids = None
    ds = getattr(loader, "dataset", None)
    if ds is not None and hasattr(ds, "slides_radio_data"):
        ids = list(ds.slides_radio_data["subject_id"])
    all_ids, all_risk, all_c, all_t, all_y = [], [], [], [], []
    count = 0
    with torch.no_grad():
        for (radio_features, path_features, genomic_features, label, event_time, c) in loader:
            n = len(label)
            sid = ids[count:count + n] if ids is not None else list(range(count, count + n))
            count += n
            if _skip(mode, radio_features, path_features, genomic_features):
                continue
            feats, label, c = _to_device(radio_features, path_features, genomic_features, label, c, device)
            hazards, S, Y_hat, _ = model(**feats)
            risk = hazards if isinstance(loss_fn, (CoxSurvLoss, RankingSurvLoss)) else -torch.sum(S, dim=1)
            all_ids.extend(sid)
            all_risk.append(risk.reshape(-1))
            all_c.append(c.reshape(-1))
            all_t.append(np.asarray(event_time).reshape(-1))
            all_y.append(label.reshape(-1))
    risks = torch.cat(all_risk).cpu().numpy()
    cens = torch.cat(all_c).cpu().numpy()
    labels = torch.cat(all_y).cpu().numpy()
    times = np.concatenate(all_t)
    patient_results = {"subject_id": np.asarray(all_ids), "risk": risks, "disc_label": labels, "survival": times,
                       "censorship": cens}
    c_index = concordance_index_censored((1 - cens).astype(bool), times, risks, tied_tol=1e-08)[0]
    return patient_results, c_index
