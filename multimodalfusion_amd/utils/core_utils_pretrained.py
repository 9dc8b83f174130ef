"""Stage-2 training / validation loops with the reference's call surface
(utils/core_utils_pretrained.py:148-247 `train_loop_survival`, :249-327 `validate_survival`): batches of exported 256-d
embeddings `(radio, path, genomic, label, event_time, c, masks)`, models that return `(risk, hazards, S)`, loss
dispatch by class, `loss / gc + loss_reg` accumulation, optimizer step every `gc` batches, c-index at epoch end.

Deviation (documented, as in the stage-1 loop): the reference converts `reg_fn(model) * lambda_reg` to a Python float
(`.detach().cpu().numpy().item()`, :215-216) before adding it to the loss, so its L1 term never reaches the gradients;
that behaviour is kept exactly (the term only shows up in the logged loss).
"""
from __future__ import annotations

import numpy as np
import torch

from .core_utils import concordance_index_censored
from .loss_utils import CoxSurvLoss, NLLSurvLoss, RankingNLLSurvLoss, RankingSurvLoss


def _loss(loss_fn, risk, hazards, S, label, event_time, c, device):
    if isinstance(loss_fn, (CoxSurvLoss, RankingSurvLoss)):
        return loss_fn(risks=risk, times=torch.as_tensor(np.asarray(event_time)).to(device), c=c)
    if isinstance(loss_fn, NLLSurvLoss):
        return loss_fn(hazards=hazards, S=S, Y=label, c=c)
    if isinstance(loss_fn, RankingNLLSurvLoss):
        return loss_fn(hazards=hazards, risks=risk, S=S, Y=label, c=c)
    raise NotImplementedError(type(loss_fn).__name__)


def _run(model, loader, loss_fn, reg_fn, lambda_reg, device, train, optimizer=None, gc=16):
    loss_surv_sum, loss_sum = 0.0, 0.0
    all_risk, all_c, all_t = [], [], []
    for batch_idx, (radio_features, path_features, genomic_features, label, event_time, c, masks) in enumerate(loader):
        radio_features, path_features = radio_features.to(device), path_features.to(device)
        genomic_features, label, c = genomic_features.to(device), label.to(device), c.to(device)
        if train:
            risk, hazards, S = model(h_radio=radio_features, h_path=path_features, h_omic=genomic_features)
        else:
            with torch.no_grad():
                risk, hazards, S = model(h_radio=radio_features, h_path=path_features, h_omic=genomic_features)
        loss = _loss(loss_fn, risk, hazards, S, label, event_time, c, device)
        loss_value = loss.item()
        loss_reg = 0 if reg_fn is None else float((reg_fn(model) * lambda_reg).detach().cpu().numpy().item())
        all_risk.append(np.atleast_1d(risk.detach().cpu().numpy().squeeze()))
        all_c.append(np.atleast_1d(c.detach().cpu().numpy()))
        all_t.append(np.atleast_1d(np.asarray(event_time)))
        loss_surv_sum += loss_value
        loss_sum += loss_value + loss_reg
        if train:
            (loss / gc + loss_reg).backward()
            if (batch_idx + 1) % gc == 0:
                optimizer.step()
                optimizer.zero_grad()
    n = max(len(loader), 1)
    risk, t, cens = np.concatenate(all_risk).flatten(), np.concatenate(all_t).flatten(), np.concatenate(all_c).flatten()
    c_index = concordance_index_censored((1 - cens).astype(bool), t, risk, tied_tol=1e-08)[0]
    return loss_surv_sum / n, loss_sum / n, c_index


def train_loop_survival(epoch, model, loader, optimizer, n_classes, mode, writer=None, loss_fn=None, reg_fn=None,
                        lambda_reg=0., gc=16, t_bin=None, train_type=None, verbose=True):
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model.train()
    loss_surv, loss, c_index = _run(model, loader, loss_fn, reg_fn, lambda_reg, device, True, optimizer, gc)
    if verbose:
        print('Epoch: {}, train_loss_surv: {:.4f}, train_loss: {:.4f}, train_c_index: {:.4f}'.format(epoch, loss_surv, loss, c_index))
    if writer:
        writer.add_scalar('train/loss_surv', loss_surv, epoch)
        writer.add_scalar('train/loss', loss, epoch)
        writer.add_scalar('train/c_index', c_index, epoch)
    return loss_surv, loss, c_index


def validate_survival(cur, epoch, model, loader, n_classes, mode, early_stopping=None, writer=None, loss_fn=None,
                      reg_fn=None, lambda_reg=0., results_dir=None, t_bin=None, train_type=None, verbose=True):
    import os
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model.eval()
    loss_surv, loss, c_index = _run(model, loader, loss_fn, reg_fn, lambda_reg, device, False)
    if writer:
        writer.add_scalar('val/loss_surv', loss_surv, epoch)
        writer.add_scalar('val/loss', loss, epoch)
        writer.add_scalar('val/c-index', c_index, epoch)
    if epoch == 10 and results_dir:
        torch.save(model.state_dict(), os.path.join(results_dir, 's_%d_mid_checkpoint.pt' % cur))
    if verbose:
        print('\nVal Set, val_loss_surv: {:.4f}, val_loss: {:.4f}, val c-index: {:.4f}'.format(loss_surv, loss, c_index))
    if early_stopping:
        assert results_dir
        early_stopping(epoch, loss_surv, model, ckpt_name=os.path.join(results_dir, "s_{}_minloss_checkpoint.pt".format(cur)))
        if early_stopping.early_stop:
            print("Early stopping")
            return True
    return False
