"""Training / validation loops with the reference's call surface (utils/core_utils.py:173-264 and :267-355):
same signature, same per-bag order of operations (forward, loss, l1 regulariser added AFTER the /gc division,
backward, optimizer step every `gc` bags), same logged quantities.  The forward, loss and backward run in the
HIP kernels through the drop-in modules; this file is host control flow only.

Differences, all outside the arithmetic:
  * the c-index is computed with a small numpy restatement of sksurv's concordance_index_censored (sksurv is
    not a dependency of this package);
  * optional one-bag-per-GPU data parallelism (`dp=True` under torch.distributed): the loader is sharded by rank
    (feed.RankShard) and the flat gradient bucket is all-reduced once per optimizer step (see dp.py);
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


def _fused_step_ok(model, loss_fn, feats):
    """One bag = one C-ABI call (model.nll_step).  Only where that is exactly what `model(**feats)` + the stock loss would
    compute: the pathology head ITSELF (a subclass that overrides forward(), or any module / global hook, would be bypassed
    by a graph-free step -- those take the autograd path), the stock NLLSurvLoss, one 2-D fp32 / bf16 bag on the GPU, a
    classifier the kernel's single-workgroup tail holds (<= 32 classes), every parameter trainable."""
    from ..models.model_attention_mil_path import MIL_Attention_fc_surv_path
    import torch.nn.modules.module as tm
    x = feats.get("path_features")
    if not (type(loss_fn) is NLLSurvLoss and torch.is_tensor(x) and x.dim() == 2 and x.is_cuda
            and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    if type(model).forward is not MIL_Attention_fc_surv_path.forward or not hasattr(model, "nll_step"):
        return False
    if getattr(getattr(model, "classifier", None), "out_features", 1 << 30) > 32:
        return False
    hooked = lambda m: bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None))
    if any(hooked(m) for m in model.modules()):
        return False
    if tm._global_forward_hooks or tm._global_forward_pre_hooks or tm._global_backward_hooks or getattr(tm, "_global_backward_pre_hooks", None):
        return False
    return all(p.requires_grad for p in model.parameters())


class _Window:
    """Gradient accumulation window of `train_loop_survival`, kept on the optimizer object between calls because the
    reference's accumulated gradients survive the end of an epoch (a trailing partial window is NOT stepped and NOT
    cleared, utils/core_utils.py:245-247: its gradients join the first window of the next epoch).

    Three ways to hold the window's gradient, one interface:
      fused   FlatAdam: p.grad are views of its flat bucket [n grads | 2 control words];
      buffer  any torch optimizer + dp.FlatGradBuffer (made here when world > 1): same bucket layout;
      plain   any torch optimizer, world == 1: p.grad as autograd leaves them.
    With `inflight` > 1 (fused only) the bags run on side streams into per-stream slots (pipeline.BagsInFlight) that
    are folded into the bucket at every boundary and at the end of the epoch."""

    def __init__(self, model, optimizer, world, grad_buffer, inflight, device):
        self.opt, self.world = optimizer, world
        self.fused = isinstance(optimizer, FlatAdam)
        self.buf = None if self.fused else grad_buffer
        if world > 1 and not self.fused and self.buf is None:
            self.buf = FlatGradBuffer(model)
        self.pipe = None
        if inflight > 1:
            if not self.fused:
                raise ValueError("inflight > 1 needs the FlatAdam optimizer (flat gradient buffer)")
            from ..pipeline import BagsInFlight
            self.pipe = BagsInFlight(model, inflight, device)
        self.kept = 0            # bags that contributed since the last optimizer step (this rank)

    @property
    def bucket(self):
        return self.opt.bucket if self.fused else (self.buf.bucket if self.buf is not None else None)

    def fold(self):
        """Side-stream slots -> the flat bucket (also called at the end of an epoch so that nothing is left in flight)."""
        if self.pipe is not None and any(self.pipe._used):
            self.opt.flat_g.add_(self.pipe.reduce(all_reduce=False))

    def boundary(self, last_bag_ran):
        """A window's last loader position has been passed.  The reference steps there only when that bag was not
        skipped (its `continue` jumps over the step, so the gradients stay and the window merges with the next one).
        world > 1: ONE all-reduce of [grads | ran-flag | kept-count]; every rank takes the same decision from the
        reduced control words (`last_bag_ran` is known to the rank that owns the window's last position only)."""
        self.fold()
        ran, kept = bool(last_bag_ran), self.kept
        if self.world > 1:
            tail = self.bucket[-2:]
            tail[0] += 1.0 if last_bag_ran else 0.0
            tail[1] += float(self.kept)
            (self.opt if self.fused else self.buf).all_reduce()
            flag, total = self.bucket[-2:].tolist()         # host sync, once per optimizer step
            ran, kept = flag > 0.5, int(round(total))
            if not ran:
                # no step: the window stays open.  The reduced sum now sits on every rank; keep it on rank 0 only so
                # that the next all-reduce counts it once.
                if torch.distributed.get_rank() != 0:
                    self.bucket.zero_()
                self.kept = 0
                if self.pipe is not None:
                    self.pipe.release()
                return False
        if not ran:
            return False
        if self.fused:
            self.opt.step(l1_micro_batches=kept)
            self.opt.zero_grad()
        else:
            self.opt.step()
            if self.buf is not None:
                self.buf.zero()
            else:
                self.opt.zero_grad()
        self.kept = 0
        if self.pipe is not None:
            self.pipe.release()
        return True


def _window_of(model, optimizer, world, grad_buffer, inflight, device):
    key = (world, inflight, id(model))
    w = getattr(optimizer, "_mmf_window", None)
    if w is None or getattr(w, "key", None) != key:
        w = _Window(model, optimizer, world, grad_buffer, inflight, device)
        w.key = key
        try:
            optimizer._mmf_window = w
        except AttributeError:
            pass
    return w


def _fused_cox_ok(model, loss_fn, feats):
    """One omic batch = one launch (model.cox_step: MaxNet forward + CoxSurvLoss + backward).  Only where that is exactly what
    `model(**feats)` + the stock loss would compute: MaxNet ITSELF with a Cox head, the stock CoxSurvLoss, no hooks."""
    from ..models.model_genomic import MaxNet
    import torch.nn.modules.module as tm
    x = feats.get("genomic_features")
    if type(loss_fn) is not CoxSurvLoss or type(model).forward is not MaxNet.forward or not hasattr(model, "cox_step"):
        return False
    hooked = lambda m: bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None))
    if any(hooked(m) for m in model.modules()):
        return False
    if tm._global_forward_hooks or tm._global_forward_pre_hooks or tm._global_backward_hooks or getattr(tm, "_global_backward_pre_hooks", None):
        return False
    return x is not None and x.dtype == torch.float32 and model.cox_step_ok(x)


def _fused_radio_ok(model, loss_fn, feats):
    """The radiology head's step without an autograd graph (model.nll_step: reduce_dim, then stack + head + loss + backward
    in one call, then reduce_dim's backward).  Only where that is exactly what `model(**feats)` + the stock loss would
    compute: MIL_Attention_fc_surv_radio ITSELF, the stock NLLSurvLoss, fp32 2-D modality bags of one shape on the GPU, no
    hooks, every parameter trainable."""
    from ..models.model_attention_mil_radio import MIL_Attention_fc_surv_radio
    import torch.nn.modules.module as tm
    if type(loss_fn) is not NLLSurvLoss or type(model).forward is not MIL_Attention_fc_surv_radio.forward:
        return False
    if not getattr(model, "mmf_one_call_step", True) or getattr(model.classifier, "out_features", 1 << 30) > 32:
        return False
    bags = [feats.get(m) for m in model.modalities]
    if any(not (torch.is_tensor(b) and b.is_cuda and b.dim() == 2 and b.dtype == torch.float32) for b in bags):
        return False
    if any(b.shape != bags[0].shape for b in bags):
        return False
    hooked = lambda m: bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None))
    if any(hooked(m) for m in model.modules()):
        return False
    if tm._global_forward_hooks or tm._global_forward_pre_hooks or tm._global_backward_hooks or getattr(tm, "_global_backward_pre_hooks", None):
        return False
    return all(p.requires_grad for p in model.parameters())


def _fused_mm_ok(model, loss_fn, feats):
    """One patient = one fixed sequence of C-ABI calls without an autograd graph (model.nll_step of the multimodal concat
    head).  Only where that is exactly what `model(**feats)` + the stock loss would compute: MM_MIL_Attention_fc_surv ITSELF
    (concat fusion, or the tensor fusion as the heads configure it), the stock NLLSurvLoss, no hooks, every parameter trainable, inputs on the GPU."""
    from ..models.model_mm_attention_mil import MM_MIL_Attention_fc_surv
    import torch.nn.modules.module as tm
    if type(loss_fn) is not NLLSurvLoss or type(model).forward is not MM_MIL_Attention_fc_surv.forward:
        return False
    if not getattr(model, "mmf_one_call_step", True):
        return False
    fusion = getattr(model, "fusion", None)
    if fusion == "concat":
        head = model.classifier
    elif fusion == "tensor":
        head = model.classifier[3]
        if not (model.mm.skip and len(model._concat_order()) * model.mm.reduce[0][0][0].weight.shape[0] <= 384):
            return False
    else:
        return False
    if head.out_features > 32:
        return False
    for v in feats.values():
        if not (torch.is_tensor(v) and v.is_cuda):
            return False
    hooked = lambda m: bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None))
    if any(hooked(m) for m in model.modules()):
        return False
    if tm._global_forward_hooks or tm._global_forward_pre_hooks or tm._global_backward_hooks or getattr(tm, "_global_backward_pre_hooks", None):
        return False
    return all(p.requires_grad for p in model.parameters())


def train_loop_survival(epoch, model, loader, optimizer, n_classes, mode, writer=None, loss_fn=None, reg_fn=None,
                        lambda_reg=0., gc=16, t_bin=None, dp=False, grad_buffer=None, inflight=1):
    """utils/core_utils.py:173-264: same per-bag order (forward, loss, regulariser added AFTER the /gc division,
    backward), and the same window rule -- the optimizer steps after loader position b when (b + 1) % gc == 0 and bag b
    was not skipped; skipped bags (missing modality) contribute nothing but still occupy their position.

    Extras, all off by default:
      dp        one bag per rank (torch.distributed initialised): rank r takes loader positions r, r + world, ...
                (feed.RankShard: only those bags are loaded); the window is gc x world positions and ends with ONE
                all-reduce (SUM) of the flat gradient bucket -- the reference's `--gc gc*world`, see dp.py.  Every rank
                issues exactly one collective per window boundary, whatever it skipped;
      FlatAdam  fused L1 + Adam tail (optim.py); reg_fn must then be l1_reg_all, or l1_reg_modules with a FlatAdam
                built with the matching `l1_modules`;
      inflight  > 1 (needs FlatAdam): the window's bags run round-robin on that many HIP streams (pipeline.py)."""
    from ..feed import RankShard
    from .utils import l1_reg_all, l1_reg_modules
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model.train()
    world, rank = 1, 0
    if dp and torch.distributed.is_available() and torch.distributed.is_initialized():
        world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
    win = _window_of(model, optimizer, world, grad_buffer, inflight, device)
    fused_tail, pipe = win.fused, win.pipe
    if fused_tail:
        if reg_fn is not None and lambda_reg:
            want_mask = reg_fn is l1_reg_modules
            if reg_fn not in (l1_reg_all, l1_reg_modules) or want_mask != (optimizer.l1_mask is not None):
                raise ValueError("FlatAdam applies the L1 term inside its kernel: reg_fn must be l1_reg_all, or "
                                 "l1_reg_modules with FlatAdam(l1_modules=[model.fc_omic, model.mm])")
        optimizer.lambda_l1 = lambda_reg if reg_fn is not None else 0.0      # before the first l1_value() / step
    G = gc * world
    shard = RankShard(loader, rank, world) if world > 1 else None
    n_total = shard.n_total if shard is not None else None
    losses, regs, all_risk, all_c, all_t = [], [], [], [], []
    n_pos = 0
    for i, batch in enumerate(shard if shard is not None else loader):
        radio_features, path_features, genomic_features, label, event_time, c = batch
        pos = shard.position(i) if shard is not None else i
        n_pos += 1
        skipped = _skip(mode, radio_features, path_features, genomic_features)
        if not skipped:
            if isinstance(loss_fn, NLLSurvLoss) and torch.is_tensor(label) and not label.is_cuda and label.numel() \
                    and (int(label.min()) < 0 or int(label.max()) >= n_classes):
                # the reference's gather (utils/loss_utils.py:30-33) raises on such a label; on the device the kernels
                # would write a NaN loss instead, which only shows in the epoch mean -- so check while it is on the host
                raise IndexError(f"survival bin label {label.tolist()} outside [0, {n_classes})")
            feats, label, c = _to_device(radio_features, path_features, genomic_features, label, c, device)

            def forward_loss():
                hazards, S, Y_hat, _ = model(**feats)
                if isinstance(loss_fn, CoxSurvLoss):
                    return hazards, loss_fn(risks=hazards, times=torch.as_tensor(np.asarray(event_time)), c=c)
                if isinstance(loss_fn, NLLSurvLoss):
                    return -torch.sum(S, dim=1), loss_fn(hazards=hazards, S=S, Y=label, c=c)
                raise NotImplementedError(type(loss_fn))

            fused_step = _fused_step_ok(model, loss_fn, feats)
            fused_cox = (not fused_step) and pipe is None and _fused_cox_ok(model, loss_fn, feats)
            fused_mm = (not fused_step) and (not fused_cox) and pipe is None and _fused_mm_ok(model, loss_fn, feats)
            fused_radio = (not fused_step) and (not fused_cox) and (not fused_mm) and pipe is None \
                and _fused_radio_ok(model, loss_fn, feats)
            if fused_radio:
                _, _, _, _, loss, risk = model.nll_step(label, c, alpha=loss_fn.alpha, loss_scale=1.0 / G, **feats)
                fused_step = True
            elif fused_mm:
                # the multimodal concat head: branches, one head + loss launch, branch backwards -- no autograd graph; the
                # gradient of loss / G is already in .grad
                _, _, _, _, loss, risk = model.nll_step(label, c, alpha=loss_fn.alpha, loss_scale=1.0 / G, **feats)
                fused_step = True
            elif fused_cox:
                # the omic batch: MaxNet forward + Cox + backward in one launch; the gradient of loss / G is already in .grad
                risk, loss = model.cox_step(feats["genomic_features"], event_time, c, loss_scale=1.0 / G)
                fused_step = True
            elif pipe is not None and fused_step:
                _, _, _, _, loss, risk = pipe.run_fused(model, feats["path_features"], label, c, loss_fn.alpha,
                                                        loss_scale=1.0 / G)
            elif pipe is not None:
                box = {}

                def bag():
                    box["risk"], box["loss"] = forward_loss()
                    return box["loss"] / G

                pipe.run(bag, inputs=list(feats.values()) + [label, c])
                risk, loss = box["risk"], box["loss"]
            elif fused_step:
                # forward + nll_surv + backward of the bag in one call; the gradient of loss / G is already in .grad
                _, _, _, _, loss, risk = model.nll_step(feats["path_features"], label, c, alpha=loss_fn.alpha,
                                                        loss_scale=1.0 / G)
            else:
                risk, loss = forward_loss()
            if fused_tail:
                # the L1 term never enters autograd: its gradient (lambda * sign(W) per kept bag) is added inside the
                # Adam kernel, its value is a device scalar for logging only
                loss_reg = optimizer.l1_value() if (reg_fn is not None and lambda_reg) else 0
            else:
                loss_reg = 0 if reg_fn is None else reg_fn(model) * lambda_reg
            losses.append(loss.detach())
            regs.append(loss_reg.detach() if torch.is_tensor(loss_reg) else torch.tensor(float(loss_reg), device=device))
            all_risk.append(risk.detach().reshape(-1))
            all_c.append(c.detach().reshape(-1))
            all_t.append(np.asarray(event_time).reshape(-1))
            # the reference: loss = loss / gc + loss_reg ; backward (core_utils.py:242-243)
            if fused_step:
                if not fused_tail and torch.is_tensor(loss_reg) and loss_reg.requires_grad:
                    loss_reg.backward()          # the autograd L1 term touches parameters only
            elif pipe is None:
                (loss / G if fused_tail else loss / G + loss_reg).backward()
            win.kept += 1
        # window boundary: the last position of this rank's window is `last`; (last + 1) % G == 0 as the reference's
        # (batch_idx + 1) % gc == 0.  With world > 1 `last` belongs to rank world - 1 and must exist in the loader.
        last = pos + (world - 1 - rank)
        if (last + 1) % G == 0 and (n_total is None or last < n_total):
            win.boundary(last_bag_ran=(not skipped) if rank == world - 1 else False)
    if pipe is not None:
        pipe.join()                  # the epoch's statistics below read tensors produced on the side streams
        win.fold()                   # a trailing partial window stays accumulated in the bucket, as in the reference
    n = max(n_pos, 1)                # the reference divides by len(loader), skipped positions included (:250-251)
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
