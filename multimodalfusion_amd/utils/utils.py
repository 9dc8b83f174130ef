"""Host-side helpers under the reference's names (utils/utils.py of the reference): optimizer factory, same-seed
parameter initialisation, the L1 regularisers, the network summary and early stopping.  Nothing here is arithmetic of
the hot path; the contract is names, signatures, printed text and -- for the two initialisers -- the order in which
the global torch RNG is consumed, because `torch.manual_seed(s)` + construction must give the reference's weights
(tests/test_boundary_cpu.py pins that against tests/golden/init.npz)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn


def _reinit(root, draw_weight, exact_type, touch_batchnorm):
    """Visit root.modules() in registration order; every Linear gets `draw_weight(weight)` and a zero bias."""
    for layer in root.modules():
        linear = (type(layer) is nn.Linear) if exact_type else isinstance(layer, nn.Linear)
        with torch.no_grad():
            if linear:
                draw_weight(layer.weight)
                layer.bias.zero_()
            elif touch_batchnorm and isinstance(layer, nn.BatchNorm1d):
                layer.weight.fill_(1.0)
                layer.bias.zero_()


def initialize_weights(module):
    """Xavier-normal Linear weights, zero biases, BatchNorm1d reset to (1, 0) -- reference utils/utils.py:217-226."""
    _reinit(module, nn.init.xavier_normal_, exact_type=False, touch_batchnorm=True)


def init_max_weights(module):
    """Linear weights ~ N(0, 1/sqrt(fan_in)), zero biases; exact nn.Linear only -- reference utils/utils.py:228-233."""
    _reinit(module, lambda w: w.normal_(0.0, 1.0 / math.sqrt(w.size(1))), exact_type=True, touch_batchnorm=False)


def l1_reg_all(model, reg_type=None):
    """||theta||_1 over every parameter of `model`, biases included (reference utils/utils.py:249-257)."""
    terms = [p.abs().sum() for p in model.parameters()]
    if not terms:
        return None
    total = terms[0]
    for t in terms[1:]:
        total = total + t
    return total


def l1_reg_modules(model, reg_type=None):
    """L1 norm of the omic SNN and, when the model has one, of the fusion block `mm` only
    (reference utils/utils.py:259-268)."""
    total = 0
    for name in ("fc_omic", "mm"):
        sub = getattr(model, name, None)
        if sub is None:
            if name == "fc_omic":
                raise AttributeError("l1_reg_modules needs a model with an fc_omic block")
            continue
        total = total + l1_reg_all(sub)
    return total


def get_optim(model, args):
    """args.opt in {'adam', 'sgd'}; L2 weight decay = args.reg (reference utils/utils.py:144-151)."""
    trainable = [p for p in model.parameters() if p.requires_grad]
    makers = {
        "adam": lambda: torch.optim.Adam(trainable, lr=args.lr, weight_decay=args.reg),
        "sgd": lambda: torch.optim.SGD(trainable, lr=args.lr, momentum=0.9, weight_decay=args.reg),
    }
    if args.opt not in makers:
        raise NotImplementedError
    return makers[args.opt]()


def print_network(net):
    """Module tree plus the two parameter counts, in the reference's log format (utils/utils.py:153-165)."""
    print(net)
    sizes = [(p.numel(), p.requires_grad) for p in net.parameters()]
    print('Total number of parameters: %d' % sum(n for n, _ in sizes))
    print('Total number of trainable parameters: %d' % sum(n for n, t in sizes if t))


class EarlyStopping:
    """Validation-loss early stopping with the reference's call surface (utils/utils.py:167-214):
    `stopper(epoch, val_loss, model, ckpt_name=None)`; attributes `early_stop`, `best_ckpt`, `val_loss_min`, `counter`.
    Nothing is tracked before epoch `warmup`; an improvement (val_loss <= best) snapshots the state-dict (and saves it
    when a checkpoint name is given) and resets the counter; otherwise the counter grows, and `early_stop` is raised
    once it reaches `patience` after epoch `stop_epoch`."""

    def __init__(self, warmup=5, patience=15, stop_epoch=50, verbose=False):
        self.warmup, self.patience, self.stop_epoch, self.verbose = warmup, patience, stop_epoch, verbose
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = np.inf
        self.best_ckpt = None

    def __call__(self, epoch, val_loss, model, ckpt_name=None):
        if epoch < self.warmup:
            return
        score = -val_loss
        if self.best_score is not None and score < self.best_score:
            self.counter += 1
            print(f'EarlyStopping counter: {self.counter} out of {self.patience}')
            self.early_stop = self.early_stop or (self.counter >= self.patience and epoch > self.stop_epoch)
            return
        self.best_score = score
        self.counter = 0
        self.save_checkpoint(val_loss, model, ckpt_name)

    def save_checkpoint(self, val_loss, model, ckpt_name):
        if self.verbose:
            print(f'Validation loss decreased ({self.val_loss_min:.6f} --> {val_loss:.6f}).  Saving model ...')
        if ckpt_name is not None:
            torch.save(model.state_dict(), ckpt_name)
        self.best_ckpt = model.state_dict()
        self.val_loss_min = val_loss
