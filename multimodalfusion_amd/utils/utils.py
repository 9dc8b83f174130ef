"""Host-side helpers with the reference's names and semantics (utils/utils.py in the reference)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.optim as optim


def initialize_weights(module):
    """xavier-normal weights, zero biases, in module.modules() order (reference utils/utils.py:217-226)."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight)
            m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm1d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def init_max_weights(module):
    """N(0, 1/sqrt(fan_in)) weights, zero biases (reference utils/utils.py:228-233)."""
    for m in module.modules():
        if type(m) == nn.Linear:
            stdv = 1.0 / math.sqrt(m.weight.size(1))
            m.weight.data.normal_(0, stdv)
            m.bias.data.zero_()


def l1_reg_all(model, reg_type=None):
    """sum_W |W|_1 over ALL parameters, biases included (reference utils/utils.py:249-257)."""
    l1_reg = None
    for W in model.parameters():
        s = torch.abs(W).sum()
        l1_reg = s if l1_reg is None else l1_reg + s
    return l1_reg


def get_optim(model, args):
    """Adam / SGD with L2 weight decay = args.reg (reference utils/utils.py:144-151)."""
    params = filter(lambda p: p.requires_grad, model.parameters())
    if args.opt == "adam":
        return optim.Adam(params, lr=args.lr, weight_decay=args.reg)
    if args.opt == "sgd":
        return optim.SGD(params, lr=args.lr, momentum=0.9, weight_decay=args.reg)
    raise NotImplementedError


def print_network(net):
    """reference utils/utils.py:153-165."""
    num_params = 0
    num_params_train = 0
    print(net)
    for param in net.parameters():
        n = param.numel()
        num_params += n
        if param.requires_grad:
            num_params_train += n
    print("Total number of parameters: %d" % num_params)
    print("Total number of trainable parameters: %d" % num_params_train)
