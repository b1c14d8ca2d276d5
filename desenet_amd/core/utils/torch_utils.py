"""Model-side helpers that define numerics the kernels must honour (reference core/utils/torch_utils.py):
initialize_weights (:160-168: BN eps 1e-3, momentum 0.03), fuse_conv_and_bn (:196-216), model_info (:219-240, without the
thop FLOP probe), intersect_dicts (:151-157), de_parallel (:47-51).  Host-side, one-off parameter algebra (plain torch)."""
from __future__ import annotations

import logging

import torch
import torch.nn as nn

LOGGER = logging.getLogger(__name__)


def initialize_weights(model):
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps = 1e-3
            m.momentum = 0.03
        elif isinstance(m, (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6)):
            m.inplace = True


def fuse_conv_and_bn(conv, bn):
    """W' = diag(g / sqrt(var + eps)) W,  b' = b - g * mean / sqrt(var + eps) (+ scaled conv bias)."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    with torch.no_grad():
        scale = bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
        fused.weight.copy_(torch.mm(torch.diag(scale), conv.weight.view(conv.out_channels, -1)).view(fused.weight.shape))
        b_conv = torch.zeros(conv.weight.size(0), device=conv.weight.device) if conv.bias is None else conv.bias
        b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
        fused.bias.copy_(torch.mm(torch.diag(scale), b_conv.reshape(-1, 1)).reshape(-1) + b_bn)
    return fused


def intersect_dicts(da, db, exclude=()):
    return {k: v for k, v in da.items() if k in db and not any(x in k for x in exclude) and v.shape == db[k].shape}


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def de_parallel(model):
    return model.module if is_parallel(model) else model


def model_info(model, verbose=False, img_size=640):
    n_p = sum(x.numel() for x in model.parameters())
    n_g = sum(x.numel() for x in model.parameters() if x.requires_grad)
    if verbose:
        for i, (name, p) in enumerate(model.named_parameters()):
            LOGGER.info("%5g %40s %9s %12g %20s" % (i, name.replace("module_list.", ""), p.requires_grad, p.numel(),
                                                     list(p.shape)))
    LOGGER.info(f"Model Summary: {len(list(model.modules()))} layers, {n_p} parameters, {n_g} gradients")
