"""Losses and penalties of the StyleGAN2 loop with the reference's names and call signatures
(reference nnutils/loss/gan.py:98-114, nnutils/loss/penalty.py:11-26,85-101).

bf16 training needs no GradScaler; the ``scaler`` arguments are kept so call sites are unchanged and a
``torch.amp.GradScaler`` passed in is honoured exactly as the reference does."""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import grad


class Loss:
    def __init__(self, return_all: bool = False) -> None:
        self.return_all = return_all


class NonSaturatingLoss(Loss):
    def real_loss(self, prob: torch.Tensor) -> torch.Tensor:
        return F.softplus(-prob).mean()

    def fake_loss(self, prob: torch.Tensor) -> torch.Tensor:
        return F.softplus(prob).mean()

    def d_loss(self, real_prob: torch.Tensor, fake_prob: torch.Tensor) -> torch.Tensor:
        rl = self.real_loss(real_prob)
        fl = self.fake_loss(fake_prob)
        loss = rl + fl
        if self.return_all:
            return loss, rl, fl
        return loss

    def g_loss(self, fake_prob: torch.Tensor) -> torch.Tensor:
        return self.real_loss(fake_prob)


def calc_grad(outputs: torch.Tensor, inputs: torch.Tensor, scaler=None) -> torch.Tensor:
    with torch.autocast('cuda', enabled=False):
        if scaler is not None:
            outputs = scaler.scale(outputs)
        ones = torch.ones(outputs.size(), device=outputs.device, dtype=outputs.dtype)
        gradients = grad(outputs=outputs, inputs=inputs, grad_outputs=ones,
                         create_graph=True, retain_graph=True, only_inputs=True)[0]
        if scaler is not None:
            gradients = gradients / scaler.get_scale()
    return gradients


class r1_regularizer(Loss):
    def __call__(self, real: torch.Tensor, D: nn.Module, scaler=None, d_aux_input: tuple = tuple()) -> torch.Tensor:
        real_loc = real.detach().requires_grad_(True)
        d_real_loc = D(real_loc, *d_aux_input)
        gradients = calc_grad(d_real_loc, real_loc, scaler)
        gradients = gradients.reshape(gradients.size(0), -1).float()
        return gradients.norm(2, dim=1).pow(2).mean() / 2.
