"""reference nnutils/training.py:7-40 -- same signatures; ``update_ema`` runs as one multi-tensor
(foreach) pass instead of ~200 per-parameter kernel pairs."""
from __future__ import annotations

from typing import Union

import torch

from .. import rng


def sample_nnoise(size, device: Union[torch.device, str], mean: float = 0., std: float = 1.) -> torch.Tensor:
    return rng.normal(size, device, mean, std)


def sample_unoise(size, device: Union[torch.device, str], start: float = 0., end: float = 1.) -> torch.Tensor:
    return rng.uniform(size, device, start, end)


@torch.no_grad()
def update_ema(model: torch.nn.Module, model_ema: torch.nn.Module, decay: float = 0.999, copy_buffers: bool = False) -> None:
    model.eval()
    param_ema = dict(model_ema.named_parameters())
    param = dict(model.named_parameters())
    keys = list(param_ema.keys())
    ema_list = [param_ema[k].data for k in keys]
    src_list = [param[k].data for k in keys]
    if decay == 0:
        # plain copy: ``mul_(0)`` would keep NaNs of a freshly constructed (torch.empty) model_ema
        torch._foreach_copy_(ema_list, src_list)
    else:
        # ema * decay + p * (1 - decay) as ONE multi-tensor pass: ema + (1 - decay) * (p - ema)  (the reference's mul_ / add_ pair reads and
        # writes every EMA tensor twice; the two forms differ by an fp32 rounding)
        torch._foreach_lerp_(ema_list, src_list, 1 - decay)
    if copy_buffers:
        buffer_ema = dict(model_ema.named_buffers())
        buffer = dict(model.named_buffers())
        keys = [k for k in buffer_ema.keys() if buffer_ema[k].data_ptr() != buffer[k].data_ptr()]
        same = [k for k in keys if buffer_ema[k].dtype == buffer[k].dtype and buffer_ema[k].device == buffer[k].device]
        if same:                                                    # one multi-tensor pass (the StyleGAN3 generator has 48 buffers)
            torch._foreach_copy_([buffer_ema[k].data for k in same], [buffer[k].data for k in same])
        for key in keys:
            if key not in same:
                buffer_ema[key].data.copy_(buffer[key].data)
    model.train()
