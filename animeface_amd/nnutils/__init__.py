"""Trainer helpers with the reference's names and signatures (reference ``nnutils/``)."""
import torch

from .training import sample_nnoise, sample_unoise, update_ema  # noqa: F401


def get_device(gpu=True):
    """reference nnutils/__init__.py:18-21 pins cuda:0; here the rank's own GPU (LOCAL_RANK) is used."""
    import os
    if gpu and torch.cuda.is_available():
        return torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    return torch.device('cpu')


def freeze(model):
    model.eval()
    for p in model.parameters():
        p.requires_grad = False


def unfreeze(model):
    model.train()
    for p in model.parameters():
        p.requires_grad = True
