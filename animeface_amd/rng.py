"""Single gateway for every random draw of the training loop.

By default draws are made on the tensor's device with torch's generator, exactly where and in the order the
reference makes them (``torch.randn`` in ``InjectNoise``, ``torch.rand`` / ``torch.randint`` in DiffAugment,
``empty().normal_()`` in ``sample_nnoise``).  ``cpu_stream()`` switches the source to torch's CPU generator
(values are then copied to the device): the stream is then bit-identical to a CPU run of the reference with the same
``torch.manual_seed``, which is how the parity tests replay the reference's own ``train()``."""
import contextlib

import torch

_cpu = False


@contextlib.contextmanager
def cpu_stream(enabled=True):
    global _cpu
    old = _cpu
    _cpu = enabled
    try:
        yield
    finally:
        _cpu = old


def _dev(device):
    return torch.device('cpu') if _cpu else device


def randn(size, device, dtype=torch.float32):
    return torch.randn(size, device=_dev(device), dtype=dtype).to(device)


def rand(size, device, dtype=torch.float32):
    return torch.rand(size, device=_dev(device), dtype=dtype).to(device)


def randint(low, high, size, device):
    return torch.randint(low, high, size=size, device=_dev(device)).to(device)


def normal(size, device, mean=0., std=1.):
    return torch.empty(size, device=_dev(device)).normal_(mean, std).to(device)


def uniform(size, device, start=0., end=1.):
    return torch.empty(size, device=_dev(device)).uniform_(start, end).to(device)
