"""reference implementations/ADA/model.py: the StyleGAN3 networks plus the ADA pipe with the (interval, target_kimg, threshold,
batch_size) argument order of that file (:6-9)."""
from ..StyleGAN3.model import Generator, Discriminator  # noqa: F401
from ...nnutils.ada import ADA as _ADA


class ADA(_ADA):
    def __init__(self, interval, target_kimg, threshold, batch_size, **augment_kwargs):
        super().__init__(batch_size, interval, target_kimg, threshold, **augment_kwargs)
