from .utils import main  # noqa: F401  (reference implementations/StyleGAN2/__init__.py:1)
