"""animeface_amd: MI355X (gfx950) native StyleGAN2/3 training hot path.

Host code is PyTorch-ROCm Python; every hot operator is a hand-written HIP kernel in
``libagf_ops.so`` reached through the C ABI declared in ``include/agf_ops.h``.
There is no CPU or eager fallback: calling an op without the library raises.
"""
__version__ = '0.1.0'
