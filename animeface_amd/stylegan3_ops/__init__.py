"""Operator surface of the reference's ``thirdparty/stylegan3_ops/ops`` on MI355X kernels:
``upfirdn2d``, ``bias_act``, ``filtered_lrelu``, ``conv2d_resample``, ``conv2d_gradfix``; plus ``layout`` (planar <->
channels-last with padding, the glue between the FIR kernels and the MFMA conv)."""
from . import upfirdn2d, bias_act, filtered_lrelu, conv2d_gradfix, conv2d_resample, layout  # noqa: F401
