"""Oracle: fused bias + activation + gain + clamp.  TEST INFRASTRUCTURE ONLY.

Restates ``thirdparty/stylegan3_ops/ops/bias_act.py:86-115`` (``_bias_act_ref``)
with the activation table of ``bias_act.py:16-26``; gradients come from plain
autograd, which for the clamp follows the native rule "zero where |y| >= clamp"
(``bias_act.cu:130-136``) up to the measure-zero boundary.
"""
import math
import torch
import torch.nn.functional as F

# name -> (fn, def_alpha, def_gain, native index, saved ref, has_2nd_grad)   bias_act.py:16-26
ACTIVATIONS = {
    'linear':   (lambda x, a: x,                       0.0, 1.0,          1, '',  False),
    'relu':     (lambda x, a: F.relu(x),               0.0, math.sqrt(2), 2, 'y', False),
    'lrelu':    (lambda x, a: F.leaky_relu(x, a),      0.2, math.sqrt(2), 3, 'y', False),
    'tanh':     (lambda x, a: torch.tanh(x),           0.0, 1.0,          4, 'y', True),
    'sigmoid':  (lambda x, a: torch.sigmoid(x),        0.0, 1.0,          5, 'y', True),
    'elu':      (lambda x, a: F.elu(x),                0.0, 1.0,          6, 'y', True),
    'selu':     (lambda x, a: F.selu(x),               0.0, 1.0,          7, 'y', True),
    'softplus': (lambda x, a: F.softplus(x),           0.0, 1.0,          8, 'y', True),
    'swish':    (lambda x, a: torch.sigmoid(x) * x,    0.0, math.sqrt(2), 9, 'x', True),
}


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """Order of operations: +b -> act(alpha) -> *gain -> clamp (bias_act.py:96-115)."""
    fn, def_alpha, def_gain, _idx, _ref, _h2 = ACTIVATIONS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    if b is not None:
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        x = x + b.reshape([-1 if i == dim else 1 for i in range(x.ndim)])
    x = fn(x, alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x
