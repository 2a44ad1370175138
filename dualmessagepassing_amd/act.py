"""Activation registry and weight initialisers with the reference's names and
numerics (SubgraphCountingMatching/utils/act.py:27,457-489 and utils/init.py:17-143).

Needed on the hot path only so that the drop-in modules build the same
sub-modules (``nmlp.1`` is the *same* activation object for every layer, as in
the reference) and initialise parameters from the same distribution.
"""
import math

import torch as th
import torch.nn as nn

from .constants import LEAKY_RELU_A


class Identity(nn.Module):
    def forward(self, x):
        return x


# utils/act.py:457-474 -- module-level singletons, shared by every layer that asks for them
supported_act_funcs = {
    "none": Identity(),
    "softmax": nn.Softmax(dim=-1),
    "sigmoid": nn.Sigmoid(),
    "tanh": nn.Tanh(),
    "relu": nn.ReLU(),
    "relu6": nn.ReLU6(),
    "leaky_relu": nn.LeakyReLU(negative_slope=LEAKY_RELU_A),
    "prelu": nn.PReLU(init=LEAKY_RELU_A),
    "elu": nn.ELU(),
    "celu": nn.CELU(),
    "selu": nn.SELU(),
    "gelu": nn.GELU(),
}


def map_activation_str_to_layer(act_func, **kw):
    # utils/act.py:477-489
    if act_func not in supported_act_funcs:
        raise NotImplementedError(act_func)
    act = supported_act_funcs[act_func]
    for k, v in kw.items():
        if hasattr(act, k):
            try:
                setattr(act, k, v)
            except Exception:
                pass
    return act


# activation name -> the nonlinearity whose gain torch's table holds (utils/init.py:17-50 maps names the same way:
# the rectifier family shares "relu", the parametrised rectifiers "leaky_relu", the normalising ones "sigmoid")
_GAIN_OF = dict.fromkeys(("none", "maximum", "minimum"), "linear")
_GAIN_OF.update(dict.fromkeys(("relu", "relu6", "elu", "selu", "celu", "gelu"), "relu"))
_GAIN_OF.update(dict.fromkeys(("leaky_relu", "prelu"), "leaky_relu"))
_GAIN_OF.update(dict.fromkeys(("softmax", "sparsemax", "gumbel_softmax", "sigmoid"), "sigmoid"))
_GAIN_OF["tanh"] = "tanh"


def calculate_gain(activation):
    if not isinstance(activation, str):
        raise ValueError(activation)
    kind = _GAIN_OF.get(activation)
    if kind is None:
        raise NotImplementedError(activation)
    return nn.init.calculate_gain(kind, LEAKY_RELU_A)


def _fans(x):
    """(fan_in, fan_out) as utils/init.py:53-64 counts them: size(1) and size(0) times the receptive field."""
    if x.dim() < 2:
        x = x.unsqueeze(-1)
    field = x[0][0].numel() if x.dim() > 2 else 1
    return x.size(1) * field, x.size(0) * field


def _glorot_bound(x, gain):
    """Xavier / Glorot uniform bound sqrt(3) * std with std = gain * sqrt(2 / (fan_in + fan_out)), evaluated in the reference's
    order of operations (utils/init.py:71-76) so that seeded draws agree to the bit."""
    std = gain * math.sqrt(2.0 / float(sum(_fans(x))))
    return 1.7320508075688772 * std


# weight initialisers by name (utils/init.py:67-99,125-143): each takes the tensor and the activation's gain
_INITS = {
    "zero": lambda x, gain: nn.init.zeros_(x),
    "uniform": lambda x, gain: nn.init.uniform_(x, -_glorot_bound(x, gain), _glorot_bound(x, gain)),
    "normal": lambda x, gain: nn.init.normal_(x, 0, gain / math.sqrt(_fans(x)[0])),
    "orthogonal": lambda x, gain: nn.init.orthogonal_(x, gain=1.0),      # the reference ignores the gain here
}


def xavier_uniform_init(x, gain=1.0):
    return _INITS["uniform"](x, gain)


def kaiming_normal_init(x, gain=1.0):
    return _INITS["normal"](x, gain)


def zero_init(x, gain=1.0):
    return _INITS["zero"](x, gain)


def orthogonal_init(x, gain=1.0):
    return _INITS["orthogonal"](x, gain)


def _check_init(init):
    if init not in _INITS:
        raise ValueError("init=%s is not supported now." % (init))


def init_weight(x, activation="none", init="uniform"):
    """utils/init.py:125-143."""
    _check_init(init)
    if isinstance(x, th.Tensor):
        _INITS[init](x, calculate_gain(activation))


# embedding tables by init name (utils/init.py:160-178): unit-scale draws, not gain-scaled
_EMBEDDING_INITS = {
    "uniform": lambda w: nn.init.uniform_(w, -1.0, 1.0),
    "normal": lambda w: nn.init.normal_(w, 0.0, 1.0),
    "orthogonal": lambda w: nn.init.orthogonal_(w, gain=math.sqrt(_fans(w)[0])),
}
_WEIGHTED = (nn.Linear, nn.Conv1d, nn.Conv2d, nn.Conv3d)
_NORMS = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.LayerNorm, nn.GroupNorm)


def init_module(x, activation="none", init="uniform"):
    """utils/init.py:146-192 for the module kinds on this path: weighted layers (weight by ``init``, zero bias), embeddings
    (``_EMBEDDING_INITS``, padding row zeroed), normalisation layers (identity affine map)."""
    _check_init(init)
    gain = calculate_gain(activation)
    if isinstance(x, _WEIGHTED):
        _INITS[init](x.weight, gain)
        if getattr(x, "bias", None) is not None:
            nn.init.zeros_(x.bias)
    elif isinstance(x, nn.Embedding):
        with th.no_grad():
            if init in _EMBEDDING_INITS:
                _EMBEDDING_INITS[init](x.weight)
            if x.padding_idx is not None:
                x.weight[x.padding_idx].fill_(0)
    elif isinstance(x, _NORMS):
        nn.init.ones_(x.weight)
        nn.init.zeros_(x.bias)
