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


def calculate_gain(activation):
    # utils/init.py:17-50
    if isinstance(activation, str):
        if activation in ["none", "maximum", "minimum"]:
            nonlinearity = "linear"
        elif activation in ["relu", "relu6", "elu", "selu", "celu", "gelu"]:
            nonlinearity = "relu"
        elif activation in ["leaky_relu", "prelu"]:
            nonlinearity = "leaky_relu"
        elif activation in ["softmax", "sparsemax", "gumbel_softmax"]:
            nonlinearity = "sigmoid"
        elif activation in ["sigmoid", "tanh"]:
            nonlinearity = activation
        else:
            raise NotImplementedError(activation)
    else:
        raise ValueError(activation)
    return nn.init.calculate_gain(nonlinearity, LEAKY_RELU_A)


def _fans(x):
    # utils/init.py:53-64 (note: "fan_in" is size(1) -- symmetric in the formula below)
    if x.dim() < 2:
        x = x.unsqueeze(-1)
    rf = 1
    if x.dim() > 2:
        rf = x[0][0].numel()
    return x.size(1) * rf, x.size(0) * rf


def xavier_uniform_init(x, gain=1.0):
    # utils/init.py:71-76
    fan_in, fan_out = _fans(x)
    std = gain * math.sqrt(2.0 / float(fan_in + fan_out))
    a = 1.7320508075688772 * std
    return nn.init.uniform_(x, -a, a)


def kaiming_normal_init(x, gain=1.0):
    fan_in, _ = _fans(x)
    return nn.init.normal_(x, 0, gain / math.sqrt(fan_in))


def zero_init(x, gain=1.0):
    return nn.init.zeros_(x)


def orthogonal_init(x, gain=1.0):
    return nn.init.orthogonal_(x, gain=1.0)


_INITS = {"zero": zero_init, "uniform": xavier_uniform_init, "normal": kaiming_normal_init,
          "orthogonal": orthogonal_init}


def init_weight(x, activation="none", init="uniform"):
    # utils/init.py:125-143
    if init not in _INITS:
        raise ValueError("init=%s is not supported now." % (init))
    if isinstance(x, th.Tensor):
        _INITS[init](x, gain=calculate_gain(activation))


def init_module(x, activation="none", init="uniform"):
    # utils/init.py:146-192 (Linear / norm branches; the only module kinds on this path)
    if init not in _INITS:
        raise ValueError("init=%s is not supported now." % (init))
    gain = calculate_gain(activation)
    if isinstance(x, (nn.Linear, nn.Conv1d, nn.Conv2d, nn.Conv3d)):
        _INITS[init](x.weight, gain=gain)
        if getattr(x, "bias", None) is not None:
            nn.init.zeros_(x.bias)
    elif isinstance(x, nn.Embedding):
        with th.no_grad():
            if init == "uniform":
                nn.init.uniform_(x.weight, -1.0, 1.0)
            elif init == "normal":
                nn.init.normal_(x.weight, 0.0, 1.0)
            elif init == "orthogonal":
                nn.init.orthogonal_(x.weight, gain=math.sqrt(_fans(x.weight)[0]) * 1.0)
            if x.padding_idx is not None:
                x.weight[x.padding_idx].fill_(0)
    elif isinstance(x, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.LayerNorm, nn.GroupNorm)):
        nn.init.ones_(x.weight)
        nn.init.zeros_(x.bias)
