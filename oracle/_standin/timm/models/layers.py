"""Weight-INIT helpers only, so that the read-only reference under /root/reference can be
imported in the build container (timm is not installed).

Written from scratch to the documented timm-0.5.4 signatures.  None of these functions
takes part in forward arithmetic, and every parity check overwrites all parameters through
``load_state_dict`` -- so the distributions produced here do not influence any golden vector.
TEST INFRASTRUCTURE: used only by oracle/make_goldens.py.
"""
import math

import torch


def to_2tuple(x):
    if isinstance(x, (tuple, list)):
        return tuple(x)
    return (x, x)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    with torch.no_grad():
        torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)
    return tensor


def variance_scaling_(tensor, scale=1.0, mode="fan_in", distribution="normal"):
    fan_out, fan_in = tensor.shape[0], int(math.prod(tensor.shape[1:]))
    denom = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2}[mode]
    variance = scale / denom
    with torch.no_grad():
        if distribution == "truncated_normal":
            std = math.sqrt(variance) / 0.87962566103423978
            torch.nn.init.trunc_normal_(tensor, std=std, a=-2 * std, b=2 * std)
        elif distribution == "normal":
            tensor.normal_(std=math.sqrt(variance))
        elif distribution == "uniform":
            bound = math.sqrt(3 * variance)
            tensor.uniform_(-bound, bound)
        else:
            raise ValueError(distribution)
    return tensor


def lecun_normal_(tensor):
    return variance_scaling_(tensor, mode="fan_in", distribution="truncated_normal")
