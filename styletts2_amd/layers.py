"""Parameter holders whose state_dict() layout is key-for-key the reference's
(tests/test_state_dict_layout.py checks this against manifests generated from the reference).

They carry no arithmetic: the engines in decoder.py / diffusion.py fold and pack these tensors once
per load and then issue HIP kernels.  Old-style weight-norm pairs are kept as `weight_g`/`weight_v`
so that reference checkpoints load unchanged ("Checkpoint").
"""
import math

import torch
import torch.nn as nn

from . import weights as W


class WNConv1d(nn.Module):
    """weight_norm(nn.Conv1d(...)) holder: weight_g [C_out,1,1], weight_v [C_out,C_in,ks], bias."""

    def __init__(self, c_in, c_out, ks, bias=True):
        super().__init__()
        v = torch.randn(c_out, c_in, ks) * 0.01
        self.weight_g = nn.Parameter(v.reshape(c_out, -1).norm(dim=1).reshape(c_out, 1, 1))
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.zeros(c_out)) if bias else None
        self.c_in, self.c_out, self.ks = c_in, c_out, ks

    def folded(self):
        # always on the host in fp32: the C++ engine takes host-folded weights (st2_load_weights) and a device-side
        # norm differs from ATen-CPU's in the last ulp, which would make the two plans differ bitwise
        return W.fold_weight_norm(self.weight_g.detach().float().cpu(), self.weight_v.detach().float().cpu())


class WNConvTranspose1d(nn.Module):
    """weight_norm(nn.ConvTranspose1d(...)) holder: weight_g [C_in,1,1], weight_v [C_in,C_out/groups,K]."""

    def __init__(self, c_in, c_out_per_group, ks, bias_dim):
        super().__init__()
        v = torch.randn(c_in, c_out_per_group, ks) * 0.01
        self.weight_g = nn.Parameter(v.reshape(c_in, -1).norm(dim=1).reshape(c_in, 1, 1))
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.zeros(bias_dim))

    def folded(self):
        # always on the host in fp32: the C++ engine takes host-folded weights (st2_load_weights) and a device-side
        # norm differs from ATen-CPU's in the last ulp, which would make the two plans differ bitwise
        return W.fold_weight_norm(self.weight_g.detach().float().cpu(), self.weight_v.detach().float().cpu())


class PlainConv1d(nn.Module):
    def __init__(self, c_in, c_out, ks, bias=True):
        super().__init__()
        bound = 1.0 / math.sqrt(c_in * ks)
        self.weight = nn.Parameter((torch.rand(c_out, c_in, ks) * 2 - 1) * bound)
        self.bias = nn.Parameter((torch.rand(c_out) * 2 - 1) * bound) if bias else None


class PlainLinear(nn.Module):
    def __init__(self, f_in, f_out, bias=True):
        super().__init__()
        bound = 1.0 / math.sqrt(f_in)
        self.weight = nn.Parameter((torch.rand(f_out, f_in) * 2 - 1) * bound)
        self.bias = nn.Parameter((torch.rand(f_out) * 2 - 1) * bound) if bias else None


class AdaINParams(nn.Module):
    """AdaIN1d (Modules/istftnet.py:15-25): only `fc` has state (InstanceNorm1d is affine=False)."""

    def __init__(self, style_dim, channels):
        super().__init__()
        self.fc = PlainLinear(style_dim, 2 * channels)
        self.channels = channels


class AdaINResBlock1Params(nn.Module):
    """AdaINResBlock1 (Modules/istftnet.py:27-81)."""

    def __init__(self, channels, ks, dilation, style_dim):
        super().__init__()
        self.convs1 = nn.ModuleList([WNConv1d(channels, channels, ks) for _ in dilation])
        self.convs2 = nn.ModuleList([WNConv1d(channels, channels, ks) for _ in dilation])
        self.adain1 = nn.ModuleList([AdaINParams(style_dim, channels) for _ in dilation])
        self.adain2 = nn.ModuleList([AdaINParams(style_dim, channels) for _ in dilation])
        self.alpha1 = nn.ParameterList([nn.Parameter(torch.ones(1, channels, 1)) for _ in dilation])
        self.alpha2 = nn.ParameterList([nn.Parameter(torch.ones(1, channels, 1)) for _ in dilation])
        self.channels, self.ks, self.dilation = channels, ks, tuple(dilation)


class AdainResBlk1dParams(nn.Module):
    """AdainResBlk1d (Modules/istftnet.py:410-454; same class in models.py:372-416)."""

    def __init__(self, dim_in, dim_out, style_dim, upsample=False):
        super().__init__()
        self.conv1 = WNConv1d(dim_in, dim_out, 3)
        self.conv2 = WNConv1d(dim_out, dim_out, 3)
        self.norm1 = AdaINParams(style_dim, dim_in)
        self.norm2 = AdaINParams(style_dim, dim_out)
        if dim_in != dim_out:
            self.conv1x1 = WNConv1d(dim_in, dim_out, 1, bias=False)
        if upsample:
            self.pool = WNConvTranspose1d(dim_in, 1, 3, dim_in)
        self.dim_in, self.dim_out, self.upsample, self.learned_sc = dim_in, dim_out, upsample, dim_in != dim_out


# ---- per-process caches never travel -------------------------------------------------------------------------------
# Modules keep lazily built, process-local state next to their parameters: packed weights (`_pk`), st2_engine handles
# (ctypes pointers: `_eng`, `_engine`, `_front_engine`, `_style_engine`) and a weakref to the module that owns their C++
# plan (`_owner`).  None of it can be pickled or deep-copied, and none of it should be: a copy repacks on first use.
TRANSIENT_ATTRS = ("_pk", "_eng", "_engine", "_engine_stale", "_front_engine", "_style_engine", "_owner")


def transient_state(cls):
    """Class decorator: `torch.save(module)`, `pickle` and `copy.deepcopy` see the transient attributes as None."""
    inherited = getattr(cls, "__getstate__", None)  # e.g. nn.RNNBase drops its own weakrefs there; object has none (3.10)

    def __getstate__(self):
        d = dict(inherited(self)) if inherited is not None and inherited is not getattr(object, "__getstate__", None) \
            else self.__dict__.copy()
        for k in TRANSIENT_ATTRS:
            if k in d:
                d[k] = None
        return d
    cls.__getstate__ = __getstate__
    return cls

