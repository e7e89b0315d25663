"""Parameter holders whose attribute names reproduce the reference's state_dict keys
(SURVEY.md §5 checkpoint contract): `weight`/`bias` for Linear, Conv and LayerNorm,
`running_mean`/`running_var`/`num_batches_tracked` for BatchNorm.  They hold storage only — all
arithmetic is done by the HIP kernels through espresso_amd.functional."""
import math

import torch
import torch.nn as nn


class LinearParams(nn.Module):
    """nn.Linear storage; init = xavier_uniform + zero bias like the reference's `Linear()` helpers
    (fairseq/models/transformer/transformer_legacy.py Linear, used by espresso's encoder for fc0/fc_out)."""

    def __init__(self, in_features, out_features, bias=True, init="xavier", gain=1.0):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        if init == "xavier":
            nn.init.xavier_uniform_(self.weight, gain=gain)
        else:  # torch.nn.Linear default
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
            if bias:
                bound = 1 / math.sqrt(in_features)
                nn.init.uniform_(self.bias, -bound, bound)


class LayerNormParams(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class ConvParams(nn.Module):
    """torch.nn.ConvNd storage with its default (kaiming_uniform a=sqrt(5)) init."""

    def __init__(self, in_ch, out_ch, kernel, groups=1, bias=True):
        super().__init__()
        kernel = tuple(kernel) if isinstance(kernel, (tuple, list)) else (kernel,)
        self.weight = nn.Parameter(torch.empty(out_ch, in_ch // groups, *kernel))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            fan_in = (in_ch // groups) * int(torch.tensor(kernel).prod())
            bound = 1 / math.sqrt(fan_in)
            self.bias = nn.Parameter(torch.empty(out_ch).uniform_(-bound, bound))
        else:
            self.bias = None


class BatchNormParams(nn.Module):
    def __init__(self, ch, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))
        self.register_buffer("running_mean", torch.zeros(ch))
        self.register_buffer("running_var", torch.ones(ch))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
