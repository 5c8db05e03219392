"""PopArt value normaliser (reference safepo/common/popart.py:45-133): debiased running mean / mean-of-squares with
decay beta, variance clamped at 1e-2.  Tiny per-update tensor arithmetic on the value dimension (1 element for the
MAPPO critics): kept in PyTorch like the Lagrange multiplier; the GAE kernel consumes (sqrt(var), mean) as scalars."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class PopArt(nn.Module):
    def __init__(self, input_shape, norm_axes=1, beta=0.99999, per_element_update=False, epsilon=1e-5,
                 device=torch.device("cpu")):
        super().__init__()
        self.input_shape, self.norm_axes, self.epsilon, self.beta = input_shape, norm_axes, epsilon, beta
        self.per_element_update = per_element_update
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.running_mean = nn.Parameter(torch.zeros(input_shape), requires_grad=False).to(**self.tpdv)
        self.running_mean_sq = nn.Parameter(torch.zeros(input_shape), requires_grad=False).to(**self.tpdv)
        self.debiasing_term = nn.Parameter(torch.tensor(0.0), requires_grad=False).to(**self.tpdv)

    def reset_parameters(self):
        self.running_mean.zero_()
        self.running_mean_sq.zero_()
        self.debiasing_term.zero_()

    def running_mean_var(self):
        d = self.debiasing_term.clamp(min=self.epsilon)
        mean = self.running_mean / d
        var = (self.running_mean_sq / d - mean ** 2).clamp(min=1e-2)
        return mean, var

    def forward(self, input_vector, train=True):
        if type(input_vector) == np.ndarray:
            input_vector = torch.from_numpy(input_vector)
        input_vector = input_vector.to(**self.tpdv)
        if train:
            x = input_vector.detach()
            axes = tuple(range(self.norm_axes))
            batch_mean, batch_sq_mean = x.mean(dim=axes), (x ** 2).mean(dim=axes)
            weight = self.beta ** np.prod(x.size()[:self.norm_axes]) if self.per_element_update else self.beta
            self.running_mean.mul_(weight).add_(batch_mean * (1.0 - weight))
            self.running_mean_sq.mul_(weight).add_(batch_sq_mean * (1.0 - weight))
            self.debiasing_term.mul_(weight).add_(1.0 * (1.0 - weight))
        mean, var = self.running_mean_var()
        return (input_vector - mean[(None,) * self.norm_axes]) / torch.sqrt(var)[(None,) * self.norm_axes]

    def denormalize(self, input_vector):
        if type(input_vector) == np.ndarray:
            input_vector = torch.from_numpy(input_vector)
        input_vector = input_vector.to(**self.tpdv)
        mean, var = self.running_mean_var()
        return (input_vector * torch.sqrt(var)[(None,) * self.norm_axes] + mean[(None,) * self.norm_axes]).detach()

    def denorm_scalars(self):
        """(sqrt(var), mean) as python floats for spo_ma_gae (value dimension 1)."""
        mean, var = self.running_mean_var()
        return float(torch.sqrt(var).reshape(-1)[0]), float(mean.reshape(-1)[0])
