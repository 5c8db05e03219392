"""Lagrange multiplier (reference safepo/common/lagrange.py:24-105).

Scalar host-side state: one fp32 parameter stepped once per epoch by Adam on
loss = -lambda * (Jc - cost_limit), then projected onto [0, upper_bound].  Kept in PyTorch on the
host exactly like the reference (nil cost, exact parity for free)."""
from __future__ import annotations

import torch


class Lagrange:
    def __init__(self, cost_limit: float, lagrangian_multiplier_init: float, lagrangian_multiplier_lr: float,
                 lagrangian_upper_bound: float | None = None) -> None:
        self.cost_limit = cost_limit
        self.lagrangian_multiplier_lr = lagrangian_multiplier_lr
        self.lagrangian_upper_bound = lagrangian_upper_bound
        start = max(lagrangian_multiplier_init, 0.0)
        self._lagrangian_multiplier = torch.nn.Parameter(torch.as_tensor(start), requires_grad=True)
        self.lambda_range_projection = torch.nn.ReLU()
        self.lambda_optimizer = torch.optim.Adam([self._lagrangian_multiplier], lr=lagrangian_multiplier_lr)

    @property
    def lagrangian_multiplier(self) -> float:
        return self.lambda_range_projection(self._lagrangian_multiplier).detach().item()

    def compute_lambda_loss(self, mean_ep_cost: float) -> torch.Tensor:
        return -self._lagrangian_multiplier * (mean_ep_cost - self.cost_limit)

    def update_lagrange_multiplier(self, Jc: float) -> None:
        self.lambda_optimizer.zero_grad()
        self.compute_lambda_loss(Jc).backward()
        self.lambda_optimizer.step()
        self._lagrangian_multiplier.data.clamp_(0.0, self.lagrangian_upper_bound)
