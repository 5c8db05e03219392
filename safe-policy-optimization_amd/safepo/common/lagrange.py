"""Lagrange multiplier (reference safepo/common/lagrange.py:24-105).

Scalar host-side state: one fp32 parameter stepped once per epoch by Adam on
loss = -lambda * (Jc - cost_limit), then projected onto [0, upper_bound].  Kept in PyTorch on the
host exactly like the reference (nil cost, exact parity for free)."""
from __future__ import annotations

from collections import deque

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


class PIDLagrangian:
    """PID-controlled multiplier (reference safepo/common/lagrange.py:108-200; Stooke et al. 2020).

    Host-side python-float state, updated once per epoch from the mean episode cost:
    integral term clipped at 0 (and at 1 with diff_norm), EMA-smoothed proportional term, EMA-smoothed cost whose
    increase over a `pid_d_delay`-epoch-old value drives the derivative term; penalty = max(0, kp*P + I + kd*D)
    with the reference's normalisation / saturation rules."""

    def __init__(self, cost_limit: float, lagrangian_multiplier_init: float = 0.005, pid_kp: float = 0.1,
                 pid_ki: float = 0.01, pid_kd: float = 0.01, pid_d_delay: int = 10,
                 pid_delta_p_ema_alpha: float = 0.95, pid_delta_d_ema_alpha: float = 0.95, sum_norm: bool = True,
                 diff_norm: bool = False, penalty_max: int = 100.0) -> None:
        self._pid_kp, self._pid_ki, self._pid_kd = pid_kp, pid_ki, pid_kd
        self._pid_d_delay = pid_d_delay
        self._pid_delta_p_ema_alpha, self._pid_delta_d_ema_alpha = pid_delta_p_ema_alpha, pid_delta_d_ema_alpha
        self._penalty_max, self._sum_norm, self._diff_norm = penalty_max, sum_norm, diff_norm
        self._pid_i = lagrangian_multiplier_init
        self._cost_ds = deque(maxlen=pid_d_delay)
        self._cost_ds.append(0.0)
        self._delta_p = 0.0
        self._cost_d = 0.0
        self._cost_limit = cost_limit
        self._cost_penalty = 0.0

    @property
    def lagrangian_multiplier(self) -> float:
        return self._cost_penalty

    def update_lagrange_multiplier(self, ep_cost_avg: float) -> None:
        delta = float(ep_cost_avg - self._cost_limit)
        self._pid_i = max(0.0, self._pid_i + delta * self._pid_ki)
        if self._diff_norm:
            self._pid_i = max(0.0, min(1.0, self._pid_i))
        self._delta_p = self._delta_p * self._pid_delta_p_ema_alpha + (1 - self._pid_delta_p_ema_alpha) * delta
        self._cost_d = self._cost_d * self._pid_delta_d_ema_alpha + (1 - self._pid_delta_d_ema_alpha) * float(ep_cost_avg)
        pid_d = max(0.0, self._cost_d - self._cost_ds[0])
        pid_o = self._pid_kp * self._delta_p + self._pid_i + self._pid_kd * pid_d
        self._cost_penalty = max(0.0, pid_o)
        if self._diff_norm:
            self._cost_penalty = min(1.0, self._cost_penalty)
        if not (self._diff_norm or self._sum_norm):
            self._cost_penalty = min(self._cost_penalty, self._penalty_max)
        self._cost_ds.append(self._cost_d)
