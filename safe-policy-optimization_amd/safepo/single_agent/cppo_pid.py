"""CPPO-PID: reference safepo/single_agent/cppo_pid.py = ppo_lag with PIDLagrangian in place of Lagrange
(cppo_pid.py:39).  Same kernels; the multiplier is host-side scalar arithmetic.
"""
from __future__ import annotations

from safepo.single_agent import _first_order
from safepo.utils.config import run_as_script

default_cfg = {
    'hidden_sizes': [64, 64],
    'gamma': 0.99,
    'target_kl': 0.02,
    'batch_size': 64,
    'learning_iters': 40,
    'max_grad_norm': 40.0,
}


def main(args, cfg_env=None):
    return _first_order.run(args, cfg_env, default_cfg, multiplier="pid", clip=0.2)


if __name__ == "__main__":
    run_as_script(main, __file__)
