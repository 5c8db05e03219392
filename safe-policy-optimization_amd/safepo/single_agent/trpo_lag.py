"""TRPO-Lagrangian (reference safepo/single_agent/trpo_lag.py): trpo on the Lagrangian-mixed advantage (trpo_lag.py:326-327).
Runs on the CPO kernels (safepo.single_agent._second_order).
"""
from __future__ import annotations

from safepo.single_agent import _second_order
from safepo.utils.config import run_as_script

default_cfg = {
    'hidden_sizes': [64, 64],
    'gamma': 0.99,
    'target_kl': 0.01,
    'batch_size': 128,
    'learning_iters': 10,
    'max_grad_norm': 40.0,
}


def main(args, cfg_env=None):
    return _second_order.run(args, cfg_env, default_cfg, line_search=True, use_lagrange=True)


if __name__ == "__main__":
    run_as_script(main, __file__)
