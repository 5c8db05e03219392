"""FOCOPS (First Order Constrained Optimization in Policy Space): reference safepo/single_agent/focops.py.
The ppo_lag epoch loop with (a) the multiplier bounded by FOCOPS_NU, (b) the actor loss
    ((KL(pi || pi_old) - (1/FOCOPS_LAM) * ratio * adv) * [KL <= target_kl]).mean()          (focops.py:326-337)
evaluated in the persistent update kernel (spo_update_iter_ex, SPO_ACTOR_LOSS_KL_PENALTY).
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
    return _first_order.run(args, cfg_env, default_cfg, multiplier="adam", clip=0.2, variant="focops")


if __name__ == "__main__":
    run_as_script(main, __file__)
