"""CUP (Constrained Update Projection): reference safepo/single_agent/cup.py.  Stage one is the PPO update on adv_r
(cup.py:300-352); stage two trains the actor alone on  nu*coef*ratio*adv_c + KL(pi || pi_after_stage_one)
(cup.py:354-405) with its own clip_grad_norm_ over the actor's parameters.  Both stages run on the persistent update
kernel (spo_update_iter_ex); the multiplier is bounded by CUP_NU.
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
    return _first_order.run(args, cfg_env, default_cfg, multiplier="adam", clip=0.2, variant="cup")


if __name__ == "__main__":
    run_as_script(main, __file__)
