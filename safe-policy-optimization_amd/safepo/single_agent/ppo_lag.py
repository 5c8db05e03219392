"""PPO-Lagrangian, MI355X-native hot path behind the reference entry point.

`main(args, cfg_env=None)`, `default_cfg` and the `python ppo_lag.py --task ... --num-envs ...` command line are
those of the reference script (safepo/single_agent/ppo_lag.py:45-52,67,390-426).  The loop itself lives in
safepo.single_agent._first_order (shared with ppo / pg / cppo_pid); every hot loop is a HIP kernel driven by
safepo.common.engine.PPOLagEngine.

Sharding: under torchrun (WORLD_SIZE > 1) `--num-envs` is the GLOBAL env count, split over ranks; each rank samples
minibatches of `batch_size` rows from its own shard and the flat gradient is all-reduced (RCCL) at every minibatch
step, so all replicas stay bit-identical.
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


# The reference's configuration for Isaac Gym tasks (ppo_lag.py:54-65).  The simulator itself is out of scope here, but its
# network / batch regime is built: hidden_sizes other than [64, 64] run on the wide-network kernels (safepo.common.wide),
# `num_mini_batch` without `batch_size` gives minibatches of steps_per_epoch // num_mini_batch rows; select it with
# args.cfg_override = isaac_gym_specific_cfg on a synthetic or host env.
isaac_gym_specific_cfg = {
    'total_steps': 100000000,
    'steps_per_epoch': 32768,
    'hidden_sizes': [1024, 1024, 512],
    'gamma': 0.96,
    'target_kl': 0.016,
    'num_mini_batch': 4,
    'use_value_coefficient': True,
    'learning_iters': 8,
    'max_grad_norm': 1.0,
    'use_critic_norm': False,
    'batch_size': None,        # the reference REPLACES default_cfg for Isaac tasks (ppo_lag.py:77-91), so no batch_size survives:
}                              # an override of None drops default_cfg's 64 and num_mini_batch decides (steps_per_epoch // 4)


def main(args, cfg_env=None):
    return _first_order.run(args, cfg_env, default_cfg, multiplier="adam", clip=0.2)


if __name__ == "__main__":
    run_as_script(main, __file__)
