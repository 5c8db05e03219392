"""HAPPO (multi-agent, unconstrained): reference safepo/multi_agent/happo.py.

MAPPO-Lagrangian without the cost side, on the same MI355X kernels (csrc/ma_net.hip): the clipped surrogate on the joint
ratio with the sequential-update factor (happo.py:144-157; spo_ma_actor_loss with a zero multiplier), the clipped Huber
value loss optionally averaged over active rows (happo.py:106-122; spo_ma_value_loss with the active-mask pointer), PopArt
and clip_grad_norm_ + Adam kernels.  Advantages are standardised with plain torch.mean / torch.std and + 1e-5
(happo.py:171-175).  Surface: HAPPO_Policy, HAPPO_Trainer, Runner, train(args, cfg_train).
"""
from __future__ import annotations

from safepo.multi_agent import mappolag as _base

# marl_cfg/happo/config.yaml and its `mamujoco` block
default_cfg = dict(_base.default_cfg, env_name="happo", algorithm_name="happo", episode_length=75, actor_lr=5.0e-4,
                   critic_lr=5.0e-4)
for _k in ("cost_limit", "lagrangian_coef_rate", "lamda_lagr"):
    default_cfg.pop(_k)
mamujoco_cfg = dict(num_env_steps=10000000, episode_length=1000, n_rollout_threads=10, n_eval_rollout_threads=10,
                    hidden_size=128, gamma=0.99, entropy_coef=0.01)


class HAPPO_Policy(_base.MAPPO_L_Policy):
    """happo.py:46-93: actor + one critic."""
    use_cost = False


class HAPPO_Trainer(_base.MAPPO_L_Trainer):
    """happo.py:96-205."""
    algo = "happo"


class Runner(_base.Runner):
    """happo.py:208-540."""
    policy_cls = HAPPO_Policy
    trainer_cls = HAPPO_Trainer


def train(args, cfg_train):
    return _base.train(args, cfg_train, runner_cls=Runner)


if __name__ == "__main__":
    _base.cli("happo", train)
