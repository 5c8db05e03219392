"""MAPPO (multi-agent, unconstrained): reference safepo/multi_agent/mappo.py.

As happo.py, except that the importance ratio stays PER ACTION DIMENSION -- surr = exp(logp - logp_old)[B, A] * adv[B, 1],
min(surr1, surr2) summed over the dimensions (mappo.py:138-148) -- which is `per_dim_ratio` of spo_ma_actor_loss; the
sequential-update factor is tracked by the Runner (mappo.py:408-437) but does not enter the loss, and the value loss is
always a plain mean (mappo.py:106-117).  Surface: MAPPO_Policy, MAPPO_Trainer, Runner, train(args, cfg_train).
"""
from __future__ import annotations

from safepo.multi_agent import mappolag as _base

# marl_cfg/mappo/config.yaml and its `mamujoco` block
default_cfg = dict(_base.default_cfg, env_name="mappo", algorithm_name="mappo", n_rollout_threads=80, use_valuenorm=False)
for _k in ("cost_limit", "lagrangian_coef_rate", "lamda_lagr"):
    default_cfg.pop(_k)
mamujoco_cfg = dict(num_env_steps=10000000, episode_length=1000, n_rollout_threads=10, n_eval_rollout_threads=10,
                    hidden_size=128, gamma=0.99, entropy_coef=0.01, max_grad_norm=10.0, use_value_active_masks=True,
                    use_policy_active_masks=True, data_chunk_length=10, use_valuenorm=False)


class MAPPO_Policy(_base.MAPPO_L_Policy):
    """mappo.py:46-93: actor + one critic."""
    use_cost = False


class MAPPO_Trainer(_base.MAPPO_L_Trainer):
    """mappo.py:96-189."""
    algo = "mappo"


class Runner(_base.Runner):
    """mappo.py:192-532."""
    policy_cls = MAPPO_Policy
    trainer_cls = MAPPO_Trainer


def train(args, cfg_train):
    return _base.train(args, cfg_train, runner_cls=Runner)


if __name__ == "__main__":
    _base.cli("mappo", train)
