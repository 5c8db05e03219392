"""TEST INFRASTRUCTURE -- import the UNMODIFIED reference from /root/reference.

Only usable in the build container (the GPU box has no /root/reference); used
by oracle/make_golden.py to generate tests/golden/*.npz and by
oracle/time_reference.py to time the reference CPU path.  Nothing in the
product path, in `-m gpu` tests, in smoke() or in bench.py imports this file.

What it does ("Making main() itself the oracle"):
  * pre-seeds sys.modules with empty stubs for packages the reference imports
    at module scope but that are not installed here (tensorboard,
    gymnasium, safety_gymnasium);
  * imports safepo.single_agent.{ppo_lag,cpo} from /root/reference;
  * replaces `make_sa_mujoco_env` by a factory returning oracle.synth_env.SynthEnv;
  * wraps `LinearLR` to drop the `verbose=` kwarg removed in torch 2.10
    (/root/reference/safepo/single_agent/ppo_lag.py:105-111).
No reference source is copied: the reference code runs from where it lies.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "safepo"))


class _NoopWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def _stub(name: str, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__path__ = []  # behave like a package so sub-imports resolve
    sys.modules[name] = mod
    return mod


def _install_stubs():
    class _Any:
        def __init__(self, *a, **k):
            pass

    if "tensorboard" not in sys.modules:
        _stub("tensorboard")
    import torch.utils  # noqa: F401
    tb = _stub("torch.utils.tensorboard", SummaryWriter=_NoopWriter)
    _stub("torch.utils.tensorboard.writer", SummaryWriter=_NoopWriter)
    import torch
    torch.utils.tensorboard = tb
    _stub("safety_gymnasium", make=None)
    _stub("safety_gymnasium.wrappers", SafeAutoResetWrapper=_Any, SafeRescaleAction=_Any,
          SafeUnsqueeze=_Any)
    _stub("safety_gymnasium.vector")
    _stub("safety_gymnasium.vector.async_vector_env", SafetyAsyncVectorEnv=_Any)
    _stub("safety_gymnasium.vector.utils")
    _stub("safety_gymnasium.vector.utils.tile_images", tile_images=None)
    _stub("safety_gymnasium.tasks")
    _stub("safety_gymnasium.tasks.safe_multi_agent")
    _stub("safety_gymnasium.tasks.safe_multi_agent.safe_mujoco_multi", SafeMAEnv=_Any)
    _stub("gymnasium")
    _stub("gymnasium.spaces", Box=_Any)
    _stub("gymnasium.vector")
    _stub("gymnasium.vector.vector_env", VectorEnv=_Any)
    _stub("gymnasium.wrappers")
    _stub("gymnasium.wrappers.normalize", NormalizeObservation=_Any)


def load_reference(algo: str = "ppo_lag"):
    """Return the reference module safepo.single_agent.<algo>, patched as above."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected on the GPU box)")
    for name in list(sys.modules):
        if name == "safepo" or name.startswith("safepo."):
            mod = sys.modules[name]
            where = getattr(mod, "__file__", None) or next(iter(getattr(mod, "__path__", [""])), "")
            if not str(where).startswith(REF_ROOT):
                raise RuntimeError(
                    "a non-reference `safepo` package is already imported in this process; "
                    "run the reference in its own interpreter")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _install_stubs()
    mod = importlib.import_module(f"safepo.single_agent.{algo}")
    assert mod.__file__.startswith(REF_ROOT), mod.__file__
    if hasattr(mod, "LinearLR"):
        real = mod.LinearLR
        if not getattr(real, "_shimmed", False):
            def linear_lr(*a, **k):
                k.pop("verbose", None)
                return real(*a, **k)
            linear_lr._shimmed = True
            mod.LinearLR = linear_lr
    return mod


def set_env_factory(mod, factory):
    """`factory(num_envs, env_id, seed)` -> (env, obs_space, act_space)."""
    mod.make_sa_mujoco_env = lambda num_envs, env_id, seed=None: factory(num_envs, env_id, seed)


def make_args(**over):
    """Namespace equal to single_agent_args() defaults
    (/root/reference/safepo/utils/config.py:145-162) plus overrides."""
    import argparse
    d = dict(seed=0, use_eval=False, task="SynthSafe-v0", num_envs=4, experiment="oracle",
             log_dir="/tmp/oracle_runs/exp/task/run", device="cpu", device_id=0,
             write_terminal=True, headless=False, total_steps=512, steps_per_epoch=512,
             randomize=False, cost_limit=25.0, lagrangian_multiplier_init=0.001,
             lagrangian_multiplier_lr=0.035)
    d.update(over)
    return argparse.Namespace(**d)
