"""PCPO (Projection-based Constrained Policy Optimization): reference safepo/single_agent/pcpo.py.
CPO's collect / GAE / two policy gradients / two conjugate-gradient solves / line search / critic fit, with the
projection step of pcpo.py:392-402 in place of CPO's case analysis.  Runs on the CPO kernels (CPOEngine.pcpo_update)."""
from __future__ import annotations

from safepo.single_agent import cpo as _cpo
from safepo.utils.config import run_as_script

default_cfg = dict(_cpo.default_cfg)


def main(args, cfg_env=None):
    return _cpo.main(args, cfg_env, _update="pcpo")


if __name__ == "__main__":
    run_as_script(main, __file__)
