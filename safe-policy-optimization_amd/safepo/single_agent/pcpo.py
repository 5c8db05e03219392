"""PCPO (Projection-based Constrained Policy Optimization): reference safepo/single_agent/pcpo.py.
CPO's collect / GAE / two policy gradients / two conjugate-gradient solves / line search / critic fit, with the
projection step of pcpo.py:392-402 in place of CPO's case analysis.  Runs on the CPO kernels (CPOEngine.pcpo_update)."""
from __future__ import annotations

import os
import sys
import time

from safepo.single_agent import cpo as _cpo
from safepo.utils.config import single_agent_args

default_cfg = dict(_cpo.default_cfg)


def main(args, cfg_env=None):
    return _cpo.main(args, cfg_env, _update="pcpo")


if __name__ == "__main__":
    args, cfg_env = single_agent_args()
    relpath = time.strftime("%Y-%m-%d-%H-%M-%S")
    subfolder = "-".join(["seed", str(args.seed).zfill(3)])
    relpath = "-".join([subfolder, relpath])
    algo = os.path.basename(__file__).split(".")[0]
    args.log_dir = os.path.join(args.log_dir, args.experiment, args.task, algo, relpath)
    if not args.write_terminal:
        os.makedirs(args.log_dir, exist_ok=True)
        with open(os.path.join(args.log_dir, f"seed{args.seed}_terminal.log"), "w", encoding="utf-8") as f_out, \
                open(os.path.join(args.log_dir, f"seed{args.seed}_error.log"), "w", encoding="utf-8") as f_err:
            sys.stdout, sys.stderr = f_out, f_err
            main(args, cfg_env)
    else:
        main(args, cfg_env)
