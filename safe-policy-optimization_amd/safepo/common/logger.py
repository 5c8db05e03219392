"""Logger / EpochLogger with the reference's API and on-disk formats.

Consumer contract kept (reference safepo/common/logger.py:101-373): `progress.csv` (header row on
first dump, one row per dump), `config.json` (sorted keys, tab after colon, plus exp_name),
`torch_save/model{itr}.pt` holding the saved module's state_dict, `state{itr}.pkl` via joblib,
optional TensorBoard scalars under `tb/` (skipped when tensorboard is not installed), and the
`store / log_tabular / get_stats / dump_tabular / logged` protocol used by the training loops
(get_stats returns 0.0 until the key has appeared in a dumped header, logger.py:369-373).
No compute happens here."""
from __future__ import annotations

import atexit
import csv
import json
import os
import os.path as osp
import warnings

import numpy as np
import torch

try:  # optional
    from torch.utils.tensorboard.writer import SummaryWriter  # type: ignore
except Exception:  # pragma: no cover - tensorboard absent in this image
    SummaryWriter = None

_COLORS = dict(gray=30, red=31, green=32, yellow=33, blue=34, magenta=35, cyan=36, white=37, crimson=38)


def colorize(string, color, bold=False, highlight=False):
    code = _COLORS[color] + (10 if highlight else 0)
    parts = [str(code)] + (["1"] if bold else [])
    return f"\x1b[{';'.join(parts)}m{string}\x1b[0m"


def _jsonable(obj):
    try:
        json.dumps(obj)
        return obj
    except Exception:
        pass
    if isinstance(obj, dict):
        return {_jsonable(k): _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(x) for x in obj]
    if hasattr(obj, "__name__") and "lambda" not in obj.__name__:
        return _jsonable(obj.__name__)
    if hasattr(obj, "__dict__") and obj.__dict__:
        return {str(obj): {_jsonable(k): _jsonable(v) for k, v in obj.__dict__.items()}}
    return str(obj)


class Logger:
    def __init__(self, log_dir, seed=None, output_fname="progress.csv", debug: bool = False, level: int = 1,
                 use_tensorboard=True, verbose=True):
        self.log_dir, self.debug, self.level, self.verbose = log_dir, debug, level, verbose
        os.makedirs(log_dir, exist_ok=True)
        self.output_file = open(osp.join(log_dir, output_fname), encoding="utf-8", mode="w")
        atexit.register(self.output_file.close)
        self._csv_writer = csv.writer(self.output_file)
        self.epoch = 0
        self.first_row = True
        self.log_headers = []
        self.log_current_row = {}
        parts = log_dir.split("/")
        self.exp_name = "-".join([parts[-3], parts[-2], "seed", str(seed)]) if len(parts) >= 3 else f"seed-{seed}"
        self.torch_saver_elements = None
        self.use_tensorboard = bool(use_tensorboard) and SummaryWriter is not None
        self.logged = True
        if self.use_tensorboard:
            self.summary_writer = SummaryWriter(osp.join(log_dir, "tb"))

    def close(self):
        self.output_file.close()

    def log(self, msg, color="green"):
        if self.verbose and self.level > 0:
            print(colorize(msg, color, bold=False))

    def log_tabular(self, key, val):
        if self.first_row:
            self.log_headers.append(key)
        else:
            assert key in self.log_headers, f"Trying to introduce a new key {key} that you didn't include in the first iteration"
        assert key not in self.log_current_row, f"You already set {key} this iteration. Maybe you forgot to call dump_tabular()"
        self.log_current_row[key] = val

    def save_config(self, config):
        cfg = _jsonable(config)
        if self.exp_name is not None:
            cfg["exp_name"] = self.exp_name
        with open(osp.join(self.log_dir, "config.json"), "w") as out:
            out.write(json.dumps(cfg, separators=(",", ":\t"), indent=4, sort_keys=True))

    def save_state(self, state_dict, itr=None):
        import joblib
        fname = "state.pkl" if itr is None else "state%d.pkl" % itr
        try:
            joblib.dump(state_dict, osp.join(self.log_dir, fname))
        except Exception:
            self.log("Warning: could not pickle state_dict.", color="red")
        if hasattr(self, "torch_saver_elements"):
            self.torch_save(itr)

    def setup_torch_saver(self, what_to_save):
        self.torch_saver_elements = what_to_save

    def torch_save(self, itr=None):
        """model{itr}.pt = state_dict of the registered module (what evaluate.py loads,
        reference evaluate.py:53).  Parameters are views into the flat GPU vector, so they are
        cloned to contiguous CPU tensors first (otherwise torch.save would pickle all of theta)."""
        self.log("Save model to disk...")
        assert self.torch_saver_elements is not None, "First have to setup saving with self.setup_torch_saver"
        fpath = osp.join(self.log_dir, "torch_save")
        os.makedirs(fpath, exist_ok=True)
        fname = osp.join(fpath, "model" + ("%d" % itr if itr is not None else "") + ".pt")
        sd = {k: v.detach().cpu().clone() for k, v in self.torch_saver_elements.state_dict().items()}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.save(sd, fname)
        self.log("Done.")

    def dump_tabular(self) -> None:
        self.epoch += 1
        show = self.verbose and self.level > 0
        width = max(15, max(len(k) for k in self.log_headers))
        fmt = "| %" + str(width) + "s | %15s |"
        dashes = "-" * (22 + width)
        if show:
            print(dashes)
        for key in self.log_headers:
            val = self.log_current_row.get(key, "")
            if show:
                print(fmt % (key, "%8.3g" % val if hasattr(val, "__float__") else val))
        if show:
            print(dashes, flush=True)
        if self.first_row:
            self._csv_writer.writerow(self.log_current_row.keys())
        self._csv_writer.writerow(self.log_current_row.values())
        self.output_file.flush()
        if self.use_tensorboard:
            for key, val in self.log_current_row.items():
                self.summary_writer.add_scalar(key, val, global_step=self.epoch)
        self.log_current_row.clear()
        self.first_row = False


class EpochLogger(Logger):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.epoch_dict = {}

    def dump_tabular(self):
        self.logged = True
        super().dump_tabular()
        for k, v in self.epoch_dict.items():
            if len(v) > 0:
                print(f"epoch_dict: key={k} was not logged.")

    def store(self, add_value=False, **kwargs):
        for k, v in kwargs.items():
            if add_value:
                self.log_current_row.setdefault(k, [0])[0] += v
            else:
                self.epoch_dict.setdefault(k, []).append(v)

    def log_tabular(self, key, val=None, min_and_max=False, std=False):
        if val is not None:
            super().log_tabular(key, val)
        else:
            vals = self.epoch_dict[key]
            super().log_tabular(key, np.mean(vals))
            if min_and_max:
                super().log_tabular(key + "/Min", np.min(vals))
                super().log_tabular(key + "/Max", np.max(vals))
            if std:
                super().log_tabular(key + "/Std", np.std(vals))
        self.epoch_dict[key] = []

    def get_stats(self, key):
        if key not in self.log_headers:
            return 0.0
        return np.mean(self.epoch_dict[key])
