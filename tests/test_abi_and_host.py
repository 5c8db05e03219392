"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/safepo_hip.h declares;
host-side logic (CLI surface, logger formats, Lagrange, env sharding) matches the reference's contract."""
import csv
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    return g.LIB


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    from safepo import _abi
    header = open(os.path.join(ROOT, "include", "safepo_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(spo_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_abi.PROTOTYPES), (declared ^ set(_abi.PROTOTYPES))
    lib = _abi.load(built_lib)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.spo_abi_version() == _abi.ABI_VERSION == 2
    # geometry helpers are host functions: 2*8129 + 8592 = 24850
    assert lib.spo_param_count(60, 8) == 24850
    assert [lib.spo_param_offset(60, 8, k) for k in range(3)] == [0, 8129, 16258]
    # 4 rows per block while the buffer is cache-resident (reward / cost scans in separate lane groups), 8 when it streams
    raw = open(os.path.join(ROOT, "include", "safepo_hip.h")).read()
    assert int(re.search(r"#define SPO_GAE_PARTIAL_STRIDE (\d+)", raw).group(1)) == _abi.GAE_PARTIAL_STRIDE == 16
    assert lib.spo_gae_num_blocks(4096, 128) == 1024
    assert lib.spo_gae_num_blocks(262144, 128) == 32768


def test_argument_errors_without_gpu(built_lib):
    from safepo import _abi
    lib = _abi.load(built_lib)
    assert lib.spo_gae_fused(None, None, None, None, None, None, None, None, None, None, None, None, 4, 8, 0.99, 0.95, 0.95, None) < 0
    assert b"null" in lib.spo_last_error()
    assert lib.spo_policy_step(None, None, None, None, None, None, None, None, None, None, None, None, 4, 1, 0, 500, 8, None) < 0
    assert b"obs_dim" in lib.spo_last_error()


def test_feature_split_entry_points_shape_envelope_and_argument_errors(built_lib):
    """Round 5: the persistent feature-split kernels (csrc/update_ks.hip) say which shapes they take without touching a GPU, and
    their entry points refuse bad arguments before any launch; the exchange-form selectors know which forms exist per world size."""
    import ctypes
    from safepo import _abi
    lib = _abi.load(built_lib)
    assert lib.spo_ks_supported(376, 17, 64) == 1 and lib.spo_ks_supported(512, 32, 1) == 1 and lib.spo_ks_supported(1, 1, 64) == 1
    assert lib.spo_ks_supported(513, 17, 64) == 0 and lib.spo_ks_supported(376, 33, 64) == 0 and lib.spo_ks_supported(376, 17, 65) == 0
    assert lib.spo_critic_fit_ks_supported(376, 128) == 1 and lib.spo_critic_fit_ks_supported(512, 64) == 1
    assert lib.spo_critic_fit_ks_supported(513, 128) == 0 and lib.spo_critic_fit_ks_supported(376, 129) == 0
    cfg = _abi.PpoCfg(obs_dim=600, act_dim=17, batch=64, use_critic_norm=1, use_value_coefficient=0, clip=0.2, max_grad_norm=40.0,
                      lr_actor=3e-4, lr_critic=3e-4, beta1=0.9, beta2=0.999, adam_eps=1e-8, l2_coef=0.001)
    assert lib.spo_ppo_lag_update_iter_ks(None, None, None, 0, None, None, None, None, None, None, None, 64, ctypes.byref(cfg),
                                          None, None, None) < 0
    assert b"obs_dim 600" in lib.spo_last_error()
    cfg.obs_dim = 376
    assert lib.spo_ppo_lag_update_iter_ks(None, None, None, 0, None, None, None, None, None, None, None, 64, ctypes.byref(cfg),
                                          None, None, None) < 0
    assert b"null pointer" in lib.spo_last_error()
    cfg.batch = 200
    assert lib.spo_critic_fit_iter_ks(None, None, None, 0, None, None, None, None, 64, ctypes.byref(cfg), None, None, None, None) < 0
    assert b"batch 200" in lib.spo_last_error()
    # forms of the in-kernel exchange: doubling needs a power-of-two world, the helper-wave forms exist at 2 / 4 / 8
    for world in (2, 4, 8):
        assert all(lib.spo_p2p_form_valid(f, world) == 1 for f in (0, 1, 2, 3)), world
    assert lib.spo_p2p_form_valid(0, 3) == 1 and lib.spo_p2p_form_valid(1, 3) == 0 and lib.spo_p2p_form_valid(1, 6) == 0
    assert lib.spo_p2p_form_valid(7, 2) == 0 and lib.spo_p2p_form_valid(0, 9) == 0


def test_round6_entry_points_shape_envelope_and_argument_errors(built_lib):
    """Round 6 without a GPU: which shapes the row-group gradient kernel (csrc/mlp_rows.hip) and the row-split kernel
    (csrc/update_rs.hip) take, the size of the partial-gradient buffer, and that the new entry points refuse bad arguments before
    any launch."""
    import ctypes
    from safepo import _abi
    lib = _abi.load(built_lib)
    net = _abi.MlpNet.of
    ok = lib.spo_wide_grad_rows_supported
    assert ok(net([60, 128, 128, 1]), net([60, 128, 128, 8]), 64) == 1 and ok(net([60, 128, 128, 1]), net([60, 128, 128, 8]), 256) == 1
    assert ok(net([60, 128, 128, 1]), net([60, 128, 128, 8]), 257) == 0 and ok(net([60, 128, 128, 1]), net([60, 128, 128, 8]), 0) == 0
    assert ok(net([60, 256, 256, 1]), net([60, 256, 256, 8]), 64) == 1            # 157.4 KB of the 160 KB
    assert ok(net([376, 256, 256, 1]), net([376, 256, 256, 17]), 64) == 0         # the images do not fit a CU's LDS
    assert ok(net([60, 1024, 1024, 512, 1]), net([60, 1024, 1024, 512, 8]), 64) == 0
    assert ok(net([60, 128, 1]), None, 128) == 1                                  # the two critics alone (critic fit)
    assert ok(net([60, 128, 2]), None, 64) == 0 and ok(net([60, 128, 1]), net([61, 128, 8]), 64) == 0
    assert ok(net([60, 128, 1]), net([60, 128, 65]), 64) == 0                     # act_dim beyond SPO_WIDE_MAX_ACT
    P = 2 * (60 * 128 + 128 + 128 + 1) + 8 + (60 * 128 + 128 + 128 * 8 + 8)
    assert lib.spo_wide_grad_rows_part_floats(P, 64) == 4 * ((P + 4 + 3) // 4 * 4) + 4
    assert lib.spo_wide_grad_rows_part_floats(P, 17) == 2 * ((P + 4 + 3) // 4 * 4) + 4 and lib.spo_wide_grad_rows_part_floats(0, 64) < 0
    assert lib.spo_wide_ppo_grad_rows(None, net([60, 128, 1]), None, None, None, None, None, None, None, None, None, 64, 0.2, None, None) < 0
    assert b"null pointer" in lib.spo_last_error()
    assert lib.spo_wide_reduce_parts(None, 64, P, 3, None, None, None) < 0 and b"bad args" in lib.spo_last_error()
    assert lib.spo_update_rs_supported(60, 8, 64, 3) == 1 and lib.spo_update_rs_supported(64, 16, 128, 2) == 1
    assert lib.spo_update_rs_supported(65, 8, 64, 3) == 0 and lib.spo_update_rs_supported(60, 17, 64, 3) == 0
    assert lib.spo_update_rs_supported(60, 8, 65, 3) == 0 and lib.spo_update_rs_supported(60, 8, 129, 2) == 0
    cfg = _abi.PpoCfg(obs_dim=376, act_dim=17, batch=64, use_critic_norm=1, use_value_coefficient=0, clip=0.2, max_grad_norm=40.0,
                      lr_actor=3e-4, lr_critic=3e-4, beta1=0.9, beta2=0.999, adam_eps=1e-8, l2_coef=0.001)
    assert lib.spo_ppo_lag_grad_ks(None, None, None, None, None, None, None, None, 65, ctypes.byref(cfg), None, None, None, None) < 0
    assert b"rows 65" in lib.spo_last_error()
    assert lib.spo_ppo_lag_grad_ks(None, None, None, None, None, None, None, None, 64, ctypes.byref(cfg), None, None, None, None) < 0
    assert b"null pointer" in lib.spo_last_error()


def test_missing_library_fails_loudly(tmp_path):
    from safepo import _abi
    with pytest.raises(_abi.SpoError, match="no CPU fallback"):
        _abi.load(str(tmp_path / "nope.so"))


def test_cpu_tensors_are_rejected(built_lib):
    from safepo import _abi
    from safepo.common.buffer import VectorizedOnPolicyBuffer
    from safepo.common.model import ActorVCritic
    pol = ActorVCritic(60, 8)
    with pytest.raises(_abi.SpoError, match="GPU tensor"):
        pol.step(torch.zeros(4, 60))
    class S:  # noqa
        shape = (60,)
    with pytest.raises(_abi.SpoError, match="no CPU fallback"):
        VectorizedOnPolicyBuffer(S(), S(), size=8, num_envs=2, device="cpu")


def test_policy_parameter_layout_matches_reference_order():
    from safepo.common.model import ActorVCritic
    torch.manual_seed(0)
    pol = ActorVCritic(60, 8)
    names = [n for n, _ in pol.named_parameters()]
    assert names[:2] == ["reward_critic.critic.0.weight", "reward_critic.critic.0.bias"]
    assert names[6] == "cost_critic.critic.0.weight" and names[12] == "actor.log_std"
    assert names[13:] == [f"actor.mean.{i}.{k}" for i in (0, 2, 4) for k in ("weight", "bias")]
    assert set(pol.actor.state_dict()) == {"log_std", "mean.0.weight", "mean.0.bias", "mean.2.weight",
                                           "mean.2.bias", "mean.4.weight", "mean.4.bias"}
    # parameters are views into ONE flat vector, in policy.parameters() order
    flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    assert torch.equal(flat, pol.theta) and pol.theta.numel() == 24850
    # same RNG consumption as the reference constructor => same init for the same seed
    from oracle import restatement as R
    torch.manual_seed(0)
    ref = R.OraclePolicy(60, 8)
    assert torch.equal(R.flat_params(ref), pol.theta)
    pol.actor.mean[0].weight.data[3, 5] = 42.0
    assert pol.theta[2 * 8129 + 8 + 3 * 60 + 5] == 42.0
    with pytest.raises(NotImplementedError):
        ActorVCritic(60, 8, hidden_sizes=[1024, 1024, 512])._require_kernels()
    # routing by (obs_dim, act_dim) too (round 4): what the persistent kernels cannot hold goes to the wide path, the
    # full-batch CPO kernels have the narrower envelope (obs_dim <= 64)
    for (D, A, hs), ppo_ok, cpo_ok in [((60, 8, [64, 64]), True, True), ((72, 2, [64, 64]), True, False),
                                       ((128, 16, [64, 64]), True, False), ((129, 4, [64, 64]), False, False),
                                       ((376, 17, [64, 64]), False, False), ((10, 17, [64, 64]), False, False),
                                       ((60, 8, [128, 128]), False, False)]:
        p2 = ActorVCritic(D, A, hidden_sizes=hs)
        assert p2.kernels_supported() is ppo_ok and p2.kernels_supported("cpo") is cpo_ok, (D, A, hs)
    with pytest.raises(NotImplementedError):
        ActorVCritic(72, 2)._require_kernels("cpo")


def test_isaac_cfg_override_selects_the_num_mini_batch_regime():
    """passing the exported isaac_gym_specific_cfg itself as cfg_override must give minibatches of
    steps_per_epoch // num_mini_batch rows (the reference REPLACES default_cfg for Isaac tasks, ppo_lag.py:77-91, so no
    batch_size of 64 survives) -- the merge of _first_order.run, without a GPU."""
    from safepo.single_agent import ppo_lag
    config = dict(ppo_lag.default_cfg)
    config.update(ppo_lag.isaac_gym_specific_cfg)
    config = {k: v for k, v in config.items() if v is not None}          # _first_order.run
    assert "batch_size" not in config and config["num_mini_batch"] == 4 and config["hidden_sizes"] == [1024, 1024, 512]
    M = config["steps_per_epoch"]
    assert config.get("batch_size", max(M // config.get("num_mini_batch", 1), 1)) == M // 4 == 8192      # PPOLagEngine._cfg_struct


def test_cli_flags_match_reference_table():
    from safepo.utils.config import build_parser, single_agent_args
    flags = {a.option_strings[0]: a.default for a in build_parser()._actions if a.option_strings and a.dest != "help"}
    expected = {"--seed": 0, "--use-eval": False, "--task": "SafetyPointGoal1-v0", "--num-envs": 10,
                "--experiment": "single_agent_exp", "--log-dir": "../runs", "--device-id": 0,
                "--write-terminal": True, "--headless": False, "--total-steps": 10000000,
                "--steps-per-epoch": 20000, "--randomize": False, "--cost-limit": 25.0,
                "--lagrangian-multiplier-init": 0.001, "--lagrangian-multiplier-lr": 0.035}
    for k, v in expected.items():
        assert flags[k] == v, k
    assert set(flags) == set(expected) | {"--device"}
    args, cfg_env = single_agent_args(["--task", "SynthSafe-v0", "--num-envs", "4", "--use-eval", "False",
                                       "--write-terminal", "no", "--seed", "3"])
    assert (args.num_envs, args.use_eval, args.write_terminal, args.seed, cfg_env) == (4, False, False, 3, {})


def test_logger_files_and_get_stats_protocol(tmp_path):
    from safepo.common.logger import EpochLogger
    from safepo.common.model import ActorVCritic
    d = str(tmp_path / "exp" / "task" / "run")
    lg = EpochLogger(log_dir=d, seed="7", verbose=False)
    assert lg.exp_name == "exp-task-seed-7"
    lg.save_config({"seed": 7, "hidden_sizes": [64, 64], "fn": print})
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg["exp_name"] == "exp-task-seed-7" and cfg["fn"] == "print"
    assert lg.get_stats("Metrics/EpCost") == 0.0                     # unknown key before the first dump
    lg.store(**{"Metrics/EpCost": 2.0}); lg.store(**{"Metrics/EpCost": 4.0})
    assert lg.get_stats("Metrics/EpCost") == 0.0                     # still not in a dumped header
    lg.log_tabular("Metrics/EpCost"); lg.log_tabular("Train/Epoch", 1)
    lg.dump_tabular()
    lg.store(**{"Metrics/EpCost": 10.0})
    assert lg.get_stats("Metrics/EpCost") == 10.0
    lg.log_tabular("Metrics/EpCost"); lg.log_tabular("Train/Epoch", 2)
    with pytest.raises(AssertionError):
        lg.log_tabular("New/Key", 1)
    lg.dump_tabular()
    rows = list(csv.reader(open(os.path.join(d, "progress.csv"))))
    assert rows == [["Metrics/EpCost", "Train/Epoch"], ["3.0", "1"], ["10.0", "2"]]
    pol = ActorVCritic(6, 2)
    lg.setup_torch_saver(pol.actor)
    lg.save_state({"Normalizer": {"mean": np.zeros(6)}}, itr=0)
    sd = torch.load(os.path.join(d, "torch_save", "model0.pt"))
    assert set(sd) == set(pol.actor.state_dict()) and sd["mean.0.weight"].shape == (64, 6)
    import joblib
    assert "Normalizer" in joblib.load(os.path.join(d, "state0.pkl"))
    lg.close()


def test_lagrange_matches_oracle():
    from oracle import restatement as R
    from safepo.common.lagrange import Lagrange
    a, b = Lagrange(25.0, 0.001, 0.035), R.OracleLagrange(25.0, 0.001, 0.035)
    for jc in (0.0, 30.0, 41.5, 12.0, 60.0, float(np.mean([26.0, 27.0]))):
        a.update_lagrange_multiplier(jc); b.update_lagrange_multiplier(jc)
        assert a.lagrangian_multiplier == b.lagrangian_multiplier
    assert a.lagrangian_multiplier >= 0.0
    c = Lagrange(1.0, 5.0, 0.5, lagrangian_upper_bound=5.2)
    for _ in range(5):
        c.update_lagrange_multiplier(100.0)
    assert c.lagrangian_multiplier == pytest.approx(5.2)


def test_shard_envs_partitions_all_envs():
    from safepo.parallel import shard_envs
    class C:  # noqa
        def __init__(self, w, r): self.world_size, self.rank = w, r
    for n, w in ((32768, 8), (10, 4), (7, 8), (4096, 1)):
        spans = [shard_envs(n, C(w, r)) for r in range(w)]
        assert sum(c for _, c in spans) == n
        pos = 0
        for s, c in spans:
            assert s == pos
            pos += c


def test_pid_lagrangian_matches_reference_golden(golden_dir):
    from safepo.common.lagrange import PIDLagrangian
    z = np.load(os.path.join(golden_dir, "pid.npz"))
    for tag, kw in {"default": {}, "diffnorm": {"diff_norm": True}, "nosum": {"sum_norm": False, "penalty_max": 0.5}}.items():
        pid = PIDLagrangian(cost_limit=25.0, lagrangian_multiplier_init=0.001, **kw)
        got = []
        for c in z["costs"]:
            pid.update_lagrange_multiplier(float(c))
            got.append(pid.lagrangian_multiplier)
        np.testing.assert_allclose(np.asarray(got), z[tag], rtol=1e-12, atol=0)


def test_sibling_scripts_share_the_reference_surface():
    import importlib
    for algo in ("ppo_lag", "ppo", "pg", "cppo_pid", "cpo", "natural_pg", "trpo", "rcpo", "trpo_lag"):
        m = importlib.import_module(f"safepo.single_agent.{algo}")
        assert callable(m.main) and m.default_cfg["hidden_sizes"] == [64, 64]
    from safepo.single_agent import ppo_lag, cpo
    assert ppo_lag.default_cfg == {'hidden_sizes': [64, 64], 'gamma': 0.99, 'target_kl': 0.02, 'batch_size': 64,
                                   'learning_iters': 40, 'max_grad_norm': 40.0}
    assert cpo.default_cfg["batch_size"] == 128 and cpo.default_cfg["target_kl"] == 0.01


def test_popart_matches_reference_statistics(golden_dir):
    """PopArt mirror: same debiased statistics as the reference state, update rule, normalize/denormalize round trip."""
    from safepo.common.popart import PopArt
    z = np.load(os.path.join(golden_dir, "ma_gae.npz"))
    torch.manual_seed(1)
    n = PopArt(1)
    x = torch.randn(64, 1) * 2.5 + 1.0
    y = n(x, train=True)
    assert n.debiasing_term.item() == pytest.approx(1 - 0.99999, rel=1e-3)
    np.testing.assert_allclose(n.denormalize(y).numpy(), x.numpy(), rtol=1e-4, atol=1e-4)
    n.running_mean.copy_(torch.from_numpy(z["a_rm"])); n.running_mean_sq.copy_(torch.from_numpy(z["a_rms"]))
    n.debiasing_term.copy_(torch.from_numpy(z["a_deb"]).reshape(()))
    mean, var = n.running_mean_var()
    assert np.array_equal(mean.numpy(), z["a_mean"]) and np.array_equal(var.numpy(), z["a_var"])
    sd, mu = n.denorm_scalars()
    assert sd == float(np.sqrt(z["a_var"])[0]) and mu == float(z["a_mean"][0])


def test_multi_agent_network_layout_matches_reference_state_dict(golden_dir):
    """f3: the flat parameter layout of the HIP multi-agent networks IS the reference's state_dict order -- names, shapes
    and offsets (spo_ma_param_offset is host-only arithmetic) -- so checkpoints load both ways."""
    import numpy as np
    from safepo import _abi
    from safepo.common.model import MultiAgentActor, MultiAgentCritic
    from safepo.multi_agent.mappolag import default_cfg
    z = np.load(os.path.join(golden_dir, "ma_mappolag.npz"))
    cfg = dict(default_cfg, hidden_size=int(z["default_cfg_hidden_size"]), layer_N=int(z["default_cfg_layer_N"]))

    class Sp:
        def __init__(self, n):
            self.shape = (n,)
    D, S, A = z["default_obs"].shape[1], z["default_share_obs"].shape[1], z["default_actions"].shape[1]
    lib = _abi.load()
    for nm, net in (("actor", MultiAgentActor(cfg, Sp(D), Sp(A), torch.device("cpu"))),
                    ("critic", MultiAgentCritic(cfg, Sp(S), torch.device("cpu")))):
        pre = f"default_init_{nm}_"
        ref = {k[len(pre):]: z[k].shape for k in z.files if k.startswith(pre)}
        mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert list(mine) == list(ref) and mine == ref
        assert lib.spo_ma_param_count(net._net) == net.theta.numel()
        # every parameter is a view at the offset the C side computes
        base = net.theta.data_ptr()
        sd = net.state_dict()
        nb = 1 + cfg["layer_N"]
        names = ["base.feature_norm.weight", "base.feature_norm.bias"]
        which = [(0, 0), (1, 0)]
        for k in range(nb):
            blk = "base.mlp.fc1" if k == 0 else f"base.mlp.fc2.{k - 1}"
            names += [f"{blk}.0.weight", f"{blk}.0.bias", f"{blk}.2.weight", f"{blk}.2.bias"]
            which += [(2, k), (3, k), (4, k), (5, k)]
        if nm == "actor":
            names += ["act.action_out.log_std", "act.action_out.fc_mean.weight", "act.action_out.fc_mean.bias"]
            which += [(6, 0), (7, 0), (8, 0)]
        else:
            names += ["v_out.weight", "v_out.bias"]
            which += [(7, 0), (8, 0)]
        assert names == list(sd)
        for n_, (w, k) in zip(names, which):
            assert (sd[n_].data_ptr() - base) // 4 == lib.spo_ma_param_offset(net._net, w, k), n_
        # loading the reference's weights is a plain load_state_dict
        net.load_state_dict({k[len(pre):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(pre)})
        assert np.array_equal(net.theta.numpy(), np.concatenate([z[k].reshape(-1) for k in z.files if k.startswith(pre)]))


def test_multi_agent_args_defaults_and_overrides():
    from safepo.multi_agent.mappolag import default_cfg, mamujoco_cfg
    from safepo.utils.config import multi_agent_args
    args, cfg_env, cfg = multi_agent_args("mappolag", ["--num-envs", "16", "--cost-limit", "3.5", "--seed", "7", "--total-steps", "4096"])
    assert cfg["n_rollout_threads"] == 16 and cfg["n_eval_rollout_threads"] == 16 and cfg["cost_limit"] == 3.5
    assert cfg["hidden_size"] == mamujoco_cfg["hidden_size"] and cfg["learning_iters"] == default_cfg["learning_iters"]
    assert cfg["num_env_steps"] == 4096 and cfg["algorithm_name"] == "mappolag" and cfg["device"] == "cuda:0"
    assert "seed-007" in cfg["log_dir"] and args.task.startswith("Synth")


@pytest.mark.parametrize("algo", ["happo", "mappo"])
def test_multi_agent_sibling_modules_surface_and_defaults(algo):
    """safepo/multi_agent/{happo,mappo}.py keep the reference's names and config.yaml defaults (incl. the mamujoco block)."""
    import importlib
    from safepo.utils.config import multi_agent_args
    M = importlib.import_module(f"safepo.multi_agent.{algo}")
    for name in (f"{algo.upper()}_Policy", f"{algo.upper()}_Trainer", "Runner", "train", "default_cfg", "mamujoco_cfg"):
        assert hasattr(M, name), name
    assert getattr(M, f"{algo.upper()}_Trainer").algo == algo and not getattr(M, f"{algo.upper()}_Policy").use_cost
    assert "lamda_lagr" not in M.default_cfg and M.default_cfg["algorithm_name"] == algo
    if algo == "happo":
        assert M.default_cfg["episode_length"] == 75 and M.default_cfg["actor_lr"] == 5e-4 and M.default_cfg["use_valuenorm"]
        assert "use_value_active_masks" not in M.mamujoco_cfg
    else:
        assert M.default_cfg["n_rollout_threads"] == 80 and M.default_cfg["actor_lr"] == 9e-5 and not M.default_cfg["use_valuenorm"]
        assert M.mamujoco_cfg["use_policy_active_masks"] and M.mamujoco_cfg["use_value_active_masks"]
    args, _, cfg = multi_agent_args(algo, ["--num-envs", "32", "--seed", "3"])
    assert cfg["algorithm_name"] == algo and cfg["n_rollout_threads"] == 32 and cfg["hidden_size"] == 128


def test_macpo_module_surface_and_defaults():
    from safepo.multi_agent import macpo
    from safepo.utils.config import multi_agent_args
    for name in ("MACPO_Policy", "MACPO_Trainer", "Runner", "train", "default_cfg", "mamujoco_cfg"):
        assert hasattr(macpo, name), name
    d = macpo.default_cfg
    assert d["algorithm_name"] == "macpo" and d["conjugate_gradient_iters"] == 10 and d["fraction_coef"] == 0.1 and d["step_fraction"] == 0.5
    assert "lamda_lagr" not in d and macpo.mamujoco_cfg["layer_N"] == 1 and macpo.mamujoco_cfg["target_kl"] == 0.01
    args, _, cfg = multi_agent_args("macpo", ["--num-envs", "32", "--cost-limit", "2.0"])
    assert cfg["algorithm_name"] == "macpo" and cfg["layer_N"] == 1 and cfg["cost_limit"] == 2.0 and cfg["hidden_size"] == 128
    assert macpo.Runner.log_keys[2] == "Loss/Loss_actor_improve" and "Misc/KL" in macpo.Runner.log_keys


def test_benchmark_launchers_build_the_reference_command_lines():
    """single_agent/benchmark.py and multi_agent/benchmark.py: one command per (seed, task, algo) with the reference's
    flags, seeds start + 1000 * k, runs dealt round-robin over the GPUs."""
    from safepo.multi_agent import benchmark as mb
    from safepo.single_agent import benchmark as sb
    a = sb.parse_args(["--tasks", "SynthSafe-v0", "SafetyDoggoGoal1-v0", "--algo", "ppo_lag", "cpo", "--num-seeds", "2", "--start-seed", "5",
                       "--total-steps", "4096", "--num-envs", "16", "--steps-per-epoch", "2048", "--workers", "0"])
    cmds = sb.build_commands(a, n_gpus=4)
    assert len(cmds) == 2 * 2 * 2
    assert cmds[0].split()[1].endswith("single_agent/ppo_lag.py") and "--seed 5 " in cmds[0] and "--write-terminal False" in cmds[0]
    assert "--total-steps 4096 --num-envs 16 --steps-per-epoch 2048 --device-id 0" in cmds[0]
    assert "--seed 1005 " in cmds[-1] and cmds[-1].split()[1].endswith("cpo.py")
    doggo = [c for c in cmds if "Doggo" in c]
    assert all("--total-steps 100000000 --num-envs 20 --steps-per-epoch 200000" in c for c in doggo)
    assert [c.rsplit(" ", 1)[1] for c in cmds] == ["0", "1", "2", "3", "0", "1", "2", "3"]
    assert sb.main(["--tasks", "SynthSafe-v0", "--algo", "pg", "--num-seeds", "1", "--workers", "0"])[0].count("pg.py") == 1
    m = mb.parse_args(["--tasks", "SynthMultiAgent-v0", "--num-seeds", "1", "--total-steps", "1024", "--num-envs", "8"])
    mc = mb.build_commands(m, n_gpus=2)
    assert [c.split()[1].rsplit("/", 1)[1] for c in mc] == ["macpo.py", "mappo.py", "mappolag.py", "happo.py"]
    assert all("--headless True --total-steps 1024 --num-envs 8" in c for c in mc)
    assert len(sb.NAVI_TASKS) == 40 and sb.NAVI_TASKS[0] == "SafetyAntButton1-v0" and len(sb.VEL_TASKS) == 6


def test_evaluate_picks_the_newest_checkpoint_numerically(tmp_path):
    """evaluate.py:40-47 sorts checkpoint names as strings (model9.pt after model10.pt); the epoch number decides here."""
    from safepo import evaluate
    d = tmp_path / "torch_save"
    d.mkdir()
    for n in ("model2.pt", "model10.pt", "model9.pt", "notes.txt"):
        (d / n).write_bytes(b"")
    assert os.path.basename(evaluate._latest(str(d), ".pt")) == "model10.pt"
    assert evaluate._latest(str(d), ".pkl") is None
    assert evaluate.MULTI_AGENT_ALGOS == ("macpo", "mappo", "mappolag", "happo")



def test_bench_refuses_to_report_n_gpus_it_does_not_have():
    """`python bench.py --gpus 8` without a launcher spawns its own ranks; on a box with fewer GPUs it must fail (exit
    code 2, no JSON line) instead of benching one GPU under an `n_gpus: 1` label; a WORLD_SIZE that disagrees with
    --gpus is refused too."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPO_BENCH_ONE_GPU")}
    env["HIP_VISIBLE_DEVICES"] = ""           # the check must not depend on what the test box has
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 2 and "only 0 GPU(s) are visible" in r.stderr and not r.stdout.strip()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


@pytest.mark.parametrize("hidden", [[64, 64], [128, 128], [256, 96], [1024, 1024, 512], [32]])
def test_wide_network_layout_matches_reference_parameter_order(built_lib, hidden):
    """The wide-network kernels address one flat vector per network: every Linear's weight [out, in] then its bias, in layer
    order -- nn.Sequential.parameters() order, so ActorVCritic's parameters stay views of the flat vector for any
    hidden_sizes (reference model.py:30-48,131-135).  Host-only check of spo_mlp_param_count and of the offsets
    safepo.common.wide.WideNets derives (reward critic, cost critic, log_std, actor)."""
    from oracle import restatement as R
    from safepo import _abi
    lib = _abi.load(built_lib)
    D, A = 60, 8
    ref = R.OraclePolicy(D, A, hidden_sizes=tuple(hidden))
    n_c = sum(p.numel() for p in ref.reward_critic.parameters())
    n_a = sum(p.numel() for p in ref.actor.mean.parameters())
    assert lib.spo_mlp_param_count(_abi.MlpNet.of([D] + hidden + [1])) == n_c
    assert lib.spo_mlp_param_count(_abi.MlpNet.of([D] + hidden + [A])) == n_a
    names = [k for k, _ in ref.named_parameters()]
    assert names[0].startswith("reward_critic") and names[2 * (len(hidden) + 1)].startswith("cost_critic")
    assert names[4 * (len(hidden) + 1)] == "actor.log_std" and names[4 * (len(hidden) + 1) + 1] == "actor.mean.0.weight"
    assert sum(p.numel() for p in ref.parameters()) == 2 * n_c + A + n_a
    assert lib.spo_mlp_workspace_floats(_abi.MlpNet.of([D] + hidden + [A]), 10) == 10 * (sum(hidden) + A)
    with pytest.raises(_abi.SpoError):
        _abi.MlpNet.of([D, 1, 1, 1, 1, 1, 1])          # more than 5 Linear layers


def test_episode_log_running_means_equal_the_per_episode_loop():
    """engine.deque_running_means (the finished-episode statistics of one epoch in one pass) against the reference's loop --
    append to a deque of 50, np.mean of the deque, per episode (ppo_lag.py:216-230): bit-identical, from an empty deque, across
    several calls, for batches shorter and longer than the window."""
    from collections import deque
    from safepo.common.engine import deque_running_means
    rng = np.random.default_rng(0)
    a, b = deque(maxlen=50), deque(maxlen=50)
    for n in (0, 3, 20, 26, 1, 120, 50, 49, 4096, 7):
        new = rng.standard_normal(n) * 100.0
        ref = []
        for v in new:
            a.append(v)
            ref.append(np.mean(a))
        got = deque_running_means(b, new)
        assert len(got) == n and all(np.float64(x).tobytes() == np.float64(y).tobytes() for x, y in zip(got, ref)), n
        assert list(a) == list(b)
    c = deque(a, maxlen=50)
    assert deque_running_means(c, rng.standard_normal(70), want_means=False) == [] and len(c) == 50
