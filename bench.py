"""bench.py -- env-steps/sec (collect + GAE + update) of the PPO-Lagrangian hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one epoch of BASELINE.json config[1] per GPU: 4096 synthetic envs x 128 steps
(obs 60, act 8), device-resident env, collect -> fused reward/cost GAE -> 40 learning iterations
of 8192 minibatches of 64 (default_cfg of the reference, ppo_lag.py:45-52; KL early stopping
disabled so the work per step is fixed).  N > 1: one process per GPU, 4096 envs per
rank (weak scaling); the per-minibatch gradient all-reduce runs inside the persistent update kernel over
IPC-mapped peer regions (xGMI), or through RCCL between kernels when peer mapping is unavailable.
Prints ONE JSON line (rank 0) with the driver's fields plus
  `roofline`      GAE scan kernel at the headline size (HBM bound by its algorithmic bytes, 33 B per (env, step) + 8 B per
                  path end); the 17 MB buffer is Infinity-Cache resident, which the entry says; `achieved` = bytes / mean
                  per-dispatch duration (HIP events around every launch: the figure rocprofv3 --kernel-trace reports);
  `roofline_hbm_streaming`  the same kernel on a 1.1 GB buffer that cannot sit in the 256 MiB Infinity Cache: the HBM claim;
  `cpu_baseline`  oracle port of the reference loop timed on this box's host cores (rank 0, N=1 only) + the unmodified
                  reference's own figures recorded in the build container (profiles/r03/cpu_reference_timing.json);
  `config3_cpo`   BASELINE config 3 (CPO, same sizes) with its own CPU baseline, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GAE_BYTES_PER_ELEM = 33.0       # 4 f32 in + 1 u8 mask + 4 f32 out
GAE_REPS = 100                   # back-to-back launches per hipGraph replay
GAE_DISPATCHES = 50              # individually event-bracketed launches per timed epoch
PEER_EXCHANGE_USED = [False]     # set by run_epochs on rank 0 (N > 1): which form of the minibatch exchange ran


def _critic_fit_cases():
    """us per 128-row minibatch step of the second-order scripts' critic fit (cpo.py:541-571) at HumanoidVelocity's dims: the
    feature-split persistent kernel with two networks against the launch-per-layer wide path (SPO_WIDE_KS=0)."""
    import ks_cfit_bench
    out = [dict(ks_cfit_bench.one(376, 17, 128 * 2048), kernel="critic_fit_ks_kernel: ONE persistent launch per learning iteration, "
                "2 networks x 6 feature slices = 12 workgroups, a 128-row minibatch as two 64-column chunks (csrc/update_ks.hip)")]
    prev = os.environ.get("SPO_WIDE_KS")
    os.environ["SPO_WIDE_KS"] = "0"
    try:
        out.append(dict(ks_cfit_bench.one(376, 17, 128 * 256), kernel="launch-per-layer wide step (SPO_WIDE_KS=0), HIP-graph replay"))
    finally:
        os.environ.pop("SPO_WIDE_KS", None) if prev is None else os.environ.__setitem__("SPO_WIDE_KS", prev)
    return out


def profile_path(name: str) -> str:
    """The newest committed copy of a profile artefact (profiles/r06, else r05, r04, r03, r02)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "r03", name)


def recorded_port_full_size(algo: str):
    """The oracle port timed ONCE at the full workload size in the build container (profiles/r06/cpu_port_full_size.json, round 6):
    the sample's linearity as a measurement -- same box as reference_recorded's full-size figure."""
    try:
        rec = json.load(open(profile_path("cpu_port_full_size.json")))
        if rec.get("algo") != algo:
            return None
        return {k: rec[k] for k in ("num_envs", "num_steps", "port_env_steps_per_s", "time_rollout_s", "time_update_s", "threads", "box", "date")}
    except Exception:
        return None


def recorded_reference(algo: str):
    """Figures of the UNMODIFIED reference main() recorded in the build container by oracle/time_reference.py (the
    reference tree cannot travel to the GPU box).  Provenance (box, torch, command) is carried along."""
    path = profile_path("cpu_reference_timing.json")
    try:
        recs = [r for r in json.load(open(path)) if r.get("algo") == algo]
    except Exception:
        return None
    return [{k: r.get(k) for k in ("num_envs", "num_steps", "env_steps_per_s", "time_rollout_s", "time_update_s", "threads",
                                   "port_env_steps_per_s", "port_over_reference", "box", "date", "what")} for r in recs] or None


def cpu_baseline(sample_envs: int, T: int, threads: int = 4, algo: str = "ppo_lag"):
    """Oracle port of the reference epoch (oracle/restatement.{ppo_lag,cpo}_epoch_port: per-env Python store loop, per-path
    Python GAE, DataLoader minibatches, torch CPU) on a bounded sample of the same workload: `sample_envs` envs x T steps,
    full default_cfg (ppo_lag: batch 64, 40 iterations; cpo: 33 FVPs, line search, critic fit batch 128 x 10)."""
    from collections import deque
    from oracle import restatement as R
    from oracle.synth_env import SynthEnv
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    env = SynthEnv(sample_envs, 60, 8, seed=0, p_term=0.0, trunc_len=64)
    pol = R.OraclePolicy(60, 8)
    stats = R.StatsLog()
    obs, _ = env.reset()
    obs = torch.as_tensor(obs)
    timers = {}
    dq = (deque(maxlen=50), deque(maxlen=50), deque(maxlen=50))
    acc = (np.zeros(sample_envs), np.zeros(sample_envs), np.zeros(sample_envs))
    t0 = time.time()
    if algo == "ppo_lag":
        cfg = {"gamma": 0.99, "target_kl": float("inf"), "batch_size": 64, "learning_iters": 40}
        R.ppo_lag_epoch_port(env, pol, R.PPOLagUpdater(pol, epochs=1), R.OracleLagrange(25.0, 0.001, 0.035), obs,
                             sample_envs, T, stats, dq, acc, cfg, timers)
        what = "batch 64, 40 learning iters"
    else:
        cfg = {"gamma": 0.99, "target_kl": 0.01, "batch_size": 128, "learning_iters": 10, "cg_iters": 15}
        R.cpo_epoch_port(env, pol, R.CriticFitter(pol), obs, sample_envs, T, stats, dq, acc, cfg, timers=timers)
        what = "15 CG iters (33 FVPs by double backward), line search, critic fit batch 128 x 10 iters"
    wall = time.time() - t0
    steps = sample_envs * T
    host = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    return {"value": round(steps / (timers["rollout"] + timers["update"]), 1), "unit": "env-steps/s", "cores": threads,
            "cores_available": os.cpu_count(), "cores_note": "torch.set_num_threads(4) is the reference's own setting (ppo_lag.py:73, cpo.py:166)",
            "kind": "port",
            "sample": f"1 epoch of {sample_envs} envs x {T} steps (={steps} env-steps), {what}, "
                      f"torch CPU {threads} threads on {host} ({os.cpu_count()} logical cores); "
                      f"rollout {timers['rollout']:.2f}s update {timers['update']:.2f}s wall {wall:.2f}s",
            "reference_recorded": recorded_reference(algo),
            "port_full_size_recorded": recorded_port_full_size(algo)}


def cpu_baseline_mappolag(sample_threads: int, T: int = 64, agents: int = 4, hidden: int = 128, threads: int = 4):
    """Oracle port of one MAPPO-L epoch (oracle/ma_restatement.py: the reference's networks, masked GAE with PopArt and
    MAPPO_L_Trainer.train through torch CPU autograd) on a bounded sample of the config-5 workload: `sample_threads` rollout
    threads x T steps x `agents` agents, mamujoco overrides (hidden 128, 5 full-batch iterations per agent).  Collect = the three
    forward passes per agent and step + sampling; compute = the per-step Python GAE recurrences; train = HAPPO-sequential
    updates.  The synthetic environment and the buffer inserts of the reference Runner are not timed (a lower bound on its time)."""
    from oracle import ma_restatement as MR
    from safepo.multi_agent import mappolag
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    cfg = dict(mappolag.default_cfg)
    cfg.update(mappolag.mamujoco_cfg)
    cfg.update(hidden_size=hidden, episode_length=T, n_rollout_threads=sample_threads)
    D, A = 48, 6
    S = D * agents // 2
    N, nb = sample_threads, 1 + int(cfg["layer_N"])
    trainers = [MR.OracleMATrainer(cfg, MR.MANet(D, hidden, nb, A, True, cfg["std_x_coef"], cfg["std_y_coef"]), MR.MANet(S, hidden, nb, 1, False),
                                   MR.MANet(S, hidden, nb, 1, False)) for _ in range(agents)]
    g = torch.Generator().manual_seed(1)
    t0 = time.time()
    bufs = []
    for a_ in range(agents):
        b = {"share_obs": torch.zeros(T + 1, N, S), "obs": torch.zeros(T + 1, N, D), "actions": torch.zeros(T, N, A),
             "action_log_probs": torch.zeros(T, N, A), "value_preds": torch.zeros(T + 1, N, 1), "cost_preds": torch.zeros(T + 1, N, 1),
             "rewards": torch.zeros(T, N, 1), "costs": torch.zeros(T, N, 1), "masks": torch.ones(T + 1, N, 1),
             "active_masks": torch.ones(T + 1, N, 1), "aver_episode_costs": torch.full((N, 1), 30.0)}
        bufs.append(b)
    for t in range(T + 1):                                    # Runner.collect (+ the bootstrap values of compute() at t = T)
        for a_, tr in enumerate(trainers):
            b = bufs[a_]
            b["obs"][t], b["share_obs"][t] = torch.randn(N, D, generator=g), torch.randn(N, S, generator=g)
            with torch.no_grad():
                b["value_preds"][t], b["cost_preds"][t] = tr.critic(b["share_obs"][t]), tr.cost_critic(b["share_obs"][t])
                if t < T:
                    mean, std = tr.actor(b["obs"][t]), tr.actor.std()
                    act = mean + std * torch.randn(N, A, generator=g)
                    b["actions"][t], b["action_log_probs"][t] = act, MR.log_probs(mean, std, act)
                    b["rewards"][t], b["costs"][t] = torch.randn(N, 1, generator=g), (torch.rand(N, 1, generator=g) < 0.2).float()
    t1 = time.time()
    for a_, tr in enumerate(trainers):                        # Runner.compute
        b = bufs[a_]
        b["returns"] = MR.masked_gae(b["rewards"], b["value_preds"], b["masks"], tr.popart, cfg["gamma"], cfg["gae_lambda"])
        b["cost_returns"] = MR.masked_gae(b["costs"], b["cost_preds"], b["masks"], tr.popart, cfg["gamma"], cfg["gae_lambda"])
    t2 = time.time()
    iters = int(cfg["learning_iters"])
    order = list(range(agents))
    MR.runner_train(trainers, bufs, order, {a_: [torch.randperm(T * N, generator=g) for _ in range(iters)] for a_ in order}, cfg)
    t3 = time.time()
    steps = N * T
    host = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    return {"value": round(steps / (t3 - t0), 1), "unit": "env-steps/s", "cores": threads, "cores_available": os.cpu_count(),
            "cores_note": "torch.set_num_threads(4) is the reference's own setting (mappolag.py:633)", "sample_threads": N, "kind": "port",
            "sample": f"1 epoch of {N} rollout threads x {T} steps x {agents} agents (={steps} env-steps), hidden {hidden}, {iters} full-batch "
                      f"iterations per agent, torch CPU {threads} threads on {host} ({os.cpu_count()} logical cores); collect {t1 - t0:.2f}s "
                      f"GAE {t2 - t1:.2f}s train {t3 - t2:.2f}s (environment and buffer inserts not timed)"}


FP32_MATRIX_PEAK_TFLOPS = 157.3          # dense FP32 MFMA (= packed-FP32 VALU) peak of one MI355X (MI355X_MICROARCH.md)


def time_fp32_kernel(fn, flops, reps, name, dev):
    """Average device time of `fn` (one launch of an FP32-matrix-bound full-batch kernel + its tiny reduction) over `reps`
    back-to-back calls between two HIP events on the launch stream; 3 warm-up calls first."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = flops / (us * 1e-6) / 1e12
    return {"kernel": name, "bound": "mfma", "flops_per_launch": flops, "avg_us": round(us, 2), "achieved": round(tf, 2),
            "unit": "TFLOP/s", "peak": FP32_MATRIX_PEAK_TFLOPS, "frac": round(tf / FP32_MATRIX_PEAK_TFLOPS, 4),
            "note": f"{reps} back-to-back calls of the entry point between two HIP events on the launch stream (the full-batch kernel "
                    "+ its small fixed-order reduction kernel; no host synchronisation inside)"}


def fvp_entry(eng, N, T, D, A, dev):
    """Roofline entry of the Fisher-vector product (cpo.py:132-157; 33 per epoch): algorithmic flops of the analytic J^T M J
    form = actor forward + tangent forward + backward (input and weight gradients) = 4 x the forward's 2*(D*64+64*64+64*A) per row."""
    from safepo import _abi
    v = torch.randn(eng.Pa, device=dev)
    out = torch.empty_like(v)

    def fvp_launch():               # eng.fvp() without its three small torch vector ops
        _abi.check(eng.lib.spo_cpo_fvp(_abi.ptr(eng.policy.theta), _abi.ptr(eng.buffer.data["obs"]), _abi.ptr(v), eng.M, D, A,
                                       _abi.ptr(eng.partial_ws), _abi.ptr(eng.loss_ws), _abi.ptr(out), _abi.stream_ptr()), "spo_cpo_fvp")
    return time_fp32_kernel(fvp_launch, 4 * 2.0 * (D * 64 + 64 * 64 + 64 * A) * N * T, 20,
                            "cpo_actor_kernel<64,MODE_FVP> + fixed-order reduction (spo_cpo_fvp)", dev)


def _lib_note():
    """Which libsafepo_hip.so the run used: 'in-tree' or the absolute path of an SPO_LIB_PATH override."""
    from safepo import _abi
    p = _abi.loaded_library()
    return "in-tree" if p == os.path.abspath(_abi.LIB_PATH) else f"OVERRIDE: {p}"


def self_launch(n: int, one_gpu: bool) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this command as N ranks under torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1 at a free port) and pass the exit code on.  Refuses when the box has fewer than N GPUs
    (unless SPO_BENCH_ONE_GPU=1 puts all ranks on cuda:0) -- a --gpus 8 line must never be a 1-GPU measurement."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < n and not one_gpu:
        print(f"bench.py: --gpus {n} but only {ndev} GPU(s) are visible (SPO_BENCH_ONE_GPU=1 runs all ranks on cuda:0 "
              "as a development aid)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE: spawning {n} ranks: {' '.join(cmd[1:9])} ...", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def run_epochs(algo, a, comm, dev, N, T, D, A, steps, warmup, time_gae):
    """Builds the engine + device env for `algo` and runs `warmup` untimed and `steps` timed epochs bracketed by a
    barrier + device synchronisation on both sides.  Returns a dict with the MAX-over-ranks elapsed time and the pieces
    the JSON line needs."""
    from safepo import _abi
    from safepo.common.engine import PPOLagEngine
    from safepo.common.env import SynthDeviceEnv
    from safepo.common.model import ActorVCritic
    world = comm.world_size
    torch.manual_seed(0)
    policy = ActorVCritic(D, A).to(dev)
    comm.broadcast_(policy.theta, 0)
    if algo == "cpo":
        from safepo.single_agent.cpo import CPOEngine, default_cfg as cpo_cfg
        cfg = dict(cpo_cfg)
        eng = CPOEngine(policy, N, T, cfg, dev, comm=comm)
    else:
        cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": float("inf"), "batch_size": 64,
               "learning_iters": a.learning_iters, "max_grad_norm": 40.0, "dp_batch": a.dp_batch}
        eng = PPOLagEngine(policy, N, T, cfg, dev, comm=comm)
    # a-2 inside the timed region: the env hands out RAW observations and the collect step normalises them as it loads them
    # (running mean / variance merged per step, spo_policy_step_norm) -- what SafeNormalizeObservation does in the
    # reference's env stack (wrappers.py:42-49).  The CPU baselines run without the wrapper (gymnasium is not installed).
    env = SynthDeviceEnv(N, D, A, seed=1234 + comm.rank, p_term=0.0, p_cost=0.1, trunc_len=64, device=dev,
                         normalize_obs=not a.no_normalize_obs)
    rms = env.fuse_normalize(True)
    obs, _ = env.reset()
    lam = 0.001
    gae_graph, gae_disp = [], []

    def epoch(timed: bool):
        nonlocal obs
        t0 = time.time()
        # T x (collect_step -> env.step -> post_step), the loop of safepo/single_agent/_first_order.py: replayed from one HIP
        # graph after its first use (engine.rollout_epoch; SPO_ROLLOUT_GRAPH=0: the eager loop)
        obs = eng.rollout_epoch(env, obs, rms=rms)
        n_ep = eng.drain_episode_events(None)
        torch.cuda.synchronize(dev)
        t1 = time.time()
        if os.environ.get("SPO_BENCH_DUMP_DISPATCHES") == "2" and timed and time_gae and getattr(eng.buffer, "_scan_args", None):
            pre = np.asarray(eng.buffer.time_scan_dispatches(GAE_DISPATCHES)) * 1e6     # development aid: before the update
            print(f"[bench] before update: mean {pre.mean():.2f} median {np.median(pre):.2f} max {pre.max():.2f}", file=sys.stderr)
        if algo == "cpo":
            eng.buffer.compute_gae(None, comm)
            pu = eng.policy_update(-1.0)
            eng.critic_fit()
            eng.buffer.reset()
            out = {"stop_iter": pu["acceptance_step"], "kl": pu["kl"]}
        else:
            out = eng.update(lam)
        if timed and time_gae:
            # GAE scan of THIS epoch's buffer, inside the timed region: (1) every launch between its own pair of HIP events
            # (per-dispatch duration, what rocprofv3 --kernel-trace reports), (2) a hipGraph of GAE_REPS launches
            # between two events (back-to-back throughput; the command processor overlaps dispatch set-up).  ~1 ms.
            gae_disp.extend(eng.buffer.time_scan_dispatches(GAE_DISPATCHES))
            if os.environ.get("SPO_BENCH_DUMP_DISPATCHES") == "2":
                for _ in range(3):
                    again = np.asarray(eng.buffer.time_scan_dispatches(GAE_DISPATCHES)) * 1e6
                    print(f"[bench] after update, again: mean {again.mean():.2f} median {np.median(again):.2f} max {again.max():.2f}", file=sys.stderr)
            gae_graph.append(eng.buffer.time_scan(GAE_REPS))
        torch.cuda.synchronize(dev)
        t2 = time.time()
        return t1 - t0, t2 - t1, out, n_ep

    # N > 1: the in-kernel gradient exchange has a self-test at start-up; should a peer still time out in a full epoch
    # (bounded spins, the error is max-reduced so every rank sees it), all ranks drop to the RCCL form together and the
    # warm-up starts over.  One guard epoch runs even with --warmup 0 so the timed region never hits this first.
    guard = max(warmup, 1) if (world > 1 and getattr(eng, "p2p", None) is not None) else warmup
    done_w = 0
    while done_w < guard:
        try:
            if os.environ.pop("SPO_BENCH_INJECT_PEER_TIMEOUT", "0") == "1" and comm.rank == world - 1:
                eng.sync_ws[8] = 2          # development aid: exercise the fallback below on one rank's error word
            epoch(False)
            done_w += 1
        except _abi.SpoError as e:
            if world == 1 or getattr(eng, "p2p", None) is None:
                raise
            if comm.rank == 0:
                print(f"[bench] {e}; falling back to the RCCL form of the minibatch step", file=sys.stderr)
            eng.drop_peer_exchange()
            eng.buffer.reset()
            obs, _ = env.reset()
            done_w = 0
    comm.barrier()
    torch.cuda.synchronize(dev)
    t_start = time.time()
    roll = upd = 0.0
    last, n_ep = None, 0
    for _ in range(steps):
        r, u, last, n_ep = epoch(True)
        roll += r
        upd += u
    comm.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.time() - t_start
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    comm.all_reduce_max_(tmax)
    # per-rank figures for the line (N > 1): elapsed, rollout and update seconds of every rank
    per_rank = [[elapsed, roll, upd]]
    if world > 1:
        mine = torch.tensor([elapsed, roll, upd], dtype=torch.float64, device=dev)
        both = torch.zeros(world * 3, dtype=torch.float64, device=dev)
        both[3 * comm.rank:3 * comm.rank + 3] = mine
        comm.all_reduce_sum_(both)
        per_rank = both.view(world, 3).tolist()
    px = getattr(eng, "p2p", None)
    PEER_EXCHANGE_USED[0] = px is not None
    exchange = None
    replicas_identical = None
    if world > 1:
        import torch.distributed as dist
        # data-parallel replicas must hold identical bits under whatever exchange form ran: element-wise max == min over the ranks
        th = eng.policy.theta.detach()
        hi, lo = th.clone(), (-th).clone()
        comm.all_reduce_max_(hi); comm.all_reduce_max_(lo)
        replicas_identical = bool(torch.equal(hi, -lo))
        if px is None:
            form = "rccl all-reduce between kernels (spo_ppo_lag_grad -> all_reduce -> spo_clip_adam_then_grad)"
        else:
            cur = int(px.lib.spo_p2p_current_form(world))          # what spo_ppo_lag_update_iter_dp runs (environment / auto-tune / policy)
            how = ("chosen by the start-up auto-tune" if getattr(px, "form", None) is not None else "environment or default policy")
            form = f"in-kernel over IPC-mapped peer regions ({how}): " + px.FORM_NAMES.get(cur, str(cur))
        exchange = {"form": form, "autotune": getattr(eng, "exchange_autotune", None), "selftest_s": round(getattr(px, "last_selftest_s", float("nan")), 4) if px is not None else None,
                    "selftest_result": list(getattr(px, "last_selftest", ())) if px is not None else None,
                    "host_collectives_backend": dist.get_backend(), "host_collectives_world": dist.get_world_size(),
                    "dp_batch": a.dp_batch, "all_ranks_on_one_gpu": os.environ.get("SPO_BENCH_ONE_GPU", "0") == "1"}
    return {"elapsed": float(tmax.item()), "roll": roll, "upd": upd, "last": last, "n_ep": n_ep, "eng": eng, "cfg": cfg,
            "epoch": epoch, "gae_graph": gae_graph, "gae_disp": gae_disp, "per_rank": per_rank, "exchange": exchange,
            "replicas_identical": replicas_identical}


def _update_kernel_entry(us_step, upd_counters, world):
    """The kernel that owns ~99 % of config 2's GPU time, named by the form that ran (SPO_UPDATE_FORM, csrc/update.hip)."""
    form = int(os.environ.get("SPO_UPDATE_FORM", "3"))
    if world == 1 and form >= 3:
        mfma, other = 192, 2860          # per SIMD and step: 72 (optimiser wave) + 120 (column wave) MFMAs; other instructions of both
        return {"kernel": "ppo_update_rs_kernel<64, 2> (persistent; row-split: 3 networks x 2 row groups of 32 rows = 6 co-XCD workgroups, "
                          "4 optimiser + 4 column waves each; one L2 hand-off per layer group and step; csrc/update_rs.hip)",
                "bound": "issue time of a SIMD (matrix + vector instructions of its two waves add) along the step's dependency chain",
                "us_per_minibatch_step": round(us_step, 3),
                "mfma_floor_us": round(mfma * 32 / 2.4e9 * 1e6, 3),
                "frac": round((mfma * 32 / 2.4e9) / (us_step * 1e-6), 4),
                "redo_counters": upd_counters,
                "simd_sum_floor_us": round((mfma * 32 + other * 4) / 2.4e9 * 1e6, 3),
                "frac_of_simd_sum_floor": round(((mfma * 32 + other * 4) / 2.4e9) / (us_step * 1e-6), 4),
                "previous_form_us_per_minibatch_step": 10.51,
                "note": "mfma_floor = the 192 v_mfma_f32_16x16x4_f32 issue slots (32 cycles each) of a SIMD per step at 2.4 GHz: 120 of "
                        "its column wave (half of the output features of 16 of the row group's 32 columns through forward / backward) "
                        "+ 72 of its optimiser wave (weight-gradient products over 32 rows); a SIMD's vector instructions do not "
                        "overlap its matrix instructions (tools/probes/mfma_valu_overlap.hip), so the floor of a step is MFMA cycles + "
                        "~2860 other instructions of the two waves x 4 cycles = simd_sum_floor (instruction counts: tools/"
                        "isa_scratch_map.py on the built kernel); interval budget: profiles/r06/update_phase_cycles_rs.txt; "
                        "previous_form = ppo_update_h_kernel (SPO_UPDATE_FORM=2, one workgroup per network), rounds 2-5; DESIGN.md 3.3.2"}
    return {"kernel": "ppo_update_h_kernel<64> (persistent; 4 main + 4 helper waves per network)"
                      + (" with the in-kernel cross-rank gradient exchange" if world > 1 else ""),
            "bound": "fp32 MFMA issue of one CU per network + per-step latency chain",
            "us_per_minibatch_step": round(us_step, 3),
            "mfma_floor_us": round(368 * 32 / 2.4e9 * 1e6, 3),
            "frac": round((368 * 32 / 2.4e9) / (us_step * 1e-6), 4),
            "redo_counters": upd_counters,
            "simd_sum_floor_us": round((368 * 32 + 2400 * 4) / 2.4e9 * 1e6, 3),
            "frac_of_simd_sum_floor": round(((368 * 32 + 2400 * 4) / 2.4e9) / (us_step * 1e-6), 4),
            "note": "mfma_floor = the 368 MFMA issue slots of a main wave at 2.4 GHz; on this part a SIMD's vector "
                    "instructions do not overlap its matrix instructions (tools/probes/mfma_valu_overlap.hip, "
                    "profiles/r03/mfma_valu_overlap.txt: two waves together take exactly the sum of each alone), so the "
                    "floor of one step on one CU is MFMA cycles + ~2400 VALU instructions of the main and helper wave "
                    "x 4 cycles = simd_sum_floor; DESIGN.md 3.3"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs PER GPU")
    ap.add_argument("--num-steps", type=int, default=128)
    ap.add_argument("--learning-iters", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-envs", type=int, default=512)   # ~80 s of CPU work on the GPU box host (12.5 % of the workload)
    ap.add_argument("--cpo-cpu-sample-envs", type=int, default=256)
    ap.add_argument("--no-config3", action="store_true", help="skip the CPO (BASELINE config 3) section")
    ap.add_argument("--no-config5", action="store_true", help="skip the MAPPO-L (BASELINE config 5 shape) section")
    ap.add_argument("--config5-threads", type=int, default=8192, help="rollout threads of the MAPPO-L section (TOTAL over the ranks)")
    ap.add_argument("--config5-cpu-sample-threads", type=int, default=512, help="rollout threads of the MAPPO-L CPU-baseline sample")
    ap.add_argument("--config5-cpu-all-cores", action="store_true", help="also time the MAPPO-L CPU port with every host core (slow)")
    ap.add_argument("--no-wide", action="store_true", help="skip the wide-network minibatch-step entry")
    ap.add_argument("--no-normalize-obs", action="store_true", help="rollout without the fused observation normaliser (a-2)")
    ap.add_argument("--cpo-steps", type=int, default=3)
    ap.add_argument("--stream-envs", type=int, default=262144, help="extra GAE roofline point that streams from HBM")
    ap.add_argument("--algo", choices=["ppo_lag", "cpo"], default="ppo_lag",
                    help="cpo = time BASELINE config 3 as the main workload (not the headline metric; single GPU)")
    ap.add_argument("--dp-batch", choices=["local", "global"], default="local",
                    help="N > 1: 'local' = batch_size rows per RANK (global batch 64 x N, weak scaling, the headline); "
                         "'global' = batch_size is the GLOBAL minibatch, every rank takes 64 / N rows per step -- the reference's "
                         "arithmetic on the N x num_envs buffer (BASELINE config 4 with exact semantics, SURVEY.md 8(e))")
    a = ap.parse_args()

    # SPO_BENCH_ONE_GPU=1 (development aid): all ranks share cuda:0 with gloo for the host collectives, to exercise the
    # N > 1 code path -- including the in-kernel exchange through IPC-mapped regions -- on a single-GPU box.
    one_gpu = os.environ.get("SPO_BENCH_ONE_GPU", "0") == "1"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, one_gpu))

    from safepo.parallel import init_from_env
    comm = init_from_env(backend="gloo" if one_gpu else None)
    world = comm.world_size
    if world != max(a.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} "
                         "(or run `python bench.py --gpus N` directly: it spawns its own ranks)")
    if not one_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible")
    local_rank = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    N, T, D, A = a.num_envs, a.num_steps, 60, 8

    res = run_epochs(a.algo, a, comm, dev, N, T, D, A, a.steps, a.warmup, time_gae=True)
    elapsed, roll, upd, last, n_ep, eng, cfg = (res[k] for k in ("elapsed", "roll", "upd", "last", "n_ep", "eng", "cfg"))
    res_exchange, res_per_rank = res["exchange"], res["per_rank"]
    res_replicas_identical = res.get("replicas_identical")

    def run_config5():
        try:
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ma_bench
            c5 = ma_bench.run(argparse.Namespace(threads=a.config5_threads, episode_length=64, hidden=128, episodes=2, agents=4),
                              comm=comm, dev=dev)
            # flops-based roofline of the training phase (85 % of the epoch: full-batch fp32 GEMM chains on in-tree MFMA kernels)
            c5["roofline"] = {"bound": "mfma", "achieved": c5["train_gemm_tflops"] * world, "unit": "TFLOP/s",
                              "peak": FP32_MATRIX_PEAK_TFLOPS * world, "frac": round(c5["train_gemm_tflops"] / FP32_MATRIX_PEAK_TFLOPS, 4),
                              "note": "algorithmic GEMM flops of the training phase (per agent: learning_iters x 3 networks x forward + "
                                      "backward = 3 x forward) / its wall time, against the dense FP32 matrix peak; tools/ma_bench.py"}
            if world == 1 and comm.rank == 0 and not a.no_cpu_baseline:
                # the GPU side runs 8 192 rollout threads, the CPU port a bounded sample -- so the port is timed at
                # TWO sample sizes (its env-steps/s must not depend on the size for the ratio to mean anything) and once more
                # with every host core, and all three figures are in the line next to the ratio
                n_s = a.config5_cpu_sample_threads
                c5["cpu_baseline"] = cpu_baseline_mappolag(n_s)
                half = cpu_baseline_mappolag(max(n_s // 2, 1))
                c5["cpu_baseline"]["value_at_half_the_sample"] = half["value"]
                c5["speedup_vs_cpu_baseline"] = round(c5["env_steps_per_s"] / c5["cpu_baseline"]["value"], 1)
                c5["speedup_note"] = (f"GPU: {a.config5_threads} rollout threads; CPU port: {n_s}-thread sample at the reference's 4 torch "
                                      f"threads of {os.cpu_count()} logical cores (and at {max(n_s // 2, 1)} threads: same rate = the "
                                      f"ratio carries to the full size)")
                if a.config5_cpu_all_cores:     # opt-in: with ~200 torch threads on these small products the port takes minutes
                    allc = cpu_baseline_mappolag(n_s, threads=os.cpu_count() or 4)
                    c5["cpu_baseline"]["all_host_cores"] = {"value": allc["value"], "cores": allc["cores"], "sample": allc["sample"]}
                    c5["speedup_vs_cpu_baseline_all_host_cores"] = round(c5["env_steps_per_s"] / allc["value"], 1)
            return c5
        except Exception as e:  # pragma: no cover
            return {"error": str(e)[:300]}
    # BASELINE config 5 on N > 1: every rank runs its shard of the MAPPO-L Runner (collective calls inside), so this sits
    # BEFORE the non-zero ranks leave; rank 0 reports it below.  (N = 1: after the CPO section, as before.)
    config5 = run_config5() if (world > 1 and a.algo == "ppo_lag" and not a.no_config5) else None
    if comm.rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    total_env_steps = world * N * T * a.steps
    value = total_env_steps / elapsed
    batch = 128 if a.algo == "cpo" else (64 // world if (a.dp_batch == "global" and world > 1) else 64)
    iters = 10 if a.algo == "cpo" else a.learning_iters
    n_mb = (N * T + batch - 1) // batch
    seg_ends = int(eng.buffer.seg_end.sum().item())
    folded = bool(getattr(eng.buffer, "last_scan_folded", False))
    # folded form (the engine's): gamma * bootstrap already sits in the reward / cost arrays the scan reads -> exactly
    # 33 B per element; with separate bootstrap arrays 8 more bytes are algorithmic per path end
    gae_bytes = GAE_BYTES_PER_ELEM * N * T + (0.0 if folded else 8.0 * seg_ends)
    disp = np.asarray(res["gae_disp"], np.float64)
    graph = np.asarray(res["gae_graph"], np.float64)
    disp_avg = float(disp.mean()) if disp.size else float("nan")
    if os.environ.get("SPO_BENCH_DUMP_DISPATCHES"):          # development aid: the raw per-dispatch series
        print("[bench] per-dispatch us:", np.round(disp * 1e6, 2).tolist(), file=sys.stderr)
    achieved = gae_bytes / disp_avg / 1e9
    pmc = None
    try:
        pj = json.load(open(profile_path("gae_pmc.json")))
        hit = [l for l in pj["launches"] if l["num_envs"] == N and l["folded"] == folded and T == 128]
        if hit:
            pmc = {"hbm_bytes_per_launch": round(hit[0]["hbm_bytes"]), "traffic_over_algorithmic": hit[0]["traffic_over_algorithmic"],
                   "source": os.path.relpath(profile_path("gae_pmc.json"), ROOT) + ": " + pj["source"] + "; " + pj["corrections"]}
    except Exception:
        pass
    roofline = {"kernel": "gae_kernel (spo_gae_fused" + (", folded bootstrap form)" if folded else ")"), "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "achieved_event_pairs": round(achieved, 1), "frac_event_pairs": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
                "traffic_source": (pmc["source"] if pmc else "no committed --pmc measurement for this shape"),
                "traffic_profiled": pmc,
                "bytes_per_launch": gae_bytes, "avg_launch_us": round(disp_avg * 1e6, 3),
                "median_launch_us": round(float(np.median(disp)) * 1e6, 3) if disp.size else None,
                "launches_timed": int(disp.size),
                "graph_avg_launch_us": round(float(graph.mean()) * 1e6, 3) if graph.size else None,
                "graph_frac": round(gae_bytes / float(graph.mean()) / 1e9 / HBM_PEAK_GBS, 4) if graph.size else None,
                "residency": f"{gae_bytes / 1e6:.1f} MB per launch: Infinity-Cache resident (256 MiB), re-read by every timed "
                             "launch -- NOT an HBM-streaming figure; see roofline_hbm_streaming for the HBM claim",
                "note": "achieved = algorithmic bytes / MEAN per-dispatch duration over the timed epochs (200 untimed dispatches first: "
                        "power-management transient after a light-load phase, DESIGN.md 3.1): every dispatch carries its own "
                        "start/stop HIP events on the launch stream (hipExtLaunchKernelGGL: the dispatch packet's timestamps). "
                        "rocprof_* = the committed rocprofv3 --kernel-trace of this same command (profiles/r04/"
                        "gae_dispatch_durations.json, same kernel and grid): the profiler's own per-dispatch average is ~8 % "
                        "higher than the unprofiled event pairs, and under the profiler the event pairs themselves read ~2x "
                        "(profiles/r04/bench_profiled_line.json), so the two figures cannot come from one run. graph_* = hipGraph of "
                        f"{GAE_REPS} back-to-back launches between two events (dispatch set-up overlapped). traffic: PMC "
                        "counters cannot be read inside this run: traffic / traffic_profiled are the committed rocprofv3 "
                        "--pmc measurement (FETCH_SIZE x2 + WRITE_SIZE, separate passes) with its source. frac / achieved: from the "
                        "committed rocprofv3 average when one exists for this grid (frac_source), else the event pairs"}
    try:
        dj = json.load(open(profile_path("gae_dispatch_durations.json")))
        want = f"grid={2 * N * 32}"
        hit = [v for k, v in dj["per_kernel_and_grid_size"].items() if k.endswith(want)] if (T == 128 and folded) else []
        if hit:
            roofline["rocprof_avg_launch_us"] = hit[0]["avg_us"]
            roofline["rocprof_launches"] = hit[0]["launches"]
            roofline["frac_at_rocprof_avg"] = round(gae_bytes / (hit[0]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            # the headline figure is the one that follows from the committed rocprofv3 summary of this command; the
            # live event-pair measurement of THIS run stays beside it (achieved_event_pairs / frac_event_pairs)
            roofline["achieved"] = round(gae_bytes / (hit[0]["avg_us"] * 1e-6) / 1e9, 1)
            roofline["frac"] = roofline["frac_at_rocprof_avg"]
            roofline["frac_source"] = os.path.relpath(profile_path("gae_dispatch_durations.json"), ROOT) + " (rocprofv3 --kernel-trace of this command)"
    except Exception:
        pass

    # extra roofline points (untimed, after the run): 32 768 envs (138 MB) and a buffer that cannot sit in
    # the 256 MiB Infinity Cache
    def scan_point(Ns, reps):
        from safepo.common.buffer import VectorizedOnPolicyBuffer
        from safepo.common.engine import _Space
        big = VectorizedOnPolicyBuffer(_Space(1), _Space(1), size=T, num_envs=Ns, device=dev)
        for k in ("reward", "cost", "value_r", "value_c"):
            big.data[k].normal_()
        big.seg_end[:, T - 1] = 1
        big.seg_end[:, T // 2 - 1] = 1
        # the engine's layout: bootstrap folded into the arrays the scan reads (boot arrays are zero here: fold == raw)
        big.reward_fold.copy_(big.data["reward"]); big.cost_fold.copy_(big.data["cost"])
        big.ptr, big._fold_cols = T, T
        big.compute_gae(None)
        assert big.last_scan_folded
        d_s = float(np.mean(big.time_scan_dispatches(reps)))
        g_s = big.time_scan(reps)
        b = GAE_BYTES_PER_ELEM * Ns * T
        del big
        out = {"num_envs": Ns, "bytes_per_launch": b, "avg_launch_us": round(d_s * 1e6, 1),
               "achieved": round(b / d_s / 1e9, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": round(b / d_s / 1e9 / HBM_PEAK_GBS, 4),
               "graph_avg_launch_us": round(g_s * 1e6, 1), "graph_frac": round(b / g_s / 1e9 / HBM_PEAK_GBS, 4)}
        # like the headline point: frac / achieved follow from the committed rocprofv3 per-dispatch average of
        # this command when the summary holds this grid; the in-run event pairs stay beside it
        try:
            dj_ = json.load(open(profile_path("gae_dispatch_durations.json")))
            hit_ = [v for k, v in dj_["per_kernel_and_grid_size"].items() if k.endswith(f"grid={2 * Ns * 32}") or k.endswith(f"grid={Ns * 32}")]
            if hit_ and T == 128:
                out["achieved_event_pairs"], out["frac_event_pairs"] = out["achieved"], out["frac"]
                out["rocprof_avg_launch_us"], out["rocprof_launches"] = hit_[0]["avg_us"], hit_[0]["launches"]
                out["achieved"] = round(b / (hit_[0]["avg_us"] * 1e-6) / 1e9, 1)
                out["frac"] = round(out["achieved"] / HBM_PEAK_GBS, 4)
                out["frac_source"] = os.path.relpath(profile_path("gae_dispatch_durations.json"), ROOT) + " (rocprofv3 --kernel-trace of this command)"
        except Exception:
            pass
        return out
    stream, mid = None, None
    try:
        mid = scan_point(32768, 20)
        stream = scan_point(a.stream_envs, 10)
        stream["residency"] = "1.1 GB per launch > 256 MiB Infinity Cache: streams from HBM3E -- the HBM-roofline claim of this kernel"
    except Exception as e:  # pragma: no cover
        stream = {"error": str(e)[:200]}

    # the reference-faithful epoch (KL early stopping at default_cfg's target_kl = 0.02), untimed extra
    faithful = None
    if world == 1 and a.algo == "ppo_lag":
        cfg["target_kl"] = 0.02
        r_f, u_f, out_f, _ = res["epoch"](False)
        cfg["target_kl"] = float("inf")
        faithful = {"target_kl": 0.02, "stop_iter": out_f["stop_iter"], "kl": out_f["kl"], "s_per_epoch": round(r_f + u_f, 4),
                    "env_steps_per_s": round(N * T / (r_f + u_f), 1)}

    # FP32-matrix-bound kernels of the path: the full-batch KL of the early-stop test.  Algorithmic flops =
    # the actor forward, 2 * (D*64 + 64*64 + 64*A) per row; peak = the dense FP32 matrix rate (MI355X_MICROARCH.md)
    kl_entry = None
    if a.algo == "cpo" and world == 1:
        kl_entry = fvp_entry(eng, N, T, D, A, dev)        # --algo cpo: the Fisher-vector product takes this slot
    if a.algo == "ppo_lag" and world == 1:
        from safepo import _abi

        def kl_launch():            # the launches of eng.kl_to_old() without its host read-back of the sum
            _abi.check(eng.lib.spo_actor_kl(_abi.ptr(eng.policy.theta), _abi.ptr(eng.buffer.data["obs"]), _abi.ptr(eng.mean_old),
                                            _abi.ptr(eng.logstd_old), _abi.ptr(eng.kl_partials), eng.kl_partials.numel(),
                                            _abi.ptr(eng.kl_sum), eng.M, D, A, _abi.stream_ptr()), "spo_actor_kl")
        kl_entry = time_fp32_kernel(kl_launch, 2.0 * (D * 64 + 64 * 64 + 64 * A) * N * T, 20,
                                    "actor_full_kernel<64,1> (spo_actor_kl: full-batch actor forward + KL(old || new), "
                                    "ppo_lag.py:338-345)", dev)

    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.cpu_sample_envs if a.algo == "ppo_lag" else a.cpo_cpu_sample_envs, T, algo=a.algo)

    # BASELINE config 3 (CPO) in the same line: its own engine, env, timed epochs and CPU baseline (N = 1 only)
    config3 = None
    if world == 1 and a.algo == "ppo_lag" and not a.no_config3:
        del res, eng
        torch.cuda.empty_cache()
        try:
            r3 = run_epochs("cpo", a, comm, dev, N, T, D, A, a.cpo_steps, 1, time_gae=False)
            v3 = N * T * a.cpo_steps / r3["elapsed"]
            n_mb3 = (N * T + 127) // 128
            config3 = {"workload": (f"cpo synthetic env (obs=60, act=8), num_envs={N}, num_steps={T}, default_cfg (2 surrogate "
                                    "gradients, 2 x 15 CG iters = 33 Fisher-vector products, line search, critic fit batch 128 x 10 "
                                    "iters), device-resident env"),
                       "value": round(v3, 1), "unit": "env-steps/s", "steps": a.cpo_steps, "warmup": 1,
                       "ms_per_step": round(r3["elapsed"] / a.cpo_steps * 1e3, 2),
                       "rollout_s_per_epoch": round(r3["roll"] / a.cpo_steps, 4),
                       "update_s_per_epoch": round(r3["upd"] / a.cpo_steps, 4),
                       "critic_minibatch_steps_per_epoch": n_mb3 * 10,
                       "acceptance_step": r3["last"]["stop_iter"], "kl": r3["last"]["kl"],
                       "critic_fit_form": ("one persistent launch, two workgroup pairs (64 of every 128 rows each) exchanging in-kernel"
                                           if getattr(r3["eng"], "_split", None) else "one persistent launch")}
            config3["cpo_fvp"] = fvp_entry(r3["eng"], N, T, D, A, dev)
            if not a.no_cpu_baseline:
                config3["cpu_baseline"] = cpu_baseline(a.cpo_cpu_sample_envs, T, algo="cpo")
                config3["speedup_vs_cpu_baseline"] = round(v3 / config3["cpu_baseline"]["value"], 1)
        except Exception as e:  # pragma: no cover
            config3 = {"error": str(e)[:300]}

    # BASELINE config 5 (MAPPO-L shape: 4 agents, obs 48, 8 192 rollout threads x 64 steps; one GPU here, the row is
    # "mappo_lag ... 8 x MI355X" in BASELINE.json): the multi-agent Runner of the f3 row, driver-visible (N = 1 only)
    # With --gpus N the 8 192 rollout threads are sharded over the ranks (the BASELINE row itself: strong scaling) and the section
    # already ran above, before the non-zero ranks left.
    if world == 1 and a.algo == "ppo_lag" and not a.no_config5:
        config5 = run_config5()

    us_step = upd / a.steps / (n_mb * iters) * 1e6
    upd_counters = None
    if a.algo == "ppo_lag":                      # (N > 1: rank 0's process)
        import ctypes
        from safepo import _abi
        c4 = (ctypes.c_ulonglong * 4)()
        if _abi.load().spo_debug_update_counters(c4, 0) == 0 and c4[0]:
            upd_counters = {"steps_counted": int(c4[0]), "clipped_after_speculation_redone": int(c4[1]),
                            "clipped_under_conservative_protocol": int(c4[2]), "steps_under_conservative_protocol": int(c4[3]),
                            "note": "all launches of this process (warm-up, timed epochs, early-stopping epoch); a redone step costs "
                                    "one repeated forward of the main waves"}
    # the fallback path for policies outside the persistent kernels' envelope (any hidden_sizes / dims): us per 64-row
    # minibatch step of the wide-network engine -- a widened net and HumanoidVelocity's dims (~1 s)
    wide_entry = None
    if world == 1 and a.algo == "ppo_lag" and not a.no_wide:
        try:
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import wide_bench
            wide_entry = {"what": "one 64-row minibatch step of WidePPOLagEngine.learning_iter (gather, 3 forwards, loss, 3 backwards, joint "
                                  "clip + Adam) outside the 3-workgroup persistent kernel's dims (its step is update_kernel above): hidden "
                                  "[64, 64] with obs_dim <= 512 / act_dim <= 32 takes the persistent feature-split kernel, anything else "
                                  "the row-group gradient kernel (csrc/mlp_rows.hip, round 6: one launch for gather + forwards + "
                                  "loss + backwards, two for clip + Adam; rounds 4-5: launch-per-network, 89 us at [128, 128])",
                          "cases": [dict(wide_bench.one([128, 128], 64, 256), obs_dim=60, act_dim=8),
                                    dict(wide_bench.one([64, 64], 64, 4096, D=376, A=17), obs_dim=376, act_dim=17),
                                    dict(wide_bench.one([64, 64], 64, 256, D=376, A=17, force_wide=True), obs_dim=376, act_dim=17,
                                         note="SPO_WIDE_KS=0: the path rounds 3-4 took at these dims")],
                          "critic_fit_cases": _critic_fit_cases()}
        except Exception as e:  # pragma: no cover
            wide_entry = {"error": str(e)[:300]}
    line = {
        "metric": "env-steps/sec (collect+GAE+update) at num_envs=4096, 1/2/4/8 GPU",
        "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"ppo_lag synthetic env (obs=60, act=8), num_envs={N} per GPU, num_steps={T}, "
                                f"default_cfg batch 64 {'per rank' if a.dp_batch == 'local' or world == 1 else 'GLOBAL (64/N rows per rank)'} "
                                f"x {a.learning_iters} learning iters (target_kl=inf: all iters run), "
                                f"device-resident env, observation normalisation fused into the collect step") if a.algo == "ppo_lag" else
                               (f"cpo synthetic env (obs=60, act=8), num_envs={N}, num_steps={T}, default_cfg "
                                f"(15 CG iters, 33 FVPs, line search, critic fit batch 128 x 10 iters), device-resident env"),
                   "global_envs": world * N,
                   "parallelism": (f"dp{world} over num_envs, per-minibatch gradient all-reduce "
                                   + ("inside the persistent update kernel over IPC-mapped peer regions (xGMI)"
                                      if PEER_EXCHANGE_USED[0] else "via RCCL between kernels"))
                   if world > 1 else "single GPU",
                   "minibatch_steps_per_epoch": n_mb * iters},
        "roofline": roofline,
        "roofline_32768_envs": mid,
        "roofline_hbm_streaming": stream,
        "early_stopping_epoch": faithful,
        # the kernel that owns 99 % of the GPU time is not HBM- but latency/matrix-bound: 327 680 strictly sequential
        # optimiser steps, each at least 368 v_mfma_f32_16x16x4_f32 (32 cycles each) per wave on one CU per network
        "update_kernel": (_update_kernel_entry(us_step, upd_counters, world)
                          if a.algo == "ppo_lag" else None),
        ("kl_kernel" if a.algo == "ppo_lag" else "cpo_fvp"): kl_entry,
        "cpu_baseline": cpu,
        "config3_cpo": config3,
        "config5_mappolag": config5,
        "wide_minibatch_step": wide_entry,
        "library": _lib_note(),
        "exchange": res_exchange,
        "replicas_identical_after_run": res_replicas_identical,
        "per_rank": ([{"rank": r, "ms_per_step": round(e / a.steps * 1e3, 2), "rollout_s_per_epoch": round(ro / a.steps, 4),
                       "update_s_per_epoch": round(u / a.steps, 4),
                       "update_us_per_minibatch_step": round(u / a.steps / (n_mb * iters) * 1e6, 3)}
                      for r, (e, ro, u) in enumerate(res_per_rank)] if world > 1 else None),
        "phases": {"rollout_s_per_epoch": round(roll / a.steps, 4), "update_s_per_epoch": round(upd / a.steps, 4),
                   "update_us_per_minibatch_step": round(us_step, 3),
                   "stop_iter": last["stop_iter"], "kl": last["kl"], "episodes_per_epoch": n_ep},
    }
    if cpu:
        line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
    print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
