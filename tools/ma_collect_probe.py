"""Where a MAPPO-L collect step goes (config-5 shape): collect / env.step / insert, host time and synchronised time per step,
and the replay time of the captured collect graph alone.  FUSED=0 selects the per-network launches (A/B)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
    sys.path.insert(0, p)
from safepo.multi_agent import mappolag
from safepo.common.env import SynthMultiAgentEnv
dev = torch.device("cuda:0")
cfg = dict(mappolag.default_cfg); cfg.update(mappolag.mamujoco_cfg)
cfg.update(device="cuda:0", n_rollout_threads=8192, episode_length=64, hidden_size=128, log_dir="/tmp/ma_probe", seed=0,
           env_name="SynthMultiAgent-v0", use_eval=False)
cfg["collect_fused"] = os.environ.get("FUSED", "1") == "1"
env = SynthMultiAgentEnv(8192, num_agents=4, obs_dim=48, act_dim=6, trunc_len=64, device=dev)
r = mappolag.Runner(env, None, cfg); r.logger.verbose = False; r.warmup()
def sync(): torch.cuda.synchronize()
T = {"collect": 0, "collect_cpu": 0, "env": 0, "env_cpu": 0, "insert": 0, "insert_cpu": 0}
for ep in range(3):
    for step in range(64):
        sync(); t0 = time.perf_counter()
        out = r.collect(step); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
        values, actions, lps, rnn, rnn_c, cps, rnn_k = out
        res = env.step(actions); t3 = time.perf_counter(); sync(); t4 = time.perf_counter()
        obs, share_obs, rewards, costs, dones, infos, _ = res
        r.insert((obs, share_obs, rewards, costs, dones, infos, values, actions, lps, rnn, rnn_c, cps, rnn_k, costs.mean()))
        t5 = time.perf_counter(); sync(); t6 = time.perf_counter()
        if ep > 0:
            T["collect"] += t2 - t0; T["collect_cpu"] += t1 - t0; T["env"] += t4 - t2; T["env_cpu"] += t3 - t2
            T["insert"] += t6 - t4; T["insert_cpu"] += t5 - t4
    r.compute(); r.train()
print({k: round(v / 128 * 1e3, 4) for k, v in T.items()}, "ms per step")
# graph replay alone
g = r._step_graphs[0][0] if getattr(r, "_step_graphs", None) else r._graph
sync(); t0 = time.perf_counter()
for _ in range(200): g.replay()
sync(); print("graph replay ms", (time.perf_counter() - t0) / 200 * 1e3)
