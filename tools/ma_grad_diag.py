"""Diagnostic (not a test): where the config-5-shape MAPPO-L update differs from the float64 oracle -- per parameter group, the
pre-clip gradient of step 1 (HIP vs fp32 oracle vs fp64 oracle) and the logged scalars of the first steps."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(rows_log2=19, steps=2):
    import ma_yardstick as Y
    import test_gpu_ma_full_size as F
    from oracle import ma_restatement as MR
    from safepo.multi_agent.mappolag import MAPPO_L_Policy, MAPPO_L_Trainer
    dev = torch.device("cuda:0")
    rows = 1 << rows_log2
    F.THREADS = rows // F.T
    torch.manual_seed(5)
    cfg = F._cfg(dev)
    cfg["n_rollout_threads"] = F.THREADS
    pol = MAPPO_L_Policy(cfg, F._Sp(F.D), F._Sp(F.S), F._Sp(F.A))
    with torch.no_grad():
        for net in pol.networks():
            net.theta.add_(0.05 * torch.randn_like(net.theta))
    nets0 = Y.nets_like(pol, cfg["std_x_coef"], cfg["std_y_coef"])
    s = F._sample(rows, seed=11)
    with torch.no_grad():
        lp = MR.log_probs(nets0["actor"](s["obs"]), nets0["actor"].std(), s["actions"])
        s["old_logp"] = lp + 0.03 * torch.randn(rows, F.A, generator=torch.Generator().manual_seed(3))
    tr = MAPPO_L_Trainer(cfg, pol)
    sample = (s["share_obs"], s["obs"], None, None, s["actions"], s["value_preds"], s["returns"], None, s["active_masks"],
              s["old_logp"], s["adv"], None, s["factor"], s["cost_preds"], s["cost_returns"], None, s["cost_adv"], s["aver_episode_costs"])
    sample = tuple(t.to(dev) if torch.is_tensor(t) else t for t in sample)
    opts = {"actor": pol.actor_optimizer, "critic": pol.critic_optimizer, "cost_critic": pol.cost_optimizer}
    rows_hip, grads = [], {}
    for k in range(steps):
        vl, cgn, plo, ent, agn, imp, cl, cogn = tr.ppo_update(sample)
        torch.cuda.synchronize()
        vn = tr.value_normalizer
        rows_hip.append([vl.item(), cgn.item(), plo.item(), ent.item(), agn.item(), imp.detach().mean().item(), cl.item(), cogn.item(),
                         float(tr.lamda_lagr), float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        if k == 0:
            grads = {nm: o.grad.double().cpu().numpy().copy() for nm, o in opts.items()}
            theta1 = {nm: net.theta.double().cpu().numpy().copy() for nm, net in (("actor", pol.actor), ("critic", pol.critic), ("cost_critic", pol.cost_critic))}
            lam1 = float(tr.lamda_lagr)
    r32, _, sn32 = Y.oracle_steps(cfg, nets0, s, "mappolag", steps, torch.float32, snapshots=(1,))
    r64, _, sn64 = Y.oracle_steps(cfg, nets0, s, "mappolag", steps, torch.float64, snapshots=(1,))
    # parameters after step 1, and the float64 oracle's step-2 policy loss evaluated AT the HIP parameters: separates an
    # optimiser-step deviation from a forward deviation
    for nm in ("actor", "critic", "cost_critic"):
        dh, d3 = np.abs(theta1[nm] - sn64[1][nm]), np.abs(sn32[1][nm] - sn64[1][nm])
        print(f" theta after step 1 {nm:12s}: |hip-f64| max {dh.max():.2e} rms {np.sqrt((dh ** 2).mean()):.2e}   |f32-f64| max {d3.max():.2e} rms {np.sqrt((d3 ** 2).mean()):.2e}")
    import copy
    def pl_at(theta_actor, lam, dtype):
        net = Y.to_dtype(nets0["actor"], dtype)
        off = 0
        with torch.no_grad():
            for prm in net.ordered_parameters():
                n = prm.numel()
                prm.copy_(torch.from_numpy(theta_actor[off:off + n]).view_as(prm).to(dtype))
                off += n
            sd = Y.to_dtype(s, dtype)
            mean = net(sd["obs"])
            logp = MR.log_probs(mean, net.std(), sd["actions"])
            imp = torch.prod(torch.exp(logp - sd["old_logp"]), dim=-1, keepdim=True)
            adv_h = sd["adv"] - lam * sd["cost_adv"]
            m = torch.sum(sd["factor"] * torch.min(imp * adv_h, torch.clamp(imp, 0.8, 1.2) * adv_h), dim=-1, keepdim=True)
            return float((-m * sd["active_masks"]).sum() / sd["active_masks"].sum()), float(imp.mean())
    print(" step-2 policy loss / ratio of the f64 oracle AT the HIP parameters after step 1:", pl_at(theta1["actor"], lam1, torch.float64))
    print(" ... f32 oracle AT the HIP parameters:", pl_at(theta1["actor"], lam1, torch.float32))
    print(" ... f64 oracle at the f32 oracle's parameters:", pl_at(sn32[1]["actor"], r32[0]["row"][8], torch.float64))
    names = ("value_loss", "critic_gn", "policy_loss", "entropy", "actor_gn", "ratio", "cost_loss", "cost_gn", "lamda", "pa_mean", "pa_sq", "pa_deb")
    print(f"rows {rows}")
    for k in range(steps):
        print(f" step {k + 1}")
        for c, nm in enumerate(names):
            h, a, b = rows_hip[k][c], r32[k]["row"][c], r64[k]["row"][c]
            print(f"   {nm:12s} hip {h:+.9e} f32 {a:+.9e} f64 {b:+.9e}  |hip-f64| {abs(h - b):.2e}  |f32-f64| {abs(a - b):.2e}")
    gkey = {"actor": "actor_grad", "critic": "critic_grad", "cost_critic": "cost_grad"}
    for nm, net in (("actor", pol.actor), ("critic", pol.critic), ("cost_critic", pol.cost_critic)):
        g32, g64, gh = r32[0][gkey[nm]].double().numpy(), r64[0][gkey[nm]].double().numpy(), grads[nm]
        off, groups = 0, []
        for pname, p in zip(net.state_dict().keys(), net.state_dict().values()):
            groups.append((pname, off, off + p.numel()))
            off += p.numel()
        print(f" {nm}: |g|max {np.abs(g64).max():.3e}")
        for pname, lo, hi in groups:
            dh, d3, sc = np.abs(gh[lo:hi] - g64[lo:hi]).max(), np.abs(g32[lo:hi] - g64[lo:hi]).max(), np.abs(g64[lo:hi]).max()
            print(f"   {pname:34s} scale {sc:.2e}  |hip-f64| {dh:.2e}  |f32-f64| {d3:.2e}  ratio {dh / max(d3, 1e-30):8.1f}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 19, int(sys.argv[2]) if len(sys.argv) > 2 else 2)
