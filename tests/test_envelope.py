"""CPU checks of the drift-envelope gate itself (tests/envelope.py): an independent but correct fp32 evaluation order
passes it, small real defects do not.  The GPU tests apply the same gate to the HIP trajectory."""
import numpy as np
import torch

import envelope as E
from oracle import restatement as R


def _problem(M, D, A, seed):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(M, D, generator=g)
    act = torch.randn(M, A, generator=g)
    logp = -A * 0.9 - 0.5 * (act ** 2).sum(-1) + 0.1 * torch.randn(M, generator=g)
    return obs, act, logp, torch.randn(M, generator=g), torch.rand(M, generator=g), torch.randn(M, generator=g)


def test_envelope_accepts_reordered_fp32_and_rejects_small_defects():
    K, D, A = 192, 60, 8
    M = 64 * K
    torch.manual_seed(3)
    sd = R.OraclePolicy(D, A).state_dict()
    prob = _problem(M, D, A, 77)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(5))
    ks = (8, 64, K)
    l32, t32 = E.oracle_trajectory(sd, prob, perm, 64, K, torch.float32, ks)
    l64, t64 = E.oracle_trajectory(sd, prob, perm, 64, K, torch.float64, ks)
    # (1) the same networks with renumbered hidden units: every dot product is summed in another order
    sd2, unperm = E.permuted_hidden_state(sd, seed=1)
    lp, tp = E.oracle_trajectory(sd2, prob, perm, 64, K, torch.float32, ks)
    assert not np.array_equal(lp, l32)                        # it really is a different rounding sequence
    E.assert_loss_envelope(lp, l32, l64, "reordered fp32")
    for k in ks:
        E.assert_theta_envelope(unperm(tp[k]), t32[k], t64[k], f"reordered fp32, {k} steps")
    # (2) a stale minibatch at one step (what a broken index prefetch would do)
    bad = perm.clone()
    bad[100 * 64:101 * 64] = perm[99 * 64:100 * 64]
    lb, tb = E.oracle_trajectory(sd, prob, bad, 64, K, torch.float32, ks)
    assert E.loss_envelope(lb, l32, l64)[0] > 100.0
    assert E.theta_envelope(tb[K], t32[K], t64[K])[0] > 10.0
    # (3) a 0.1 % error in the actor's step size from step 0 (what a wrong bias correction would do)
    ls, ts = E.oracle_trajectory(sd, prob, perm, 64, K, torch.float32, ks, lr_factor=1.001)
    assert E.theta_envelope(ts[K], t32[K], t64[K])[0] > 3.0


def test_reference_trace_is_inside_its_own_envelope(golden_dir):
    """The fp32 oracle replays the reference's recorded ppo_lag.main() bit for bit, and the float64 replay of the same
    inputs stays ~1e-7 away: that distance is the yardstick the GPU trace test uses."""
    import os
    z = np.load(os.path.join(golden_dir, "ppo_lag_trace.npz"))
    r32 = E.replay_ppo_lag_trace(z, torch.float32)
    r64 = E.replay_ppo_lag_trace(z, torch.float64)
    names = [k[len("init_sd_"):] for k in z.files if k.startswith("init_sd_")]
    ref_final = np.concatenate([z["final_sd_" + k].reshape(-1) for k in names]).astype(np.float64)
    assert np.array_equal(r32["theta_final"], ref_final)
    for e in range(int(z["meta_epochs"])):
        assert np.array_equal(r32["losses"][e], z[f"e{e}_mb_losses"])
    d = np.abs(ref_final - r64["theta_final"])
    assert 0 < d.max() < 1e-6


def test_adam_noise_directions_are_singled_out_by_the_reference_and_defects_still_fail(golden_dir):
    """tests/envelope.py::adam_noise_directions on the reference's own run at HumanoidVelocity's dims
    (ppo_lag_trace_humanoid.npz, round 5): after the first epoch ONE of the 86 116 parameters -- a first-layer weight of the
    cost critic whose gradient is of the size of Adam's eps -- sits ~5e-6 from the float64 replay in the reference's recorded
    parameters while the rest sit ~1e-8; the gate finds exactly such elements from the reference's behaviour, lets another
    implementation be up to 10 x as far on THEM, and keeps its power everywhere else: a real defect (every element moved by
    5e-7, far below the noise direction's own excursion) still fails with the option on."""
    import os
    z = np.load(os.path.join(golden_dir, "ppo_lag_trace_humanoid.npz"))
    r64 = E.replay_ppo_lag_trace(z, torch.float64)
    names = [k[len("init_sd_"):] for k in z.files if k.startswith("init_sd_")]
    f32 = np.concatenate([z["e1_sd_before_" + k].reshape(-1) for k in names]).astype(np.float64)     # the reference's own float32 run
    f64 = r64["theta_before"][1]
    d32 = np.abs(f32 - f64)
    mask, _ = E.adam_noise_directions(d32, d32)
    assert 1 <= mask.sum() <= 9 and d32[mask].min() > 25 * np.sqrt(np.mean(d32[~mask] ** 2))
    worst = int(np.argmax(d32))
    assert mask[worst] and d32[worst] > 1e-6 > 50 * np.median(d32)
    # another correct implementation: the reference's values, 6 x further out on the noise direction only
    other = f32.copy()
    other[worst] = f64[worst] + 6.0 * (f32[worst] - f64[worst])
    assert E.theta_envelope(other, f32, f64)[0] > 1.0                                   # the plain gate: one element sinks the L2 norm
    assert E.theta_envelope(other, f32, f64, noise_directions=True)[0] <= 1.0
    other[worst] = f64[worst] + 12.0 * (f32[worst] - f64[worst])                         # ... but not arbitrarily far
    assert E.theta_envelope(other, f32, f64, noise_directions=True)[0] == float("inf")
    # a defect of 5e-7 on every element is an order of magnitude SMALLER than the noise direction's excursion and still fails
    rng = np.random.default_rng(0)
    defect = f32 + 5e-7 * rng.choice([-1.0, 1.0], size=f32.size)
    assert E.theta_envelope(defect, f32, f64, noise_directions=True)[0] > 1.0
    ratio, _ = E.theta_envelope(f32, f32, f64, noise_directions=True)
    assert ratio < 0.5
