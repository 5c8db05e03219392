"""TEST INFRASTRUCTURE: regenerates a subset of tests/golden/*.npz from the unmodified reference in a SCRAMBLED order (a main() trace first -- the reference sets 4 torch threads there -- then the multi-agent generators in reverse) into a temp dir and compares with the committed files: every generator is self-contained (oracle/make_golden.py::_self_contained).  Needs /root/reference."""
import os, sys, numpy as np, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import importlib
mg = importlib.import_module("oracle.make_golden")
tmp = tempfile.mkdtemp()
mg.OUT = tmp
env_kw = dict(obs_dim=60, act_dim=8, p_term=0.03, p_cost=0.3, trunc_len=20)
# a trace FIRST (the reference's main() sets 4 threads), then the multi-agent generators in reverse order
mg.golden_trace("cpo", "cpo_trace.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw, cfg_over={"learning_iters": 2, "batch_size": 64}, fvp_calls=3, args_over={"cost_limit": 3.0})
mg.golden_ma_runner_trace("macpo", "ma_runner_trace_macpo.npz", N=4, T=8, EP=2)
mg.golden_ma_macpo()
mg.golden_ma_happo_mappo()
mg.golden_ma_mappolag()
mg.golden_ma_runner_trace()
mg.golden_ma_gae()
mg.golden_model()
mg.golden_gae()
bad = 0
for f in sorted(os.listdir(tmp)):
    a, b = np.load(os.path.join(tmp, f)), np.load(os.path.join("/root/repo/tests/golden", f))
    assert sorted(a.files) == sorted(b.files), f
    n = 0
    for k in a.files:
        if k.startswith("e") and "Time_" in k or "Time/" in k:
            continue
        if not np.array_equal(a[k], b[k], equal_nan=True) :
            n += 1
            if n <= 3: print("  differs:", f, k)
    print(f, len(a.files), "arrays,", n, "differ")
    bad += n
print("TOTAL differing arrays:", bad)
