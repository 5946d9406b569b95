#!/usr/bin/env python3
"""Diagnostic (GPU box): offspring statistics of the bench workload -- unique ancestors per step, largest family,
effective sample size -- from the per-step ancestor trace."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
import bench
N, T, m = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 200, 256
Q = bench.q_mag()
d = dg.bean_6D(3000, Q, bench.THETA_MAG, 0.01, seed=1)
mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], bench.THETA_MAG)
out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"][:T - 1], d["y"][:T], d["initState"], x0, P0, Q, R, N, 0.01,
                          rng=rbpf.PhiloxRNG(1), extras=True, want_xn_traj=False)
ex = out[8]
ai, w = ex["ai"], ex["w"]
for t in list(range(1, 6)) + list(range(10, T, max(1, T // 12))):
    u, c = np.unique(ai[t], return_counts=True)
    ess = 1.0 / np.sum(w[t - 1] ** 2)
    print(f"t={t:4d} unique={u.size:5d} ({u.size / N:.2f})  max_family={c.max():5d}  ESS(t-1)={ess:8.1f}")
