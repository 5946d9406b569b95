#!/usr/bin/env python3
"""examples/slam-dense-mag on the device path, one run of run_dense3D_magfield.m's flow: synthetic 'bean_6D' data (the product's
generator), the reduced-rank GP prior, then dead reckoning, the EKF baseline (ekf_dense.m), particleFilter (N_P = 100) and
particleSmootherInformationForm (N_K iterations), each scored as the runner scores them (run_dense3D_magfield.m:155-183,216-237):
per-axis RMS position error after a Procrustes alignment and RMS orientation error in degrees.

    python tools/dense_mag_demo.py [N_T=500] [m=512] [N_P=100] [N_K=10] [seed=1] [lazy_depth=0] [chol_refresh=0]

Prints one JSON line per estimator.  (One run says little about which estimator wins: the reference's own comparison, main.m:
37-57, averages 20 simulations per disturbance level.)"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
mt = importlib.import_module("rao-blackwellized-slam-smoothing_amd.metrics")
ekf = importlib.import_module("rao-blackwellized-slam-smoothing_amd.ekf")
import bench  # noqa: E402   (Q and theta of examples/slam-dense-mag/main.m:22-23)


def dead_reckoning(d):
    """The odometry integrated on its own (what the estimators start from): position sums, quaternion products."""
    T = d["y"].shape[0]
    x = np.zeros((7, T))
    x[:, 0] = d["initState"]
    for t in range(1, T):
        x[0:3, t] = x[0:3, t - 1] + d["dx"][t - 1, 0:3]
        q, p = x[3:7, t - 1], d["dx"][t - 1, 3:7]
        x[3:7, t] = [q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3], q[1] * p[0] + q[0] * p[1] - q[3] * p[2] + q[2] * p[3],
                     q[2] * p[0] + q[3] * p[1] + q[0] * p[2] - q[1] * p[3], q[3] * p[0] - q[2] * p[1] + q[1] * p[2] + q[0] * p[3]]
    return x


def run(N_T=500, m=512, N_P=100, N_K=10, seed=1, lazy_depth=0, chol_refresh=0, with_ekf=True):
    Q, theta, dt = bench.q_mag(), bench.THETA_MAG, 0.01
    d = dg.bean_6D(N_T, Q, theta, dt, seed=seed)
    mdl, x0_lin, P0_lin, R = rbpf.dense_mag_prior(m, d["LL"], theta)
    res = []

    def score(name, traj, secs, **extra):
        rp, ro = mt.rmse_dense_mag(d["pos"], d["quat"], traj)
        res.append(dict(estimator=name, seconds=round(secs, 3), rmse_pos=[round(float(v), 4) for v in rp],
                        rmse_pos_total=round(float(np.sqrt(np.mean(rp ** 2))), 4), rmse_ori_deg=[round(float(v), 3) for v in ro], **extra))

    score("dead reckoning", dead_reckoning(d), 0.0)
    if with_ekf:
        n = mdl.nLin
        x0 = np.concatenate((d["initState"][0:3], np.zeros(3), np.asarray(x0_lin).ravel()))      # run_dense3D_magfield.m:248-250
        P0 = np.zeros((6 + n, 6 + n))
        P0[6:, 6:] = P0_lin
        t0 = time.perf_counter()
        xf, qnb, _ = ekf.ekf_dense(mdl, d["LL"], d["dx"], d["y"], x0, d["initState"][3:7], P0, Q, R, dt)
        score("EKF", np.vstack((xf[0:3], qnb)), time.perf_counter() - t0)
    t0 = time.perf_counter()
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0_lin, P0_lin, Q, R, N_P, dt,
                              rng=rbpf.PhiloxRNG(seed), lazy_depth=lazy_depth)
    secs = time.perf_counter() - t0
    score("particleFilter, highest weight", out[0], secs, N_P=N_P)
    score("particleFilter, weighted mean", out[1], secs, N_P=N_P)
    marks = []
    t0 = time.perf_counter()
    XNK, XLK, PK = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],
                                                        x0_lin, P0_lin, Q, R, N_P, N_K, dt, False, lambda *a: marks.append(time.perf_counter()),
                                                        rng=rbpf.PhiloxRNG(seed + 1), lazy_depth=min(lazy_depth, 3), chol_refresh=chol_refresh)
    secs = time.perf_counter() - t0
    for k in range(N_K):
        score(f"particleSmootherInformationForm, iteration {k + 1}", XNK[:, :, k], (marks[k] - t0) if k < len(marks) else secs, N_P=N_P)
    burn = max(1, N_K // 2)                                                   # the trajectories after burn-in, averaged
    score("particleSmootherInformationForm, mean of the later iterations", XNK[:, :, burn:].mean(axis=2), secs, N_P=N_P, N_K=N_K)
    return dict(N_T=N_T, m=m, nLin=int(mdl.nLin), results=res)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    r = run(**{k: int(v) for k, v in kw.items()})
    for row in r["results"]:
        print(json.dumps(row), flush=True)
