#!/usr/bin/env python3
"""examples/slam-sparse-visual/main.m on the device path: particle filter (N_P = 100) and particle smoother (N_P = 10,
N_K = 10) on the reference's data file, with the Procrustes-aligned RMSEs the script prints (main.m:48-51,62-65).
Usage: sparse_visual_demo.py <curve-x2.mat> [seed]"""
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

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "curve-x2.mat")
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 42
for name, N_P, N_K in (("particleFilter", 100, 0), ("particleSmoother", 10, 10)):
    d = dg.sparse_visual_load(path, seed=seed, N_P=N_P)
    mdl = rbpf.SparseVisualModel(d["nLand"], *d["cam"])
    t0 = time.perf_counter()
    if N_K == 0:
        out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["odometry"], d["y"], d["x0_nonLin"], d["x0_lin"], d["P0_lin"],
                                  d["Q"], d["R"], N_P, d["dt"], True, rng=rbpf.PhiloxRNG(seed))
        traj, xl = out[1], out[3]                                              # traj_mean, xl_mean  (pfslam.m:112-114)
    else:
        XNK, XLK, PK = rbpf.particleSmoother(mdl.dynModel, mdl.measModel, [], d["odometry"], d["y"], d["x0_nonLin"], d["x0_lin"],
                                             d["P0_lin"], d["Q"], d["R"], N_P, N_K, d["dt"], True, rng=rbpf.PhiloxRNG(seed))
        traj, xl = XNK[:, :, 1:].mean(axis=2), XLK[:, 1:].mean(axis=1)          # psslam.m:122-123
    secs = time.perf_counter() - t0
    rp, rm = mt.calc_rmses(d["map"].T, xl.reshape(-1, 2), d["groundTruth"].T, traj.T)
    print(json.dumps({"estimator": name, "N_P": N_P, "N_K": N_K, "T": int(d["y"].shape[0]), "seconds": round(secs, 3),
                      "rmse_path": round(rp, 4), "rmse_map": round(rm, 4)}), flush=True)
