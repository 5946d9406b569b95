#!/usr/bin/env python3
"""Wall-clock of the conditional particle smoothers on the GPU box (device Philox, synthetic data from the
product generator).

  smoother_bench.py ref  [N_K]                 the reference's own sizes: dense-mag N_P=100 T=192 m=512
                                               (run_dense3D_magfield.m:85,134; generateData_dense.m:184-187) and
                                               dense-radio N_P=100 T=48 m=128 (run_dense2D_withHeading.m:83,108,165)
  smoother_bench.py mag   N_P T m N_K [forms] [key=value ...]  one dense-mag run   (forms: info, cov or info,cov)
  smoother_bench.py radio N_P T m N_K [forms] [key=value ...]  one dense-radio run (square_3D trajectory scaled to T points)
                                               key=value: options of the information form (lazy_depth=3, chol_refresh=32, ...)

Prints one JSON line per run."""
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
import bench  # noqa: E402  (Q / theta of examples/slam-dense-mag/main.m:22-23)

THETA_RADIO = [0.25, 2.0, 0.01]                                     # examples/slam-dense-radio/main.m:24
FORMS = {"info": rbpf.particleSmootherInformationForm, "cov": rbpf.particleSmoother}


def run(kind, N, T, m, N_K, forms, **opts):
    if kind == "mag":
        Q, dt = bench.q_mag(), 0.01
        d = dg.bean_6D(T, Q, bench.THETA_MAG, dt, seed=1)
        mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], bench.THETA_MAG)
    else:
        dt = 1.0
        Q = dg.radio_Q(T, "square_3D")
        d = dg.planar_heading(T, Q, THETA_RADIO, dt, seed=1, nLL=4, traj="square_3D")
        mdl, x0, P0, R = rbpf.dense_radio_prior(m, d["LL"], THETA_RADIO)
    for name in forms:
        marks = []
        kw = opts if name == "info" else {}
        t0 = time.perf_counter()
        XNK, XLK, PK = FORMS[name](mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, Q, R,
                                   N, N_K, dt, False, lambda *a: marks.append(time.perf_counter()), rng=rbpf.PhiloxRNG(3), **kw)
        secs = time.perf_counter() - t0
        its = [round(b - a, 3) for a, b in zip([t0] + marks[:-1], marks)]
        npos = d["pos"].shape[0]
        rmse = float(np.sqrt(np.mean((XNK[0:npos, :, -1] - d["pos"]) ** 2)))
        print(json.dumps({"smoother": name, "model": "dense-" + kind, "N_P": N, "T": T, "m": m, "nLin": mdl.nLin, "N_K": N_K,
                          "options": kw, "seconds": round(secs, 3), "seconds_per_iteration": its,
                          "ms_per_step_last_iteration": round(its[-1] / T * 1e3, 3),
                          "particle_steps_per_s": round(N * T * N_K / secs, 1),
                          "pos_rmse_last_iteration": round(rmse, 4), "finite": bool(np.all(np.isfinite(XNK)))}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "ref"
    if which == "ref":
        N_K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
        run("mag", 100, 192, 512, N_K, ("info", "cov"))
        run("radio", 100, 48, 128, N_K, ("info", "cov"))
    else:
        N, T, m, N_K = (int(v) for v in sys.argv[2:6])
        rest = sys.argv[6:]
        forms = rest[0].split(",") if rest and "=" not in rest[0] else ["info"]
        opts = {a.split("=")[0]: int(a.split("=")[1]) for a in rest if "=" in a}
        run(which, N, T, m, N_K, forms, **opts)
