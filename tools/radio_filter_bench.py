#!/usr/bin/env python3
"""Forward-filter throughput on synthetic slam-dense-radio data (run_dense2D_withHeading.m closures: planar position +
heading, scalar field, nLin = m, ny = 1), device Philox.  Usage: radio_filter_bench.py N_P m T steps [lazy_depth]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")

N, m, T, K = (int(v) for v in sys.argv[1:5])
lazy = int(sys.argv[5]) if len(sys.argv) > 5 else 3
th = [0.25, 2.0, 0.01]                                               # examples/slam-dense-radio/main.m:24
Q = dg.radio_Q(T, "square_3D")
d = dg.planar_heading(T, Q, th, 1.0, seed=1, nLL=4, traj="square_3D")
mdl, x0, P0, R = rbpf.dense_radio_prior(m, d["LL"], th)
W = 8
with rbpf.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, Q, R, N, 1.0, rng=rbpf.PhiloxRNG(1), keep_history=False,
                        lazy_depth=lazy) as s:
    s.advance(W)
    s.sync()
    s.timing(enable=True)
    t0 = time.perf_counter()
    s.advance(K)
    s.sync()
    secs = time.perf_counter() - t0
    tm = s.timing(reset=True)
avg_ms = tm["ms"] / max(tm["launches"], 1)
print(json.dumps({"model": "dense-radio", "N_P": N, "m": m, "nLin": mdl.nLin, "T": T, "steps": K, "lazy_depth": lazy,
                  "particle_steps_per_s": N * K / secs, "ms_per_step": secs / K * 1e3, "step_kernel_ms": avg_ms,
                  "algorithmic_GBps": tm["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9}), flush=True)
