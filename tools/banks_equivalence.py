"""Full-scale equivalence of the bank schedules of the headline filter (N = 65 536, T = 3000, m = 512, block-lower storage, lazy_depth 4, the same
Philox streams): two banks with the shared flush against ONE bank with the shared flush in place (r05).  Every resampling index of the run
(196.6 M), the trajectory outputs and the last weights.   python tools/banks_equivalence.py [T]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(pkg, dg, T, inplace):
    Q = bench.q_mag()
    d = dg.bean_6D(T, Q, bench.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = pkg.dense_mag_prior(512, d["LL"], bench.THETA_MAG)
    with pkg.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, Q, R, 65536, 0.01, rng=pkg.PhiloxRNG(1), keep_history=True, trace=True,
                           lazy_depth=4, inplace=inplace, storage="fp64sym") as s:
        sched = s.schedule()
        s.sync()
        t0 = time.perf_counter()                                     # the steps only (allocation and first touch of the banks excluded)
        s.advance(T)
        s.sync()
        secs = time.perf_counter() - t0
        out = s.finish(want=("traj_max", "traj_mean", "xl_max", "xl_mean", "P_max", "traj_sample_iwmax", "trace_ai"))
    return out, secs, sched


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    a, ta, sa = run(pkg, dg, T, -1)
    b, tb, sb = run(pkg, dg, T, 1)
    rel = lambda x, y: float(np.max(np.abs(x - y)) / max(np.max(np.abs(y)), 1e-300))  # noqa: E731
    rec = {"T": T, "N": 65536, "two_banks": {"schedule": sa, "seconds": round(ta, 2)}, "one_bank": {"schedule": sb, "seconds": round(tb, 2)}}
    if "trace_ai" in a:
        rec["ancestor_indices_compared"] = int(a["trace_ai"].size)
        rec["ancestor_indices_equal"] = bool(np.array_equal(a["trace_ai"], b["trace_ai"]))
    for k in ("traj_max", "traj_mean", "xl_max", "xl_mean", "P_max", "traj_sample_iwmax"):
        rec["rel_" + k] = rel(a[k], b[k])
    print(json.dumps(rec))
