#!/usr/bin/env python3
"""The metric's smoother run (slam-dense-mag m = 512, T = 3000, N_K = 2, block-lower P, lazy_depth 3, device Philox) at the per-GPU
share N_P twice on the same random streams: the library default (ancestor-weight factors carried, refactorised every 32nd step) and
the reference's arithmetic (chol_refresh = 1: chol(Imat_i + ImatAddt) for every particle at every step,
particleSmootherInformationForm.m:224-236).  Reports whether every ancestor index and both trajectory draws are identical and how
far the ancestor probabilities, weights and outputs are apart.  python tools/carried_vs_fresh.py [N_P] [T] [--refresh-free]   (one JSON line; --refresh-free: chol_refresh >= N_T with one covariance
bank in place -- the configuration of the N_P = 65 536 single-GPU run -- instead of the default)"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")


def main():
    pos = [v for v in sys.argv[1:] if not v.startswith("--")]
    N = int(pos[0]) if len(pos) > 0 else 8192
    T = int(pos[1]) if len(pos) > 1 else 3000
    Q = bench.q_mag()
    d = dg.bean_6D(T, Q, bench.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(512, d["LL"], bench.THETA_MAG)
    res, secs = {}, {}
    first = ("refresh_free_in_place", dict(chol_refresh=10 ** 6, inplace=1)) if "--refresh-free" in sys.argv else ("default", dict(chol_refresh=0))
    for v in sys.argv[1:]:
        if v.startswith("--origin="):                    # refresh every K-th step from the origin, no information matrix stored, one bank in place
            first = (f"origin_rebuild_K{int(v[9:])}_in_place", dict(chol_refresh=int(v[9:]), info_rebuild=1, inplace=1))
    # --baseline-inplace: the from-scratch run keeps one covariance bank in place as well, so that both runs round P alike and the
    # difference is the ancestor-weight arithmetic alone
    base_kw = dict(chol_refresh=1, inplace=1) if "--baseline-inplace" in sys.argv else dict(chol_refresh=1)
    for tag, kw in (first, ("from_scratch", base_kw)):
        t0 = time.perf_counter()
        XNK, XLK, PK, ex = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0,
                                                                Q, R, N, 2, 0.01, rng=rbpf.PhiloxRNG(3), extras=True, lazy_depth=3, storage="fp64sym",
                                                                **kw)
        secs[tag] = round(time.perf_counter() - t0, 2)
        res[tag] = (XNK, XLK, PK, ex["ai"], ex["ak"], ex["paNt"][1, 1:], ex["w"])
    a, b = res[first[0]], res["from_scratch"]
    rel = lambda x, y: float(np.max(np.abs(x - y)) / max(np.max(np.abs(y)), 1e-300))      # noqa: E731
    dp = np.max(np.abs(a[5] - b[5]), axis=1)
    print(json.dumps({"compared": first[0] + " vs from_scratch" + (" (in place as well)" if "--baseline-inplace" in sys.argv else ""), "N_P": N, "T": T, "m": 512, "N_K": 2, "chol_refresh_in_use_default": rbpf.chol_refresh_in_use(mdl, 0), "seconds_with_traces": secs,
                      "ancestor_indices_identical": bool(np.array_equal(a[3][:, 1:], b[3][:, 1:])), "ancestor_indices_compared": int(a[3][:, 1:].size),
                      "trajectory_draws_identical": bool(np.array_equal(a[4], b[4])), "max_abs_diff_paNt": float(dp.max()), "at_step": int(dp.argmax()) + 1,
                      "rel_diff_weights": rel(a[6], b[6]), "rel_diff_XNK": rel(a[0], b[0]), "rel_diff_XLK": rel(a[1], b[1]), "rel_diff_PK": rel(a[2], b[2]),
                      "finite": bool(np.all(np.isfinite(a[0])) and np.all(np.isfinite(a[5])))}))


if __name__ == "__main__":
    main()
