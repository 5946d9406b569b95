"""dense-radio on block-lower storage (n_y = 1, nLin = 128, two tile rows): parity against the numpy oracle at N = 8 (filter, both
smoothers) and the step time at N = 65 536 next to the full square.  Tuning aid / test infrastructure."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


if __name__ == "__main__":
    import cases
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    c = cases.radio_case(8, 11, 128, seed=5)
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(pkg, c)
    for lazy, inplace in ((0, -1), (2, -1), (3, -1), (3, 1), (4, -1), (4, 1)):
        out = pkg.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                                 rng=cases.device_rng(pkg, c), extras=True, lazy_depth=lazy, inplace=inplace, storage="fp64sym")
        ex, tr = out[8], ref["trace"]
        print(f"filter lazy {lazy} inplace {inplace}: ai {np.array_equal(ex['ai'][1:], tr['ai'][1:])} w {rel(ex['w'], tr['w']):.1e} xl {rel(ex['xl'], tr['xl']):.1e} "
              f"P {rel(ex['P'], tr['P']):.1e} P_max {rel(out[4], ref['P_max']):.1e}", flush=True)
    import test_gpu_smoother as ts
    c = cases.radio_case(8, 9, 128, seed=7, N_K=3)
    for info in (False, True):
        for kw in (dict(), dict(lazy_depth=3), dict(lazy_depth=3, chol_refresh=1)):
            if not info and kw:
                continue
            r, o = ts.run_both(pkg, c, info_form=info, storage="fp64sym", **kw)
            try:
                ts.check(r, o, 3)
                print("smoother info" if info else "smoother cov", kw, "ok", flush=True)
            except AssertionError as e:
                print("smoother info" if info else "smoother cov", kw, "FAIL", str(e)[:200], flush=True)
    T, N = 48, 65536
    Qr = dg.radio_Q(T, "square_3D")
    th = [0.25, 2.0, 0.01]
    d = dg.planar_heading(T, Qr, th, 1.0, seed=1, nLL=4, traj="square_3D")
    mdl, x0, P0, R = pkg.dense_radio_prior(128, d["LL"], th)
    for storage in ("fp64", "fp64sym"):
        for lazy in (0, 3, 4):
            with pkg.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, Qr, R, N, 1.0, rng=pkg.PhiloxRNG(3), keep_history=True,
                                   lazy_depth=lazy, storage=storage) as s:
                s.advance(8); s.sync()
                s.timing(enable=True)
                t0 = time.perf_counter()
                s.advance(36); s.sync()
                dt = time.perf_counter() - t0
                tm = s.timing(reset=True)
            print(json.dumps({"storage": storage, "lazy_depth": lazy, "ms_per_step": dt / 36 * 1e3, "step_kernel_ms": tm["ms"] / max(tm["launches"], 1),
                              "Mps": N * 36 / dt / 1e6}), flush=True)
    for storage in ("fp64", "fp64sym"):
        ts_ = []
        for _ in range(2):
            t0 = time.perf_counter()
            pkg.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, Qr, R, N, 3, 1.0,
                                                rng=pkg.PhiloxRNG(3), lazy_depth=3, chol_refresh=16, storage=storage)
            ts_.append(round(time.perf_counter() - t0, 3))
        print("smoother N=65536", storage, ts_, flush=True)
