#!/usr/bin/env python3
"""Diagnostic (r05): dense-radio siblings tie structurally (measModel sees the position only, which propagates without noise:
run_dense2D_withHeading.m:75-76,168).  Prints the device's and the oracle's log-weights bit for bit for the case
profiles/r04_size_scan.txt recorded as FAIL (radio m = 512, N_P = 6, T = 7, seed 41)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import cases  # noqa: E402

rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    kw = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        kw[k] = int(v) if v.lstrip("-").isdigit() else v
    c = cases.radio_case(6, 7, m, seed=41, N_K=2)
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                              rng=cases.device_rng(rbpf, c), extras=True, **kw)
    ex = out[8]
    tr = ref["trace"]
    print("options", kw, "iw_max device", ex["iw_max"], "oracle", ref["iw_max"])
    for t in range(7):
        print("t", t, "ai", None if t == 0 else list(ex["ai"][t]), "equal", t == 0 or bool(np.all(ex["ai"][t] == tr["ai"][t])))
        print("   dev", [float(x).hex() for x in ex["logw"][t]])
        print("   ora", [float(x).hex() for x in tr["logw"][t]])
    print("xn dev", [[float(v).hex() for v in r] for r in ex["xn"]])
    print("xn ora", [[float(v).hex() for v in r] for r in tr["xn"][-1] if False] or [[float(v).hex() for v in r] for r in tr["xn"]])


if __name__ == "__main__":
    main()
