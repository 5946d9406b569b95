"""Block-lower storage at sixteen tile rows (nLin = 1027) and fp32 tiles: a quick parity check against the numpy oracle at N = 6 and
the step time at BASELINE.json configs[4]'s per-GPU share (N = 32 768, m = 1024) for every storage.  Test infrastructure / tuning aid
(the oracle is only the checker here); run on the GPU box:  python tools/sym16_probe.py [parity] [time] [N=32768]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import bench  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def parity(pkg):
    import cases
    for m in (512, 1024):
        c = cases.mag_case(6, 9, m, seed=29)
        ref = cases.oracle_filter(c)
        mdl, x0, P0, R = cases.device_model(pkg, c)
        for storage, lazy, inplace in (("fp64sym", 0, -1), ("fp64sym", 2, -1), ("fp64sym", 3, 1), ("fp64sym", 4, -1), ("fp64sym", 4, 1),
                                       ("fp32sym", 0, -1), ("fp32sym", 2, -1), ("fp32sym", 3, 1), ("fp32sym", 4, -1), ("fp32", 2, -1)):
            if m == 512 and storage == "fp64sym":
                continue
            out = pkg.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                                     rng=cases.device_rng(pkg, c), extras=True, lazy_depth=lazy, inplace=inplace, storage=storage)
            ex, tr = out[8], ref["trace"]
            print(f"m={m} {storage} lazy {lazy} inplace {inplace}: ai equal {np.array_equal(ex['ai'][1:], tr['ai'][1:])}  w {rel(ex['w'], tr['w']):.2e}  "
                  f"xl {rel(ex['xl'], tr['xl']):.2e}  P {rel(ex['P'], tr['P']):.2e}  P_max {rel(out[4], ref['P_max']):.2e}  traj_mean {rel(out[1], ref['traj_mean']):.2e}",
                  flush=True)


def timing(pkg, N):
    dg = importlib.import_module(pkg.__name__ + ".datagen")
    legs = (("fp32", 2, 0), ("fp32sym", 2, 0), ("fp32sym", 3, 0), ("fp32sym", 4, 0), ("fp32sym", 4, 1), ("fp64sym", 4, 0), ("fp64", 2, 0))
    if "deep" in sys.argv[1:]:
        legs = (("fp32sym", 4, 0), ("fp32sym", 5, 0), ("fp32sym", 6, 0), ("fp32sym", 8, 0), ("fp64sym", 6, 0), ("fp64sym", 8, 0))
    for storage, lazy, inplace in legs:
        try:
            r, *_ = bench.filter_leg(pkg, dg, N, 1024, 3000, 24, 4, 1, lazy, inplace, storage)
            print(json.dumps({"storage": storage, "lazy_depth": lazy, "inplace": inplace, "N": N, "Mps": r["value"] / 1e6, "ms_per_step": r["ms_per_step"],
                              "kernel_ms": r["roofline"]["avg_launch_ms"], "scheduled_GBps": r["roofline"]["achieved"]}), flush=True)
        except Exception as exc:
            print(json.dumps({"storage": storage, "lazy_depth": lazy, "error": repr(exc)[:300]}), flush=True)


if __name__ == "__main__":
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    N = 32768
    for a in sys.argv[1:]:
        if a.startswith("N="):
            N = int(a[2:])
    if "parity" in sys.argv[1:] or len(sys.argv) == 1:
        parity(pkg)
    if "time" in sys.argv[1:] or "deep" in sys.argv[1:] or len(sys.argv) == 1:
        timing(pkg, N)
