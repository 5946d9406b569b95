#!/usr/bin/env python3
"""Diagnostic: filter / information-form / covariance-form parity against the numpy oracle over a scan of basis sizes and options
(r04 found a silent size class -- three row chunks -- that no test had touched; this scan looks for more).  Prints one line per case:
ok / FAIL / the library's own refusal.  A filter case that fails against the numpy oracle is judged once more against the
EXTENDED-PRECISION build of the C restatement (the arbiter): "ok (arbiter)" means the disagreement was a near-tie of two weights that
the numpy oracle's own rounding decided (profiles/r04_size_scan.txt's FAIL at dense-radio m = 512: tests/test_gpu_r05_parity.py).
   python tools/size_scan.py [mag|radio] [m]      (restrict the scan)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import cases  # noqa: E402
import test_gpu_smoother as ts  # noqa: E402
from test_gpu_filter import check_filter  # noqa: E402

rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")


def one(tag, fn):
    try:
        fn()
        return "ok"
    except NearTie:
        return "ok (arbiter)"
    except AssertionError:
        return "FAIL"
    except rbpf.RBPFError as e:
        return "refused(" + str(e)[:40] + ")"
    except Exception as e:                                           # noqa: BLE001
        return "ERR " + repr(e)[:60]


class NearTie(Exception):
    pass


def filt(c, **kw):
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                              rng=cases.device_rng(rbpf, c), extras=True, **kw)
    try:
        check_filter(ref, out)
    except AssertionError:
        import oracle_c
        arb, _ = oracle_c.particle_filter(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                                          cases.device_rng(rbpf, c), lib_path=oracle_c.build_arbiter())
        ex = out[8]
        rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))      # noqa: E731
        assert np.array_equal(ex["ai"][1:], arb["trace_ai"].T[1:]) and int(ex["iw_max"]) == int(arb["iw_max"][0])
        assert np.array_equal(np.argmax(ex["w"], axis=1), np.argmax(arb["trace_w"], axis=0))
        assert rel(ex["w"], arb["trace_w"].T) <= 1e-9 and rel(out[0], arb["traj_max"]) <= 1e-9 and rel(out[1], arb["traj_mean"]) <= 1e-9
        assert rel(out[2], arb["xl_max"]) <= 1e-9 and rel(out[4], arb["P_max"]) <= 1e-9
        raise NearTie()


def smooth(c, info, **kw):
    ref, out = ts.run_both(rbpf, c, info_form=info, **kw)
    ts.check(ref, out, c["N_K"])


def main():
    bad = 0
    cfgs = []
    for m in (60, 100, 125, 128, 129, 200, 253, 256, 300, 380, 381, 400, 509, 510, 512, 560, 600, 636, 637, 700, 1021):
        cfgs.append(("mag", m, {}))
    for m in (253, 256, 300, 380, 509, 510, 512, 560, 600, 636):
        for lz in (0, 3, 4):
            cfgs.append(("mag", m, dict(storage="fp64sym", lazy_depth=lz)))
    for m in (130, 200, 256, 300, 400, 512, 600, 700):
        for lz in (2, 3):
            cfgs.append(("mag", m, dict(lazy_depth=lz)))
    for m in (24, 100, 128, 200, 256, 300, 384, 400, 500, 512, 600):
        cfgs.append(("radio", m, {}))
        cfgs.append(("radio", m, dict(lazy_depth=3)))
    only = sys.argv[1:]
    if only:
        cfgs = [q for q in cfgs if q[0] == only[0] and (len(only) < 2 or q[1] == int(only[1]))]
    for kind, m, kw in cfgs:
        c = (cases.mag_case if kind == "mag" else cases.radio_case)(6, 7, m, seed=41, N_K=2)
        r = [one("f", lambda: filt(c, **kw)), one("i", lambda: smooth(c, True, **{k: v for k, v in kw.items() if not (k == "lazy_depth" and v > 3)})),
             one("c", lambda: smooth(c, False, **{k: v for k, v in kw.items() if k == "storage"}))]
        bad += sum(x == "FAIL" or x.startswith("ERR") for x in r)
        print(kind, m, kw, "filter", r[0], "| info", r[1], "| cov", r[2], flush=True)
    print("cases with wrong results:", bad)


if __name__ == "__main__":
    main()
