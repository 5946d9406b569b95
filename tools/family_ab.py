#!/usr/bin/env python3
"""A / B of the filter's read-only steps at the headline size (N = 65 536, m = 512, block-lower storage): family products
(rbpf_options.family_products = 1) against the per-particle stream (0, the default), over lazy depths.
   python tools/family_ab.py [--lazy 4 6 8] [--steps 48] [--warmup 9] [--N 65536]
One JSON line per (lazy_depth, family_products): particle-steps/s over the timed steps, ms per step."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=65536)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--lazy", type=int, nargs="+", default=[4, 6, 8])
    ap.add_argument("--family", type=int, nargs="+", default=[1, 0])
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=9)
    ap.add_argument("--T", type=int, default=3000)
    args = ap.parse_args()
    import bench
    import numpy as np
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    datagen = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    Q = bench.q_mag()
    data = datagen.bean_6D(args.T, Q, bench.THETA_MAG, 0.01, seed=1)
    model, x0_lin, P0, R = pkg.dense_mag_prior(args.m, data["LL"], bench.THETA_MAG)
    for C in args.lazy:
        ref = None
        for fam in args.family:
            with pkg.FilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R, args.N, 0.01, rng=pkg.PhiloxRNG(1),
                                   keep_history=True, lazy_depth=C, inplace=0, storage="fp64sym", family_products=fam) as sess:
                sess.advance(args.warmup)
                sess.sync()
                t0 = time.perf_counter()
                sess.advance(args.steps)
                sess.sync()
                dt = time.perf_counter() - t0
                chk = sess.finish(want=("traj_mean",))
            tm = chk["traj_mean"][:, :args.warmup + args.steps]
            line = {"lazy_depth": C, "family_products": fam, "particle_steps_per_s": args.N * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                    "finite": bool(np.all(np.isfinite(tm)))}
            if ref is None:
                ref = tm
            else:
                line["traj_mean_rel_to_first"] = float(np.max(np.abs(tm - ref)) / np.max(np.abs(ref)))
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
