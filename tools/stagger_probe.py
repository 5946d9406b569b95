#!/usr/bin/env python3
"""TIMING PROBE (diagnostic build librbpf_hip_tuning.so; results of the probed step are wrong by design): what would one step of a
STAGGERED flush cost -- a quarter of the families flushing (writers + their read-only siblings) while the other three quarters run
the read-only variant -- with the three launches one after the other (RBPF_STAGGER_PROBE=1) and with the read-only particles on a
second stream beside the flush (=2)?  Baseline: the lock-step flush step and the read-only steps of the product schedule.
Headline configuration (N = 65 536, m = 512, block-lower storage, lazy_depth 4, two banks).  One JSON line per setting."""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(first, count):
    sys.path.insert(0, ROOT)
    import bench
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    Q = bench.q_mag()
    d = dg.bean_6D(3000, Q, bench.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = pkg.dense_mag_prior(512, d["LL"], bench.THETA_MAG)
    with pkg.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, Q, R, 65536, 0.01, rng=pkg.PhiloxRNG(1), keep_history=True,
                           lazy_depth=4, inplace=0, storage="fp64sym") as s:
        s.advance(first)
        s.sync()
        s.timing(enable=True)
        s.advance(count)
        s.sync()
        tm = s.timing(reset=True)
    print(json.dumps({"first_step": first, "steps": count, "ms": tm["ms"], "launches": tm["launches"]}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    lib = os.path.join(ROOT, "rao-blackwellized-slam-smoothing_amd", "lib", "librbpf_hip_tuning.so")
    for tag, env_extra, first, count in (("flush step t=16, product schedule", {}, 16, 1), ("read-only steps t=13..15", {}, 13, 3),
                                         ("probe, serial launches", {"RBPF_STAGGER_PROBE": "1"}, 16, 1),
                                         ("probe, read-only particles beside the flush", {"RBPF_STAGGER_PROBE": "2"}, 16, 1),
                                         ("probe, the read-only three quarters alone", {"RBPF_STAGGER_PROBE": "3"}, 16, 1),
                                         ("probe, the flushing quarter alone", {"RBPF_STAGGER_PROBE": "4"}, 16, 1),
                                         ("probe, serial launches (repeat)", {"RBPF_STAGGER_PROBE": "1"}, 16, 1),
                                         ("probe, beside (repeat)", {"RBPF_STAGGER_PROBE": "2"}, 16, 1)):
        env = dict(os.environ, RBPF_LIB_PATH=lib, **env_extra)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(first), str(count)], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(tag, "|", line[-1] if line else "FAILED " + r.stderr[-300:], flush=True)
