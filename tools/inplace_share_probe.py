"""Single covariance bank with the shared flush (r05: launch_share_inplace_plan) against the per-child in-place flush (RBPF_SHARE_INPLACE=0:
read by a -DRBPF_TUNING build only -- point RBPF_LIB_PATH at one, tools/ab_env.py says how)
and against two banks: the headline filter configuration and the per-step time of the information-form smoother at N_P = 65 536.
  python tools/inplace_share_probe.py [filter] [smoother]"""
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(what):
    import bench
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    if what.startswith("filter"):
        inplace = int(what.split(":")[1])
        r, *_ = bench.filter_leg(pkg, dg, 65536, 512, 3000, 60, 12, 1, 4, inplace, "fp64sym")
        print(json.dumps({"leg": what, "Mps": r["value"] / 1e6, "kernel_ms": r["roofline"]["avg_launch_ms"]}), flush=True)
    else:
        T = int(what.split(":")[1])
        r = bench.smoother_share_full(pkg, dg, 65536, T, 512, 2, 1, lazy_depth=3, storage="fp64sym", chol_refresh=T, inplace=1)
        print(json.dumps({"leg": what, "seconds": r["seconds"], "per_iteration": r["seconds_per_iteration"], "ms_per_step": r["ms_per_time_step_with_ancestor_sampling"]}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    legs = []
    if "filter" in sys.argv[1:] or len(sys.argv) == 1:
        legs += [("filter:-1", {}), ("filter:1", {}), ("filter:1", {"RBPF_SHARE_INPLACE": "0"})]
    if "smoother" in sys.argv[1:] or len(sys.argv) == 1:
        legs += [("smoother:200", {}), ("smoother:200", {"RBPF_SHARE_INPLACE": "0"})]
    for what, env in legs:
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", what], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(env or "default", line[-1] if line else ("FAILED " + r.stderr[-400:]), f"({time.time() - t0:.0f} s)", flush=True)
