#!/usr/bin/env python3
"""Times the carried-factor sweep kernel (rbpf_chol_sweep.hpp) for every library variant lib/librbpf_hip*.so given on the command line
(default: the product library): stand-alone (rbpf_chol_sweep_probe: 8192 distinct factors, nLin = 515) and inside the smoother
(tools/smoother_bench.py mag 8192 100 512 2: ms per time step of the second iteration).  One line per library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, sys
sys.path.insert(0, %r)
import bench
pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
r = bench.smoother_sweep_roofline(pkg)
print(json.dumps({"standalone_ms": r["avg_launch_ms"], "standalone_GBps": r["achieved"]}))
''' % ROOT
libs = sys.argv[1:] or [""]
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["RBPF_LIB_PATH"] = os.path.join(ROOT, "rao-blackwellized-slam-smoothing_amd", "lib", lib)
    a = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    la = [l for l in a.stdout.splitlines() if l.startswith("{")]
    b = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "smoother_bench.py"), "mag", "8192", "100", "512", "2", "info", "lazy_depth=3", "storage=2"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lb = [l for l in b.stdout.splitlines() if l.startswith("{")]
    sa = json.loads(la[-1]) if la else {"error": a.stderr[-200:]}
    sb = json.loads(lb[-1]) if lb else {"error": b.stderr[-200:]}
    print(lib or "product", "|", sa, "| smoother ms/step (iteration 2):", sb.get("ms_per_step_last_iteration"), sb.get("seconds_per_iteration"), flush=True)
