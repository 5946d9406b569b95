#!/usr/bin/env python3
"""A/B of tuning knobs on the GPU box: runs bench.py (filter leg only) with the -DRBPF_TUNING build of the library
(rao-blackwellized-slam-smoothing_amd/lib/librbpf_hip_tuning.so: `_build.build(defines={"RBPF_TUNING": 1}, out=...)`) once per environment setting.
Usage: python tools/ab_env.py [--steps K] NAME=VALUE[,NAME=VALUE] ...   (an empty string "" = no override)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "rao-blackwellized-slam-smoothing_amd", "lib", "librbpf_hip_tuning.so")
steps = "100"
args = sys.argv[1:]
if "--steps" in args:
    i = args.index("--steps")
    steps = args[i + 1]
    del args[i:i + 2]
for spec in args or [""]:
    env = dict(os.environ, RBPF_LIB_PATH=lib)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "12", "--no-cpu-baseline", "--no-smoother",
                        "--no-large", "--no-traffic", "--no-filter-full"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"{spec or 'default':40s} FAILED: {r.stderr[-300:]}", flush=True)
        continue
    j = json.loads(line[-1])
    print(f"{spec or 'default':40s} value={j['value'] / 1e6:7.3f} M/s  ms/step={j['ms_per_step']:.4f}  kernel_ms={j['roofline']['avg_launch_ms']:.4f}", flush=True)
