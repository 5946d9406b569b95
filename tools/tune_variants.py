#!/usr/bin/env python3
"""Runs bench.py once per (library variant, layout override) and prints one compact line each.
Tuning aid for the GPU box:  python3 tools/tune_variants.py [--steps K] [-- extra bench.py flags]
The RBPF_RS / RBPF_CS layout overrides are only read by libraries built with -DRBPF_TUNING
(`_build.build(defines={"RBPF_TUNING": 1}, out=".../lib/librbpf_hip_tuning.so")`); the product library ignores the
environment."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "rao-blackwellized-slam-smoothing_amd", "lib")
steps = "300"
if "--steps" in sys.argv:
    steps = sys.argv[sys.argv.index("--steps") + 1]
extra = []
if "--" in sys.argv:
    extra = sys.argv[sys.argv.index("--") + 1:]
variants = [("default", None, {})]
for f in sorted(os.listdir(LIBDIR)):
    if f.startswith("librbpf_hip_") and f.endswith(".so"):
        variants.append((f[len("librbpf_hip_"):-3], os.path.join(LIBDIR, f), {}))
variants += [("default rs1cs4", None, {"RBPF_RS": "1", "RBPF_CS": "4"}),
             ("default rs2cs1", None, {"RBPF_RS": "2", "RBPF_CS": "1"}),
             ("default rs1cs2", None, {"RBPF_RS": "1", "RBPF_CS": "2"})]
for name, lib, env_extra in variants:
    env = dict(os.environ)
    env.update(env_extra)
    if lib:
        env["RBPF_LIB_PATH"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "20",
                        "--no-cpu-baseline", "--no-smoother", "--no-large", "--no-traffic"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"{name:24s} FAILED: {r.stderr[-300:]}")
        continue
    j = json.loads(line[-1])
    print(f"{name:24s} value={j['value'] / 1e6:7.3f} M/s  ms/step={j['ms_per_step']:.4f}  "
          f"kernel_ms={j['roofline']['avg_launch_ms']:.4f}  GB/s={j['roofline']['achieved']:.0f}", flush=True)
