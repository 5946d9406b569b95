"""Builds the gfx950 shared library (C-ABI of include/rbpf.h) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "librbpf_hip.so")
SOURCES = ["rbpf_kernels.hip", "rbpf_api.hip", "rbpf_smoother.hip", "rbpf_shard.hip", "rbpf_plan.hip", "rbpf_resample.hip", "rbpf_sparse.hip"]
HEADERS = ["rbpf_chol64.hpp", "rbpf_chol_small.hpp", "rbpf_chol_sweep.hpp", "rbpf_internal.hpp", "rbpf_device.hpp", "rbpf_ctx.hpp", "rbpf_plan.hpp", "rbpf_shard_state.hpp", "rbpf_sparse.hpp"]


def _stale() -> bool:
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "rbpf.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defines=None, out: str = None) -> str:
    """hipcc --offload-arch=gfx950 -shared ... -> lib/librbpf_hip.so (cross-compiles without a GPU).
    `defines` / `out` build a tuning variant (e.g. {"RBPF_UC": 8}) under another file name."""
    target = out or LIBPATH
    if out is None and not force and not _stale():
        return LIBPATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 RBPF library")
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-Wno-unused-result", "-I" + os.path.join(ROOT, "include")]
    cmd += [f"-D{k}={v}" for k, v in (defines or {}).items()]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", target + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
