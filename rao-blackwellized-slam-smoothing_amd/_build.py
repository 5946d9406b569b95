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
SOURCES = ["rbpf_kernels.hip", "rbpf_step_sym.hip", "rbpf_api.hip", "rbpf_smoother.hip", "rbpf_shard.hip", "rbpf_multi.hip", "rbpf_plan.hip", "rbpf_resample.hip", "rbpf_sparse.hip"]
HEADERS = ["rbpf_model_dev.hpp", "rbpf_chol64.hpp", "rbpf_chol_small.hpp", "rbpf_chol_sweep.hpp", "rbpf_internal.hpp", "rbpf_device.hpp", "rbpf_ctx.hpp", "rbpf_plan.hpp", "rbpf_shard_state.hpp", "rbpf_sparse.hpp"]


def _stale() -> bool:
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "rbpf.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _includes(path, seen=None):
    """Local headers a source file pulls in (recursively): #include "x.hpp" next to it, <rbpf.h> from include/."""
    import re
    seen = seen if seen is not None else set()
    try:
        text = open(path).read()
    except OSError:
        return seen
    for name in re.findall(r'#include\s+"([^"]+)"', text):
        h = os.path.join(CSRC, name)
        if os.path.exists(h) and h not in seen:
            seen.add(h)
            _includes(h, seen)
    return seen


def build(force: bool = False, verbose: bool = False, defines=None, out: str = None) -> str:
    """hipcc --offload-arch=gfx950 -> lib/librbpf_hip.so (cross-compiles without a GPU).  One object per source file under
    lib/obj/, compiled in parallel and reused while neither the source nor a header it includes has changed, then one link.
    `defines` / `out` build a tuning variant (e.g. {"RBPF_UC": 8}) under another file name (its objects are not cached)."""
    from concurrent.futures import ThreadPoolExecutor
    target = out or LIBPATH
    if out is None and not force and not _stale():
        return LIBPATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 RBPF library")
    os.makedirs(LIBDIR, exist_ok=True)
    variant = bool(defines) or out is not None
    objdir = os.path.join(LIBDIR, "obj_variant" if variant else "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
             "-I" + os.path.join(ROOT, "include")] + [f"-D{k}={v}" for k, v in (defines or {}).items()]
    api_h = os.path.join(ROOT, "include", "rbpf.h")

    def compile_one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        deps = [path, api_h, os.path.abspath(__file__)] + sorted(_includes(path))
        if not force and not variant and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            return obj, None
        cmd = [hipcc] + flags + ["-c", path, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            return obj, res.stdout
        os.replace(obj + ".tmp", obj)
        return obj, None

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), max(1, (os.cpu_count() or 2) // 2))) as pool:
        results = list(pool.map(compile_one, SOURCES))
    errs = [e for _, e in results if e]
    if errs:
        raise RuntimeError("hipcc failed:\n" + "\n".join(errs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in results] + ["-o", target + ".tmp"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
