#!/usr/bin/env python3
"""Headline benchmark: forward Rao-Blackwellized particle filter throughput on synthetic slam-dense-mag trajectories at
the configuration BASELINE.json's metric is quoted on -- configs[2]: N = 65 536 particles, T = 3000, m = 512 basis
functions (nLin = 515), fp64 -- plus the smoother wall-clock of the same configuration.

  python bench.py --gpus N --steps K --warmup W

A "step" is one time step of particleFilter (src/particleFilter.m:100-218) over all particles: resample-gather,
dynModel, measModel, importance weights, normalisation and the Kalman map update.  `value` = particle-steps/s over the K
timed steps with all inputs resident in HBM.  The total particle count is fixed ("scaling": "strong"): --gpus N shards
the 65 536 particles over N ranks (multigpu.ShardedFilterSession).  One JSON line is printed by rank 0.

roofline: `achieved` = bytes the run's schedule has to move per launch of the step kernel (rbpf_timing
.scheduled_bytes_per_launch: the stored covariance is read every step and rewritten every lazy_depth-th step) / mean launch
time from HIP events on the library's stream; `frac` = achieved / 8 TB/s (<= 1 by construction).  The ratio against
SURVEY 8(d)'s full read + write per step is reported separately as `algorithmic_GBps` / `algorithmic_ratio` (it exceeds
the HBM peak when the lazy update elides writes, so it is not a roofline fraction).  `traffic` is measured in THIS run:
two child runs of the same workload under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, FETCH x2 on
gfx950 per MI355X_MICROARCH.md), null when rocprofv3 is unavailable.  See DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import csv
import glob
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

THETA_MAG = [650.0, 1.2, 200.0, 10.0]                       # examples/slam-dense-mag/main.m:23
HBM_PEAK_GBPS = 8000.0                                       # MI355X_MICROARCH.md: HBM3E 8 TB/s


def q_mag():
    import numpy as np
    # examples/slam-dense-mag/main.m:22
    return np.diag(np.concatenate((10 ** 2 * np.array([0.05 ** 2, 0.05 ** 2, 0.01 ** 2]),
                                   (np.array([0.01, 0.01, 0.3]) * np.pi / 180) ** 2)))


def usable_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(pkg, data, model, x0_lin, P0, R, m, T_s=150):
    """Times oracle/rbpf_oracle_c.c (the plain-C restatement of src/particleFilter.m) on this host's
    cores on a bounded sample of the same workload.  The oracle is only the thing *measured* here."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_c
    try:
        lib_path = oracle_c.build(native_dir=os.path.join(ROOT, "gpurun_out", "_oracle_native"))
        flags = "-O3 -march=native -fopenmp"
    except Exception:
        lib_path = oracle_c.build()
        flags = "-O3 -fopenmp"
    cores = usable_cores()
    N_s = 1024
    Q = q_mag()

    def run(T_s):
        rs = np.random.RandomState(123)
        rng = pkg.ReplayRNG(rs.random_sample((1, T_s - 1, N_s)), rs.standard_normal((1, T_s - 1, N_s, 6)))
        _, secs = oracle_c.particle_filter(pkg, model, data["dx"][:T_s - 1], data["y"][:T_s], data["initState"],
                                           x0_lin, P0, Q, R, N_s, 0.01, rng, n_threads=cores, want_full=False,
                                           lib_path=lib_path)
        return secs
    run(5)                                                     # first touch / thread pool
    secs = run(T_s)                                            # a FIXED sample (N = 1024, T = 150: about 15 s on 16 cores), so the figure is reproducible
    return {"value": N_s * T_s / secs, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"slam-dense-mag N={N_s} T={T_s} m={m} fp64, C restatement of particleFilter.m "
                      f"(gcc {flags}, OpenMP over particles), {secs:.1f} s"}


# ------------------------------------------------------------------------------------------------------------------
# the filter leg (headline and the extra configurations)
# ------------------------------------------------------------------------------------------------------------------
def roofline_of(tm, storage="fp64"):
    avg_ms = tm["ms"] / max(tm["launches"], 1)
    sched = tm["scheduled_bytes_per_launch"]
    ach = sched / (avg_ms * 1e-3) / 1e9
    alg = tm["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
            "kernel": "step_sym_kernel (all variants of a lazy cycle; a flush step = its two launches)" if storage in ("fp64sym", "fp32sym") else "step_kernel",
            "avg_launch_ms": avg_ms, "launches": tm["launches"],
            "scheduled_bytes_per_launch": sched,
            "algorithmic_bytes_per_launch": tm["bytes_per_launch"], "algorithmic_GBps": alg,
            "algorithmic_ratio": alg / HBM_PEAK_GBPS,
            "note": "achieved/frac: bytes the schedule has to move from / to memory -- every DISTINCT stored covariance read per step "
                    "(particles that share one, siblings and cousins of the lazy update, are processed on one XCD and read it through "
                    "its L2: counted on the device), the matrices written at every lazy_depth-th step (one per parent with children: "
                    "shared flush), the factor sets and states -- / HIP-event launch time.  `traffic` (HBM counters) / "
                    "hbm_counter_GBps / frac_hbm_counters price the same launch by the counters.  algorithmic_*: SURVEY 8d's one read + "
                    "one write of a full-square covariance per particle-step over the same time (not a roofline fraction: the "
                    "symmetric storage, the lazy update and the shared reads all elide bytes)"}


def filter_leg(pkg, datagen, N, m, T, K, W, seed, lazy_depth, inplace, storage, keep_history=True):
    """One single-GPU filter run: W warm-up steps, K timed steps.  keep_history: the state history and the ancestor table
    (particleFilter.m:117-118,233: xn_traj / traj_sample_iwmax) are written inside the timed steps, so the run could return the
    reference's full output set."""
    import numpy as np
    Q = q_mag()
    data = datagen.bean_6D(T, Q, THETA_MAG, 0.01, seed=seed)
    model, x0_lin, P0, R = pkg.dense_mag_prior(m, data["LL"], THETA_MAG)
    with pkg.FilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R, N, 0.01,
                           rng=pkg.PhiloxRNG(seed), keep_history=keep_history, lazy_depth=lazy_depth, inplace=inplace,
                           storage=storage) as sess:
        sess.advance(W)
        sess.sync()
        sess.timing(enable=True)
        t0 = time.perf_counter()
        sess.advance(K)
        sess.sync()
        dt_s = time.perf_counter() - t0
        tm = sess.timing(reset=True)
        chk = sess.finish(want=("traj_mean", "traj_sample_iwmax") if keep_history else ("traj_mean",))
    if not np.all(np.isfinite(chk["traj_mean"][:, :W + K])):
        raise RuntimeError("non-finite filter output")
    if keep_history and not np.all(np.isfinite(chk["traj_sample_iwmax"][:, :W + K])):
        raise RuntimeError("non-finite back-traced trajectory")
    return {"value": N * K / dt_s, "unit": "particle-steps/s", "steps": K, "warmup": W, "ms_per_step": dt_s / K * 1e3,
            "roofline": roofline_of(tm, storage)}, data, model, x0_lin, P0, R


def bank_bytes_per_particle(n, storage):
    """Bytes of one stored covariance (include/rbpf.h rbpf_options.storage)."""
    if storage in ("fp64sym", "fp32sym"):                          # lower block triangle in 64 x 64 tiles + border rows
        mc = (n // 128) * 128
        ch = mc // 64
        return (8.0 if storage == "fp64sym" else 4.0) * (ch * (ch + 1) // 2 * 4096 + (n - mc) * ((n + 1) // 2 * 2))
    return n * n * (8.0 if storage == "fp64" else 4.0)


def workload_string(N_total, T, m, n, storage, lazy_depth, world, single_bank):
    prec = {"fp64": "fp64", "fp64sym": "fp64, symmetric storage (lower block triangle)",
            "fp32sym": "fp64 arithmetic / fp32 covariance storage, lower block triangle"}.get(storage, "fp64 arithmetic / fp32 covariance storage")
    bank = "single covariance bank rewritten in place" if single_bank else "ping-pong covariance banks"
    return (f"slam-dense-mag N={N_total} T={T} m={m} (nLin={n}) {prec}, forward filter, {world} GPU, lazy_depth {lazy_depth}, "
            f"{bank}")


# ------------------------------------------------------------------------------------------------------------------
# HBM traffic of the step kernel, measured in this run (child processes under rocprofv3 --pmc, one counter per pass)
# ------------------------------------------------------------------------------------------------------------------
def under_profiler():
    return any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or \
        "rocprofiler" in os.environ.get("LD_PRELOAD", "")


def measure_traffic(args, lazy_depth):
    """Mean HBM bytes per launch of the step kernel: 2 * FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH_SIZE reports half of
    wide coalesced reads on gfx950).  Steps 1.. of a (1 + 4 * lazy_depth)-step child run, i.e. four whole lazy cycles;
    the two launches of an in-place flush count as one step."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if under_profiler():
        return None, "bench.py itself runs under a profiler: nested counter passes skipped"
    cyc = max(lazy_depth, 1)
    steps = 1 + 4 * cyc
    out = {}
    tmp = tempfile.mkdtemp(prefix="rbpf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--traffic-child", "--steps", str(steps), "--warmup", "0",
                   "--particles", str(args.particles), "--m", str(args.m), "--T", str(args.T), "--seed", str(args.seed),
                   "--lazy-depth", str(args.lazy_depth), "--inplace", str(args.inplace), "--storage", args.storage]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stdout[-300:]}"
            rows = [x for x in csv.DictReader(open(files[0])) if ("step_kernel" in x.get("Kernel_Name", "") or "step_sym_kernel" in x.get("Kernel_Name", "")) and x.get("Counter_Name") == ctr]
            rows.sort(key=lambda x: int(x.get("Dispatch_Id", 0)))
            if len(rows) < steps:
                return None, f"{ctr}: {len(rows)} step-kernel dispatches, expected >= {steps}"
            vals = [float(x["Counter_Value"]) for x in rows[1:]]           # drop t = 0 (reads the broadcast prior)
            out[ctr] = sum(vals) * 1024.0 / (steps - 1)
    except Exception as exc:                                               # report, never hide
        return None, f"traffic measurement failed: {exc}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"bytes_per_launch": 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"], "FETCH_SIZE_bytes_x2": 2.0 * out["FETCH_SIZE"],
            "WRITE_SIZE_bytes": out["WRITE_SIZE"], "steps_averaged": steps - 1}, "ok"


def traffic_child(args):
    """Child of measure_traffic: the same filter workload, nothing else (runs under rocprofv3 --pmc)."""
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    datagen = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    Q = q_mag()
    data = datagen.bean_6D(args.T, Q, THETA_MAG, 0.01, seed=args.seed)
    model, x0_lin, P0, R = pkg.dense_mag_prior(args.m, data["LL"], THETA_MAG)
    with pkg.FilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R, args.particles, 0.01,
                           rng=pkg.PhiloxRNG(args.seed), keep_history=False, lazy_depth=args.lazy_depth, inplace=args.inplace,
                           storage=args.storage) as sess:
        sess.advance(args.steps)
        sess.sync()


# ------------------------------------------------------------------------------------------------------------------
# smoother legs
# ------------------------------------------------------------------------------------------------------------------
def smoother_reference_size(pkg, datagen):
    """Wall-clock of the two conditional particle smoothers at the reference's own dense-mag size
    (N_P=100, T=192, m=512: run_dense3D_magfield.m:85,134; generateData_dense.m:184-187), N_K=3, device Philox."""
    import numpy as np
    Q = q_mag()
    d = datagen.bean_6D(192, Q, THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = pkg.dense_mag_prior(512, d["LL"], THETA_MAG)
    out = {"workload": "slam-dense-mag N_P=100 T=192 m=512 N_K=3 fp64 (the reference's own size)", "unit": "s"}
    for name, f in (("information_form", pkg.particleSmootherInformationForm), ("covariance_form", pkg.particleSmoother)):
        t0 = time.perf_counter()
        XNK, _, _ = f(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, Q, R, 100, 3, 0.01,
                      rng=pkg.PhiloxRNG(3))
        out[name] = round(time.perf_counter() - t0, 3)
        out[name + "_pos_rmse_m"] = round(float(np.sqrt(np.mean((XNK[0:3, :, -1] - d["pos"]) ** 2))), 4)
    return out


def smoother_share_full(pkg, datagen, N_share, T, m, N_K, seed, **opts):
    """The metric's smoother: particleSmootherInformationForm on slam-dense-mag m=512, T=3000, complete, for the per-GPU share
    of N=65 536 at 8 GPUs (N_share = 8192 particles; the information-form state of all 65 536 -- 2 x 139 GB of Imat next to
    the covariances -- does not fit one GPU).  Wall clock of the whole call, N_K iterations.
    opts: rbpf_options of the information form (lazy_depth, chol_refresh)."""
    import numpy as np
    Q = q_mag()
    d = datagen.bean_6D(T, Q, THETA_MAG, 0.01, seed=seed)
    mdl, x0, P0, R = pkg.dense_mag_prior(m, d["LL"], THETA_MAG)
    marks = []
    t0 = time.perf_counter()
    XNK, _, _ = pkg.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0,
                                                    Q, R, N_share, N_K, 0.01, False, lambda *a: marks.append(time.perf_counter()),
                                                    rng=pkg.PhiloxRNG(3), **opts)
    secs = time.perf_counter() - t0
    its = [round(b - a, 3) for a, b in zip([t0] + marks[:-1], marks)]
    share = ("1/8 of N=65536" if N_share == 8192 else "the metric's full N" if N_share == 65536 else f"{N_share / 65536:g} of N=65536")
    K_ref = pkg.chol_refresh_in_use(mdl, opts.get("chol_refresh", 0))
    how = ("ancestor-weight factors carried along the lineages and never refactorised after the first step, no information matrix stored"
           if K_ref >= T - 1 else
           f"ancestor-weight factors carried along the lineages, refactorised every {K_ref} steps" + (" from the origin (no information matrix stored)" if opts.get("info_rebuild") else "")
           if K_ref > 1 else "chol(Imat_i + ImatAddt) from scratch for every particle at every step (the reference's arithmetic)")
    if opts.get("inplace", 0) > 0:
        how += ", one covariance bank rewritten in place"
    return {"workload": f"slam-dense-mag N_P={N_share} ({share}) T={T} m={m} N_K={N_K} fp64, information form, complete run, {how}",
            "options": opts, "chol_refresh_in_use": K_ref, "seconds": round(secs, 3), "seconds_per_iteration": its, "unit": "s",
            "ms_per_time_step_with_ancestor_sampling": round(its[-1] / T * 1e3, 3) if len(its) > 1 else None,
            "finite": bool(np.all(np.isfinite(XNK))),
            "pos_rmse_m_last_iteration": round(float(np.sqrt(np.mean((XNK[0:3, :, -1] - d["pos"]) ** 2))), 4)}


def smoother_radio_large(pkg, datagen):
    """BASELINE.json configs[3] on one GPU: dense-radio N=65536 (T=48, m=128, N_K=3) through particleSmootherInformationForm."""
    import numpy as np
    T = 48
    Qr = datagen.radio_Q(T, "square_3D")
    th = [0.25, 2.0, 0.01]                                                       # examples/slam-dense-radio/main.m:24
    d = datagen.planar_heading(T, Qr, th, 1.0, seed=1, nLL=4, traj="square_3D")
    mdl, x0, P0, R = pkg.dense_radio_prior(128, d["LL"], th)
    runs, runs_c = [], []
    for _ in range(2):                                                           # the first run also pays for the first touch of 30 GB
        t0 = time.perf_counter()
        XNK, _, _ = pkg.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],
                                                        x0, P0, Qr, R, 65536, 3, 1.0, rng=pkg.PhiloxRNG(3), lazy_depth=3, chol_refresh=16)
        runs.append(round(time.perf_counter() - t0, 3))
    for _ in range(2):                                                           # the reference's arithmetic: from scratch at every step
        t0 = time.perf_counter()
        XNKc, _, _ = pkg.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],
                                                         x0, P0, Qr, R, 65536, 3, 1.0, rng=pkg.PhiloxRNG(3), chol_refresh=1)
        runs_c.append(round(time.perf_counter() - t0, 3))
    return {"workload": "slam-dense-radio N_P=65536 T=48 m=128 N_K=3 fp64, information form (BASELINE.json configs[3] on one GPU), lazy_depth 3, "
                        "ancestor-weight factors carried and refactorised every 16 steps",
            "seconds": min(runs), "runs": runs, "seconds_fresh_factorisation_every_step": min(runs_c), "runs_fresh_factorisation_every_step": runs_c,
            "same_trajectory_draws": bool(np.allclose(XNK, XNKc, rtol=1e-9, atol=1e-11)),
            "unit": "s", "finite": bool(np.all(np.isfinite(XNK)) and np.all(np.isfinite(XNKc)))}


def smoother_sharded_leg(pkg, mg, datagen, torch, dist, N_local, m, T_s, T_full, N_K, seed, rank, world, lazy_depth, chol_refresh=0):
    """The metric's smoother as the N-GPU job runs it: particleSmootherInformationForm with world * N_local particles sharded over
    the ranks (ShardedSmootherSession: all_gather of the forward bank and of the ancestor log-weights, all_to_all of the migrating
    particle records), timed over the first T_s of the T_full time steps of both CPF-AS iterations; max over ranks.  The full-length
    figure is an extrapolation by time steps (every step of an iteration costs the same) and is labelled as one."""
    Q = q_mag()
    d = datagen.bean_6D(T_s, Q, THETA_MAG, 0.01, seed=seed)
    model, x0_lin, P0, R = pkg.dense_mag_prior(m, d["LL"], THETA_MAG)
    sess = mg.ShardedSmootherSession(model, d["dx"], d["y"], d["initState"], x0_lin, P0, Q, R, N_local, N_K, 0.01,
                                     rng=pkg.PhiloxRNG(seed), rank=rank, world=world, lazy_depth=lazy_depth, chol_refresh=chol_refresh,
                                     force_collectives=(world == 1), storage="fp64sym" if m == 512 else "fp64")
    sess_K = int(sess.chol_refresh)                           # the K in use (0 = automatic was resolved by the session)
    try:
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        XNK, XLK, PK = sess.run()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        it = list(sess.stats.get("iter_s", []))
        st = {k: v for k, v in sess.stats.items() if k not in ("phase_s", "iter_s")}
    finally:
        sess.close()
    tt = torch.tensor([dt_s] + it, dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    tt = [float(v) for v in tt.tolist()]
    import numpy as np
    if not (np.all(np.isfinite(XNK)) and np.all(np.isfinite(PK))):
        raise RuntimeError("non-finite smoother output")
    per_step = [v / T_s for v in tt[1:]]
    return {"workload": f"slam-dense-mag particleSmootherInformationForm N_P={N_local * world} ({N_local} per GPU x {world}) m={m} N_K={N_K}, "
                        f"first {T_s} of {T_full} time steps of every iteration, lazy_depth {lazy_depth}, "
                        + (f"ancestor-weight factors carried, refreshed every {sess_K} steps" if sess_K > 1 else "from-scratch factorisation every step"),
            "seconds": tt[0], "seconds_per_iteration": tt[1:], "ms_per_time_step_per_iteration": [v * 1e3 for v in per_step],
            "extrapolated_full_T_seconds": sum(per_step) * T_full, "sharding": st}


def smoother_sharded_radio_leg(pkg, mg, datagen, torch, dist, N_local, seed, rank, world, **opts):
    """BASELINE.json configs[3] as written: slam-dense-radio (m = 128, T = 48, N_K = 3), information-form smoother sharded over the
    ranks, 8192 particles per GPU (N = 65 536 at 8 GPUs), complete run; max over ranks."""
    import numpy as np
    T, N_K, th = 48, 3, [0.25, 2.0, 0.01]
    Qr = datagen.radio_Q(T, "square_3D")
    d = datagen.planar_heading(T, Qr, th, 1.0, seed=1, nLL=4, traj="square_3D")
    mdl, x0, P0, R = pkg.dense_radio_prior(128, d["LL"], th)
    sess = mg.ShardedSmootherSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, Qr, R, N_local, N_K, 1.0, rng=pkg.PhiloxRNG(seed),
                                     rank=rank, world=world, force_collectives=(world == 1), **opts)
    try:
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        XNK, XLK, PK = sess.run()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        st = {k: v for k, v in sess.stats.items() if k not in ("phase_s", "iter_s")}
    finally:
        sess.close()
    tt = torch.tensor([dt_s], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return {"workload": f"slam-dense-radio particleSmootherInformationForm N_P={N_local * world} ({N_local} per GPU x {world}) m=128 T={T} "
                        f"N_K={N_K}, complete run" + (f", options {opts}" if opts else ""),
            "seconds": float(tt.item()), "finite": bool(np.all(np.isfinite(XNK)) and np.all(np.isfinite(PK))), "sharding": st}


def smoother_kernel_roofline(pkg):
    """The smoothers' dominant kernel on its own: the batched ancestor-weight factorisation (particleSmoother.m:221-229,
    particleSmootherInformationForm.m:224-236) of 2048 matrices of the m=512 size (n=515), timed with HIP events inside
    the library (rbpf_chol_weights).  Bound: fp64 matrix cores (n^3/3 flop per matrix against the 78.6 TFLOP/s dense peak)."""
    import numpy as np
    rs = np.random.RandomState(0)
    M, B = 515, 2048
    A = rs.standard_normal((B, M, 24))
    S = A @ np.transpose(A, (0, 2, 1)) / 24 + np.eye(M)
    e = rs.standard_normal((B, M))
    pkg.chol_weights(S[:64], e[:64])
    logw, status, ms = pkg.chol_weights(S, e, reps=5)
    flops = B * M ** 3 / 3.0
    ach = flops / (ms * 1e-3) / 1e12
    return {"kernel": "chol_solve64_kernel", "workload": f"{B} matrices, n={M}, fp64, standalone (covariance-form loaders; the figure inside the smoother is "
                                                         "smoother.kernel_roofline_in_smoother)", "bound": "mfma", "achieved": ach, "peak": 78.6,
            "unit": "TFLOP/s", "frac": ach / 78.6, "avg_launch_ms": ms, "algorithmic_flop_per_launch": flops, "traffic": None,
            "finite": bool(np.all(np.isfinite(logw))) and status == 0}


def smoother_trace_child(args):
    """Child of smoother_kernel_in_smoother: a short run of the metric's smoother configuration, nothing else (runs under
    rocprofv3 --kernel-trace --stats or --pmc).  --chol-refresh selects the ancestor-weight arithmetic (0: the library default)."""
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    datagen = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    Q = q_mag()
    d = datagen.bean_6D(args.T, Q, THETA_MAG, 0.01, seed=args.seed)
    mdl, x0, P0, R = pkg.dense_mag_prior(args.m, d["LL"], THETA_MAG)
    pkg.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, Q, R,
                                        args.particles, 2, 0.01, rng=pkg.PhiloxRNG(3), lazy_depth=3, storage="fp64sym" if args.m == 512 else "fp64",
                                        chol_refresh=args.chol_refresh)


def sweep_factor_bytes(n):
    """Bytes of one carried factor in the sweep layout (csrc/rbpf_chol_sweep.hpp: sweep_factor_doubles)."""
    NS = (n + 1 + 63) >> 6
    t = (n + 1) & 63
    tailc = 1 <= t <= 8 and n + 1 > 64
    if not tailc:
        return 8.0 * sum(64 * (NS - (k >> 6)) for k in range(n))
    M = NS - 1
    return 8.0 * (sum(64 * (M - (min(k, 64 * M) >> 6)) for k in range(n)) + 8 * n)


def smoother_kernel_in_smoother(args, N_P=8192):
    """The dominant kernel of the ancestor-weight step AS THE SMOOTHER RUNS IT, from child runs of the metric's smoother configuration
    (N_P = 8192, m = 512, lazy_depth 3, N_K = 2) under rocprofv3:

    * the library default -- carried factors: `chol_sweep_kernel` (one read of the ancestor's factor + one write of the particle's,
      2 x 1.21 MB; HBM-bound), mean launch time from --kernel-trace --stats over T = 70 steps, HBM traffic per launch from two
      --pmc passes (2 x FETCH_SIZE + WRITE_SIZE, KiB counters) over T = 40 steps; the refresh's kernels listed beside it;
    * chol_refresh = 1 -- the from-scratch factorisation `chol_solve64_kernel` (n^3 / 3 flop per particle against the fp64 matrix
      peak), T = 24 steps."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {"error": "rocprofv3 not on PATH"}
    if under_profiler():
        return {"error": "bench.py itself runs under a profiler: nested trace pass skipped"}
    n = args.m + 3
    env = dict(os.environ, TMPDIR="/tmp")

    def child(mode, T, K):
        tmp = tempfile.mkdtemp(prefix="rbpf_trace_", dir="/tmp")
        try:
            cmd = [exe] + mode + ["--output-format", "csv", "-d", tmp, "-o", "sm", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                                  "--smoother-trace-child", "--particles", str(N_P), "--m", str(args.m), "--T", str(T), "--seed", str(args.seed),
                                  "--chol-refresh", str(K)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
            pat = "*kernel_stats.csv" if "--stats" in mode else "*counter_collection.csv"
            files = glob.glob(os.path.join(tmp, "**", pat), recursive=True)
            if r.returncode != 0 or not files:
                raise RuntimeError(f"rocprofv3 {' '.join(mode)} failed (rc {r.returncode}): {r.stdout[-300:]}")
            return list(csv.DictReader(open(files[0])))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    def kernels(rows):
        ks = {}
        for x in rows:
            name = x["Name"].split("(")[0].replace("void rbpf::", "").replace("rbpf::", "")
            if any(t in name for t in ("chol_", "step_sym_kernel", "step_kernel", "gemm_kernel", "sweep_")):
                ks[name] = {"calls": int(x["Calls"]), "avg_launch_ms": float(x["AverageNs"]) / 1e6, "share_of_gpu_time_pct": float(x["Percentage"])}
        return ks

    out = {"workload": f"slam-dense-mag particleSmootherInformationForm N_P={N_P} m={args.m} N_K=2 lazy_depth 3, block-lower P (rocprofv3 child runs)"}
    # ---- the default: carried factors ----
    try:
        ks = kernels(child(["--kernel-trace", "--stats"], 70, 0))
        out["kernels"] = ks
        sw = [(k, v) for k, v in ks.items() if k.startswith("chol_sweep_kernel")]
        if not sw:
            raise RuntimeError("no chol_sweep_kernel in the trace of the default configuration")
        k, v = max(sw, key=lambda kv: kv[1]["calls"])
        by = 2.0 * sweep_factor_bytes(n) * N_P
        ach = by / (v["avg_launch_ms"] * 1e-3) / 1e9
        out.update({"kernel": k, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                    "avg_launch_ms": v["avg_launch_ms"], "launches": v["calls"], "algorithmic_bytes_per_launch": by, "traffic": None})
        tr = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = [x for x in child(["--pmc", ctr], 40, 0) if "chol_sweep_kernel" in x.get("Kernel_Name", "") and x.get("Counter_Name") == ctr]
            if not rows:
                raise RuntimeError(f"{ctr}: no chol_sweep_kernel dispatch in the counter pass")
            tr[ctr] = sum(float(x["Counter_Value"]) for x in rows) * 1024.0 / len(rows)
        traffic = 2.0 * tr["FETCH_SIZE"] + tr["WRITE_SIZE"]
        out.update({"traffic": traffic, "traffic_detail": {"FETCH_SIZE_bytes_x2": 2.0 * tr["FETCH_SIZE"], "WRITE_SIZE_bytes": tr["WRITE_SIZE"]},
                    "traffic_over_algorithmic": traffic / by, "hbm_counter_GBps": traffic / (v["avg_launch_ms"] * 1e-3) / 1e9,
                    "frac_hbm_counters": traffic / (v["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    except Exception as exc:                                               # report, never hide
        out["error"] = f"{type(exc).__name__}: {exc}"
    # ---- chol_refresh = 1: the reference's arithmetic ----
    try:
        ks = kernels(child(["--kernel-trace", "--stats"], 24, 1))
        chol = [(k, v) for k, v in ks.items() if "chol_solve" in k]
        k, v = max(chol, key=lambda kv: kv[1]["calls"])
        flops = N_P * n ** 3 / 3.0
        ach = flops / (v["avg_launch_ms"] * 1e-3) / 1e12
        out["fresh_factorisation"] = {"kernel": k, "bound": "mfma", "achieved": ach, "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6,
                                      "avg_launch_ms": v["avg_launch_ms"], "launches": v["calls"], "algorithmic_flop_per_launch": flops, "kernels": ks}
    except Exception as exc:
        out["fresh_factorisation"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def filter_full_run(pkg, datagen, N, m, T, seed, lazy_depth, inplace, storage, window=500):
    """A COMPLETE filter run (particleFilter.m:100-218 loops to N_T): all T steps of the headline configuration, timed as a whole and
    in windows of `window` steps -- how many distinct stored covariances a step reads depends on the ancestry, so the throughput of the
    first steps is not by construction that of the rest.  Per window: particle-steps/s (host clock around advance + sync), the step
    kernels' own time (HIP events) and the bytes the schedule moved per step (counted on the device)."""
    import numpy as np
    Q = q_mag()
    data = datagen.bean_6D(T, Q, THETA_MAG, 0.01, seed=seed)
    model, x0_lin, P0, R = pkg.dense_mag_prior(m, data["LL"], THETA_MAG)
    wins = []
    with pkg.FilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R, N, 0.01, rng=pkg.PhiloxRNG(seed),
                           keep_history=True, lazy_depth=lazy_depth, inplace=inplace, storage=storage) as sess:
        sess.timing(enable=True)
        t_all = time.perf_counter()
        done = 0
        while done < T:
            k = min(window, T - done)
            t0 = time.perf_counter()
            sess.advance(k)
            sess.sync()
            dt_s = time.perf_counter() - t0
            tm = sess.timing(reset=True)
            wins.append({"steps": [done, done + k], "particle_steps_per_s": N * k / dt_s, "ms_per_step": dt_s / k * 1e3,
                         "kernel_ms_per_step": tm["ms"] / max(tm["launches"], 1),
                         "scheduled_GB_per_step": tm["scheduled_bytes_per_launch"] / 1e9})
            done += k
        secs = time.perf_counter() - t_all
        chk = sess.finish(want=("traj_mean", "traj_sample_iwmax"))
    ok = bool(np.all(np.isfinite(chk["traj_mean"])) and np.all(np.isfinite(chk["traj_sample_iwmax"])))
    return {"workload": f"slam-dense-mag N={N} T={T} m={m} {storage}, lazy_depth {lazy_depth}: all {T} steps, state history and ancestor table kept",
            "seconds": secs, "particle_steps_per_s": N * T / secs, "ms_per_step": secs / T * 1e3, "windows": wins, "finite": ok}


def smoother_sweep_roofline(pkg):
    """The dominant kernel of the smoother WITH carried factors, on its own: one up/down-date sweep per particle over its
    ancestor's factor (rbpf_chol_sweep.hpp), 8192 factors of nLin = 515, n_y = 3: HBM-bound, one read + one write of the stored
    factor (1.21 MB in sweep layout)."""
    import numpy as np
    n, d, batch = 515, 3, 8192
    rs = np.random.RandomState(7)
    L = np.tril(0.02 * rs.randn(n + 1, n + 1)) + np.diag(1.0 + rs.random_sample(n + 1))
    U = rs.randn(d, n) / np.sqrt(n)
    V = 0.3 * rs.randn(d, n) / np.sqrt(n)
    Lo, logw, status, ms = pkg.chol_sweep_probe(L, U, V, rs.randn(d), batch=batch, reps=10)
    M = 8                                                     # slots below the compact tail: 64 * (64 (b M - b (b - 1) / 2) + r (M - b)) + 8 k
    per = 8.0 * (sum(64 * (M - (k >> 6)) for k in range(64 * M)) + 8 * n)      # bytes of one stored factor (columns 0 .. n-1)
    by = 2.0 * per * batch
    return {"kernel": "chol_sweep_kernel<3,9>", "workload": f"{batch} factors, nLin={n}, n_y={d}, fp64", "bound": "hbm",
            "achieved": by / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / 8000.0,
            "avg_launch_ms": ms, "algorithmic_bytes_per_launch": by, "traffic": None, "finite": bool(np.isfinite(logw)) and status == 0}


def guarded(f, *a):
    try:
        return f(*a)
    except Exception as exc:                                   # report, never hide
        return {"error": f"{type(exc).__name__}: {exc}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--particles", type=int, default=65536, help="particles: the total (strong scaling) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--T", type=int, default=3000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-smoother", action="store_true", help="skip every smoother leg")
    ap.add_argument("--no-smoother-full", action="store_true", help="skip the complete T=3000 smoother run of the per-GPU share (about a minute)")
    ap.add_argument("--no-smoother-largest", action="store_true", help="skip the complete smoother run at the metric's full N_P = 65536 on this one GPU "
                    "(about 3.5 minutes; also skipped -- and said so in the line -- when the run is already past --time-budget)")
    ap.add_argument("--smoother-32k", action="store_true", help="also run the complete smoother at N_P = 32768 with the library defaults (two banks, information "
                    "matrices stored at the refreshes: the largest size that configuration holds; about 2 minutes)")
    ap.add_argument("--no-filter-full", action="store_true", help="skip the complete T-step filter run (about 40 s)")
    ap.add_argument("--driver", default="torchrun", choices=["torchrun", "inlib"],
                    help="inlib: ALSO time the in-library multi-device driver (rbpf_options.n_devices, csrc/rbpf_multi.hip: one host process, one thread "
                         "per GPU, RCCL issued by the library -- the path a MATLAB session behind the MEX gateway uses) on the same particles: rank 0 "
                         "calls the one-shot entry point over all --gpus devices while the other ranks wait; reported under `inlib_driver`")
    ap.add_argument("--inlib-steps", type=int, default=120, help="time steps of the --driver inlib run (a complete one-shot call: upload, steps, extraction)")
    ap.add_argument("--time-budget", type=float, default=300.0, help="seconds of wall clock after which the longest optional leg (the N_P = 32768 smoother) is not started")
    ap.add_argument("--smoother-trace-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--chol-refresh", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-large", action="store_true", help="skip the extra filter configurations (configs[1], configs[4] share) and the N=65536 radio smoother")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child runs (roofline.traffic = null)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lazy-depth", type=int, default=4, help="rewrite the covariances every C-th step only (0/1: every step)")
    ap.add_argument("--inplace", type=int, default=0, help="single covariance bank rewritten in place: 1 on, -1 off, 0 automatic (when two banks do not fit)")
    ap.add_argument("--storage", default="fp64sym", choices=["fp64", "fp32", "fp64sym", "fp32sym"],
                    help="how the covariance banks are STORED (arithmetic is fp64): fp64sym = fp64, lower block triangle only (default; results within 1e-9 "
                         "of fp64), fp64 = full square as the reference holds them, fp32 = full square in float")
    ap.add_argument("--force-sharded", action="store_true", help="use the sharded session even at --gpus 1 (testing)")
    ap.add_argument("--smoother-steps", type=int, default=150, help="time steps per iteration of the sharded smoother leg (--gpus > 1); --T runs it complete")
    ap.add_argument("--smoother-particles", type=int, default=8192, help="particles per GPU of the sharded smoother leg")
    ap.add_argument("--smoother-timeout", type=float, default=420.0, help="watchdog of EACH sharded smoother leg, seconds (a leg that does not return ends every rank with exit code 3)")
    args = ap.parse_args()

    if args.traffic_child:
        traffic_child(args)
        return
    if args.smoother_trace_child:
        smoother_trace_child(args)
        return
    t_start = time.perf_counter()

    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the RBPF path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    # RCCL prints a version banner to STDOUT when its first communicator comes up; the contract is ONE JSON line on stdout, so the
    # process's stdout (file descriptor 1) points at stderr until that line is printed
    saved_stdout = None
    if sharded:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(obj), flush=True)
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    datagen = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    K, W, T = args.steps, args.warmup, args.T
    if K + W > T:
        raise SystemExit(f"--steps + --warmup must be <= T={T}")
    if args.scaling == "strong":
        if args.particles % world:
            raise SystemExit("--particles must be a multiple of --gpus")
        N_local, N_total = args.particles // world, args.particles
    else:
        N_local, N_total = args.particles, args.particles * world
    Q = q_mag()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    shard_stats = None
    if sharded:
        data = datagen.bean_6D(T, Q, THETA_MAG, 0.01, seed=args.seed)           # data seed 1
        model, x0_lin, P0, R = pkg.dense_mag_prior(args.m, data["LL"], THETA_MAG)
        mg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.multigpu")
        sess = mg.ShardedFilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R,
                                       N_local, 0.01, rng=pkg.PhiloxRNG(args.seed), rank=rank, world=world,
                                       lazy_depth=args.lazy_depth, storage=args.storage, force_collectives=(world == 1))
        sess.advance(W)
        sess.sync()
        sess.timing(enable=True)
        barrier()
        t0 = time.perf_counter()
        sess.advance(K)
        sess.sync()
        barrier()
        dt_s = time.perf_counter() - t0
        tm = sess.timing(reset=True)
        sess.timing(enable=False)
        tt = torch.tensor([dt_s], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_s = float(tt.item())
        chk = sess.finish(want=("traj_mean",))
        if not np.all(np.isfinite(chk["traj_mean"][:, :W + K])):
            raise SystemExit("non-finite filter output")
        shard_stats = dict(getattr(sess, "stats", None) or {})
        shard_stats.pop("phase_s", None)                  # host enqueue times only: the timed run has one sync per step
        sess.close()
        # diagnostic pass (not timed into `value`): the same steps with a stream synchronisation after every phase, so that
        # the first multi-GPU run shows where a step's time goes (gather / normalise+plan / exchange / step kernel)
        diag_steps = min(24, K)
        try:
            sess = mg.ShardedFilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R,
                                           N_local, 0.01, rng=pkg.PhiloxRNG(args.seed), rank=rank, world=world,
                                           lazy_depth=args.lazy_depth, storage=args.storage, sync_phases=True, force_collectives=(world == 1))
            sess.advance(6)
            sess.sync()
            sess.stats["phase_s"] = dict(gather=0.0, normalise=0.0, plan=0.0, exchange=0.0, step=0.0)
            sess.stats["steps"] = 0
            barrier()
            sess.advance(diag_steps)
            sess.sync()
            ph = sess.stats.get("phase_s", {})
            shard_stats["phase_ms_per_step_synchronised"] = {k: round(v / max(diag_steps, 1) * 1e3, 4) for k, v in ph.items()}
            sess.close()
        except Exception as exc:                                   # the diagnostic must never cost the measurement
            shard_stats["phase_ms_per_step_synchronised"] = {"error": f"{type(exc).__name__}: {exc}"}
        head = {"value": N_total * K / dt_s, "ms_per_step": dt_s / K * 1e3, "roofline": roofline_of(tm, args.storage)}
        single_bank = False
    else:
        barrier()
        head, data, model, x0_lin, P0, R = filter_leg(pkg, datagen, N_local, args.m, T, K, W, args.seed, args.lazy_depth,
                                                     args.inplace, args.storage)
        _free_b, tot_b = torch.cuda.mem_get_info()
        n_ = model.nLin
        single_bank = args.inplace > 0 or (args.inplace == 0 and args.lazy_depth >= 2 and
                                           2.0 * N_local * bank_bytes_per_particle(n_, args.storage) > 0.85 * tot_b)

    inlib = None
    if args.driver == "inlib":
        # every rank has finished its own leg; rank 0 now owns all --gpus devices through the library's threads
        barrier()
        if rank == 0:
            try:
                Ti = min(args.inlib_steps, T)
                di = datagen.bean_6D(Ti, Q, THETA_MAG, 0.01, seed=args.seed)
                mi, x0i, P0i, Ri = pkg.dense_mag_prior(args.m, di["LL"], THETA_MAG)
                kw = dict(n_devices=world, device_ids=list(range(world))) if world > 1 else dict(n_devices=1, device_ids=[local_rank])
                t0 = time.perf_counter()
                out = pkg.particleFilter(mi.dynModel, mi.measModel, di["dx"], di["y"], di["initState"], x0i, P0i, Q, Ri, N_total, 0.01,
                                         rng=pkg.PhiloxRNG(args.seed), want_xn_traj=False, lazy_depth=args.lazy_depth, storage=args.storage, **kw)
                dt_i = time.perf_counter() - t0
                inlib = {"workload": f"slam-dense-mag N={N_total} m={args.m} {args.storage} lazy_depth {args.lazy_depth}, {Ti} time steps, ONE host process, "
                                     f"{world} device(s) through rbpf_options.n_devices (one-shot rbpf_particle_filter: upload + steps + extraction)",
                         "seconds": dt_i, "particle_steps_per_s_incl_setup": N_total * Ti / dt_i, "finite": bool(np.all(np.isfinite(out[1])))}
            except Exception as exc:                               # report, never hide
                inlib = {"error": f"{type(exc).__name__}: {exc}"}
        barrier()

    if rank == 0:
        n = model.nLin
        line = {
            "metric": "particle-steps/s (filter) + smoother wall-clock, N=65k T=3k", "value": head["value"], "unit": "particle-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64" if args.storage in ("fp64", "fp64sym") else "f64 arithmetic, f32 covariance storage", "data": "synthetic",
            "config": {"workload": workload_string(N_total, T, args.m, n, args.storage, args.lazy_depth, world, single_bank),
                       "baseline_config": "BASELINE.json configs[2] (filter part)" if (N_total, T, args.m) == (65536, 3000, 512) and args.storage in ("fp64", "fp64sym") else "custom",
                       "particles_total": N_total, "particles_per_gpu": N_local, "rng": "device Philox4x32-10", "data_seed": args.seed,
                       "lazy_depth": args.lazy_depth, "inplace": args.inplace, "storage": args.storage, "filter_seed": args.seed, "keep_history": True},
            "roofline": head["roofline"],
        }
        if shard_stats:
            line["config"]["sharding"] = shard_stats
        if inlib is not None:
            line["inlib_driver"] = inlib
        solo = world == 1 and not args.force_sharded
        if solo and not args.no_traffic:
            tr, why = measure_traffic(args, args.lazy_depth)
            if tr:
                line["roofline"]["traffic"] = tr["bytes_per_launch"]
                line["roofline"]["traffic_detail"] = tr
                line["roofline"]["traffic_over_scheduled"] = tr["bytes_per_launch"] / line["roofline"]["scheduled_bytes_per_launch"]
                line["roofline"]["hbm_counter_GBps"] = tr["bytes_per_launch"] / (line["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
                line["roofline"]["frac_hbm_counters"] = line["roofline"]["hbm_counter_GBps"] / HBM_PEAK_GBPS
            else:
                line["roofline"]["traffic_note"] = why
        if solo and not args.no_smoother:
            sm = {"reference_size": guarded(smoother_reference_size, pkg, datagen),
                  "kernel_roofline_in_smoother": guarded(smoother_kernel_in_smoother, args),
                  "kernel_roofline": guarded(smoother_kernel_roofline, pkg),
                  "sweep_kernel_roofline": guarded(smoother_sweep_roofline, pkg)}
            # the smoothers take the filter's storage option where it applies to them (symmetric covariance storage: nLin = 515)
            sm_storage = args.storage if args.storage in ("fp64", "fp64sym") and args.m == 512 else "fp64"
            if not args.no_smoother_full:
                # THE metric's smoother number (`smoother_wall_clock_s`) is the library's default configuration of
                # particleSmootherInformationForm at the per-GPU share: covariances rewritten every third step, ancestor-weight factors
                # carried along the lineages and refactorised every 32nd step (indices identical, ancestor probabilities within 2e-9,
                # outputs within 1e-9 of the from-scratch factorisation: tests/test_gpu_r05_parity.py).  The reference's own arithmetic --
                # chol(Imat_i + ImatAddt) for every particle at every step, :228 -- stays selectable (chol_refresh = 1) and is timed beside it.
                sm["share_full"] = guarded(lambda: smoother_share_full(pkg, datagen, 8192, 3000, 512, 2, args.seed, lazy_depth=3, storage=sm_storage))
                sm["share_full_fresh_factorisation"] = guarded(lambda: smoother_share_full(pkg, datagen, 8192, 3000, 512, 2, args.seed, lazy_depth=3,
                                                                                           chol_refresh=1, storage=sm_storage))
                line["smoother_metric_rule"] = ("smoother_wall_clock_s = complete T = 3000, N_K = 2 run of particleSmootherInformationForm at the per-GPU share "
                                                "N_P = 8192 with the library's defaults (carried ancestor-weight factors); *_fresh_factorisation_s = the same run "
                                                "with the reference's from-scratch factorisation at every step (chol_refresh = 1); smoother_wall_clock_N65536_1gpu_s = "
                                                "the metric's FULL N_P = 65536 on this one GPU (factors never refactorised, no information matrix stored, one "
                                                "covariance bank in place)")
                if "seconds" in sm["share_full"]:
                    line["smoother_wall_clock_s"] = sm["share_full"]["seconds"]
                    line["smoother_wall_clock_workload"] = sm["share_full"]["workload"] + ", lazy_depth 3"
                if "seconds" in sm["share_full_fresh_factorisation"]:
                    line["smoother_wall_clock_fresh_factorisation_s"] = sm["share_full_fresh_factorisation"]["seconds"]
            if not args.no_large:
                sm["radio_N65536"] = guarded(smoother_radio_large, pkg, datagen)
            line["smoother"] = sm
        if solo and not args.no_large:
            def extra(N, m, Kx, Wx, lazy, storage):
                r, *_ = filter_leg(pkg, datagen, N, m, 3000, Kx, Wx, args.seed, lazy, 0, storage)
                r["workload"] = workload_string(N, 3000, m, m + 3, storage, lazy, 1, 2.0 * N * bank_bytes_per_particle(m + 3, storage) > 0.85 * 288e9)
                return r
            # BASELINE.json configs[1].  Full-square storage: at N = 8192, nLin = 259 a step is bound by the fixed per-workgroup work (16 rounds of
            # workgroups, ~58 us each), not by bytes -- symmetric storage (supported at this size too) gives the same 8.0-8.2 M/s
            # the headline configuration in ONE covariance bank (78 GB instead of 156 GB): the shared flush in place (r05), bit-identical results
            def single_bank():
                r, *_ = filter_leg(pkg, datagen, N_local, args.m, T, 40, 8, args.seed, args.lazy_depth, 1, args.storage)
                r["workload"] = workload_string(N_local, T, args.m, args.m + 3, args.storage, args.lazy_depth, 1, True)
                return r
            if args.inplace <= 0 and args.storage in ("fp64sym", "fp32sym"):
                line["headline_single_bank_filter"] = guarded(single_bank)
            line["configs1_filter"] = guarded(extra, 8192, 256, 600, 30, 3, "fp64")
            # 1/8 of BASELINE.json configs[4] (fp32 storage is the config's own dtype): block-lower fp32 tiles at sixteen tile rows (r05), and the
            # full-square fp32 storage it replaces
            line["configs4_share_filter"] = guarded(extra, 32768, 1024, 32, 4, 4, "fp32sym")
            line["configs4_share_filter_full_square"] = guarded(extra, 32768, 1024, 30, 4, 2, "fp32")
        if solo and not args.no_filter_full:
            line["filter_full_T"] = guarded(lambda: filter_full_run(pkg, datagen, N_local, args.m, T, args.seed, args.lazy_depth, args.inplace, args.storage))
            if "seconds" in line["filter_full_T"]:
                line["filter_full_T_s"] = line["filter_full_T"]["seconds"]
        if solo and not args.no_smoother and not args.no_smoother_full and args.smoother_32k:
            sm_storage = args.storage if args.storage in ("fp64", "fp64sym") and args.m == 512 else "fp64"
            r32 = guarded(lambda: smoother_share_full(pkg, datagen, 32768, 3000, 512, 2, args.seed, lazy_depth=3, storage=sm_storage))
            line["smoother"]["N32768_library_defaults"] = r32
            if "seconds" in r32:
                line["smoother_wall_clock_N32768_defaults_s"] = r32["seconds"]
        if solo and not args.no_smoother and not args.no_smoother_full and not args.no_smoother_largest:
            # THE METRIC'S FULL SMOOTHER ON ONE GPU: N_P = 65 536, T = 3000, m = 512, N_K = 2.  What makes it fit (r05): the carried factors need
            # no information matrix once they are never refactorised (chol_refresh >= N_T: 2999 sweeps on end; ancestor probabilities 1.5e-9
            # from the from-scratch arithmetic at N_P = 8192 over T = 3000, every index identical: profiles/r05_refresh_free_vs_fresh_*.jsonl),
            # and one covariance bank rewritten in place.  Per particle 1.19 MB covariance + 2 x 1.21 MB factors + 0.1 MB of vectors = 3.7 MB ->
            # 243 GB + 11 GB of state history + 14 GB of chunk workspaces.  The longest leg of the run (about 3.5 minutes), so it goes last and
            # only while the run is inside its time budget.
            elapsed = time.perf_counter() - t_start
            if elapsed > args.time_budget:
                line["smoother"]["full_N65536_single_gpu"] = {"skipped": f"{elapsed:.0f} s of wall clock used before this leg (--time-budget {args.time_budget:.0f})"}
            else:
                sm_storage = args.storage if args.storage in ("fp64", "fp64sym") and args.m == 512 else "fp64"
                r64 = guarded(lambda: smoother_share_full(pkg, datagen, 65536, 3000, 512, 2, args.seed, lazy_depth=3, storage=sm_storage,
                                                          chol_refresh=3000, inplace=1))
                line["smoother"]["full_N65536_single_gpu"] = r64
                if "seconds" in r64:
                    line["smoother_wall_clock_N65536_1gpu_s"] = r64["seconds"]
                    line["smoother_wall_clock_N65536_1gpu_workload"] = r64["workload"] + ", lazy_depth 3"
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(pkg, data, model, x0_lin, P0, R, args.m)
            except Exception as exc:                                  # report, never hide
                line["cpu_baseline"] = {"value": None, "unit": "particle-steps/s", "cores": None, "kind": "port",
                                        "sample": f"failed: {exc}"}
    else:
        line = None
    if sharded and not args.no_smoother:
        # The smoother as the N-GPU job runs it.  Every rank takes part (collectives), so the leg runs under a watchdog: if it
        # does not come back in time, rank 0 prints the line it already has (the filter measurement) and every rank leaves.
        import threading
        done = {}                          # legs that came back, in order

        def give_up(name):
            # A leg that does not return (a hung collective, or ranks deadlocked because one rank's leg failed while the others
            # wait in RCCL) must not look like success: rank 0 prints what it has, names the leg, and every rank exits NON-ZERO.
            if rank == 0:
                line.update(done)
                line["smoother_sharded_timeout"] = {"leg": name, "error": f"no result within {args.smoother_timeout} s"}
                emit(line)
            sys.stdout.flush()
            os._exit(3)

        def timed(name, fn):
            # one watchdog per leg (four legs share no budget: a slow but healthy run is not cut off by its predecessors)
            dog = threading.Timer(args.smoother_timeout, give_up, args=(name,))
            dog.daemon = True
            dog.start()
            try:
                r = fn()
            except Exception as exc:
                r = {"error": f"{type(exc).__name__}: {exc}"}
            dog.cancel()
            # a leg that failed on one rank only would leave the others waiting in the next collective: agree on it
            bad = torch.tensor([1 if "error" in r else 0], device=f"cuda:{local_rank}")
            if dist is not None:
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()) and "error" not in r:
                r = {"error": "another rank failed in this leg", "partial": r}
            done[name] = r
            return r

        def leg(chol_refresh):
            return smoother_sharded_leg(pkg, mg, datagen, torch, dist, args.smoother_particles, args.m, min(args.smoother_steps, T), T, 2,
                                        args.seed, rank, world, min(args.lazy_depth, 3), chol_refresh)

        def radio(**o):
            return smoother_sharded_radio_leg(pkg, mg, datagen, torch, dist, args.smoother_particles, args.seed, rank, world, **o)
        # the library's defaults (factors carried along the lineages; tolerance: DESIGN.md 4), then the reference's arithmetic (a
        # from-scratch factorisation per particle and step); BASELINE.json configs[3]: dense-radio, 8192 particles per GPU, both ways
        res = timed("smoother_sharded", lambda: leg(0))
        res_f = timed("smoother_sharded_fresh_factorisation", lambda: leg(1))
        res_r = timed("smoother_sharded_radio", lambda: radio(lazy_depth=3, chol_refresh=16))
        res_rf = timed("smoother_sharded_radio_fresh_factorisation", lambda: radio(chol_refresh=1))
        if rank == 0:
            line["smoother_sharded"] = res
            line["smoother_sharded_fresh_factorisation"] = res_f
            line["smoother_sharded_radio"] = res_r
            line["smoother_sharded_radio_fresh_factorisation"] = res_rf
            if "extrapolated_full_T_seconds" in res:
                line["smoother_wall_clock_extrapolated_s"] = res["extrapolated_full_T_seconds"]
            if "extrapolated_full_T_seconds" in res_f:
                line["smoother_wall_clock_fresh_factorisation_extrapolated_s"] = res_f["extrapolated_full_T_seconds"]
    if rank == 0:
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
