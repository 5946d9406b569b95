#!/usr/bin/env python3
"""Headline benchmark: forward Rao-Blackwellized particle filter throughput on synthetic
slam-dense-mag trajectories (BASELINE.json configs[1]: N=8192, T=3000, m=256, fp64, filter only).

  python bench.py --gpus N --steps K --warmup W

A "step" is one time step of particleFilter (src/particleFilter.m:100-218) over all particles:
resample-gather, dynModel, measModel, importance weights, normalisation and the Kalman map update.
`value` = particle-steps/s over the K timed steps with all inputs resident in HBM.  One JSON line is
printed by rank 0.  See DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

Q_MAG_DIAG = None


def q_mag():
    import numpy as np
    # examples/slam-dense-mag/main.m:22
    return np.diag(np.concatenate((10 ** 2 * np.array([0.05 ** 2, 0.05 ** 2, 0.01 ** 2]),
                                   (np.array([0.01, 0.01, 0.3]) * np.pi / 180) ** 2)))


THETA_MAG = [650.0, 1.2, 200.0, 10.0]                       # examples/slam-dense-mag/main.m:23


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


def cpu_baseline(pkg, data, model, x0_lin, P0, R, m, target_s=15.0):
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
    t_cal = run(5)
    per_step = max(t_cal / 5.0, 1e-6)
    T_s = int(min(1200, max(8, target_s / per_step)))
    secs = run(T_s)
    return {"value": N_s * T_s / secs, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"slam-dense-mag N={N_s} T={T_s} m={m} fp64, C restatement of particleFilter.m "
                      f"(gcc {flags}, OpenMP over particles), {secs:.1f} s"}


def smoother_wallclock(pkg, datagen):
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


def smoother_large(pkg, datagen):
    """Information-form smoother at the large configurations (BASELINE.json configs[2], configs[3]) on one GPU:
    dense-radio N=65536 (T=48, m=128, N_K=3) complete, and 24- and 72-step runs of dense-mag N=8192 (the per-GPU share of
    N=65536 at 8 GPUs), m=512, N_K=2 (a full T=3000 pass takes 3000 such steps per iteration)."""
    import numpy as np
    out = {"unit": "s"}
    T = 48
    Qr = datagen.radio_Q(T, "square_3D")
    th = [0.25, 2.0, 0.01]                                                       # examples/slam-dense-radio/main.m:24
    d = datagen.planar_heading(T, Qr, th, 1.0, seed=1, nLL=4, traj="square_3D")
    mdl, x0, P0, R = pkg.dense_radio_prior(128, d["LL"], th)
    runs = []
    for _ in range(2):                                                           # the first run also pays for the first touch of 30 GB
        t0 = time.perf_counter()
        XNK, _, _ = pkg.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],
                                                        x0, P0, Qr, R, 65536, 3, 1.0, rng=pkg.PhiloxRNG(3))
        runs.append(round(time.perf_counter() - t0, 3))
    out["dense_radio_N65536_T48_m128_NK3"] = min(runs)
    out["dense_radio_N65536_T48_m128_NK3_runs"] = runs
    out["dense_radio_finite"] = bool(np.all(np.isfinite(XNK)))
    Q = q_mag()
    secs = {}
    for T in (4, 24, 24, 72, 72):                                                # the T=4 run only warms the allocator up; best of two:
        d = datagen.bean_6D(T, Q, THETA_MAG, 0.01, seed=1)                       # creating / first-touching the 87 GB of banks
        mdl, x0, P0, R = pkg.dense_mag_prior(512, d["LL"], THETA_MAG)            # takes anything between 0 and 6 s
        t0 = time.perf_counter()
        pkg.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, Q,
                                            R, 8192, 2, 0.01, rng=pkg.PhiloxRNG(3))
        secs[T] = min(secs.get(T, 1e30), time.perf_counter() - t0)
    out["dense_mag_N8192_T24_m512_NK2"] = round(secs[24], 3)
    out["dense_mag_N8192_T72_m512_NK2"] = round(secs[72], 3)
    out["note"] = ("dense-mag N=8192 is the per-GPU share of N=65536 at 8 GPUs; the run times include creating the 87 GB of "
                   "particle banks and the T-long histories (wall clock, best of two); kernel times from rocprofv3: 6.5 ms (plain step) + 23 ms (step with ancestor sampling) per time step, "
                   "profiles/r01x_smoother_mag_N8192_m512_chol64_summary.txt")
    return out


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
    return {"kernel": "chol_solve64_kernel", "workload": f"{B} matrices, n={M}, fp64", "bound": "mfma", "achieved": ach, "peak": 78.6,
            "unit": "TFLOP/s", "frac": ach / 78.6, "avg_launch_ms": ms, "algorithmic_flop_per_launch": flops, "traffic": None,
            "finite": bool(np.all(np.isfinite(logw))) and status == 0,
            "in_smoother": "16.6 ms per launch of 8192 (28.6 % of the peak; 77.6 GB of HBM traffic = 4.7 TB/s) with the Imat gather "
                           "folded in, profiles/r01x_smoother_mag_N8192_m512_chol64_summary.txt, profiles/r01x_chol64_counters_summary.txt"}


def config2_filter(pkg, datagen, args):
    """BASELINE.json configs[2], filter part, on this one GPU: N=65536, m=512 (nLin=515), fp64 -- a single covariance
    bank of 139 GB rewritten in place (rbpf_options.inplace, automatic).  45 timed steps after 6 warm-up steps."""
    Q = q_mag()
    T, K, W = 3000, 45, 6
    data = datagen.bean_6D(T, Q, THETA_MAG, 0.01, seed=args.seed)
    model, x0_lin, P0, R = pkg.dense_mag_prior(512, data["LL"], THETA_MAG)
    with pkg.FilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R, 65536, 0.01,
                           rng=pkg.PhiloxRNG(args.seed), keep_history=False, lazy_depth=3, inplace=0) as sess:
        sess.advance(W)
        sess.sync()
        sess.timing(enable=True)
        t0 = time.perf_counter()
        sess.advance(K)
        sess.sync()
        dt_s = time.perf_counter() - t0
        tm = sess.timing(reset=True)
    avg_ms = tm["ms"] / max(tm["launches"], 1)
    ach = tm["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
    return {"workload": "slam-dense-mag N=65536 T=3000 m=512 (nLin=515) fp64 filter, 1 GPU, single bank in place",
            "value": 65536 * K / dt_s, "unit": "particle-steps/s", "steps": K, "warmup": W, "ms_per_step": dt_s / K * 1e3,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": tm["bytes_per_launch"]},
            "full_run": "profiles/r01e_filter_N65536_m512_T3000_bench.json (2980 timed steps: 2.05 M/s, 95 s)"}


def config4_share_filter(pkg, datagen, args):
    """Per-GPU share of BASELINE.json configs[4] (N=262144, m=1024, fp32, 8 GPUs): N=32768 particles, nLin=1027, covariance
    banks STORED in fp32 (arithmetic fp64), one 138 GB bank rewritten in place, lazy_depth 2.  30 timed steps."""
    Q = q_mag()
    T, K, W = 3000, 30, 4
    data = datagen.bean_6D(T, Q, THETA_MAG, 0.01, seed=args.seed)
    model, x0_lin, P0, R = pkg.dense_mag_prior(1024, data["LL"], THETA_MAG)
    with pkg.FilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R, 32768, 0.01,
                           rng=pkg.PhiloxRNG(args.seed), keep_history=False, lazy_depth=2, inplace=0, storage="fp32") as sess:
        sess.advance(W)
        sess.sync()
        sess.timing(enable=True)
        t0 = time.perf_counter()
        sess.advance(K)
        sess.sync()
        dt_s = time.perf_counter() - t0
        tm = sess.timing(reset=True)
    avg_ms = tm["ms"] / max(tm["launches"], 1)
    ach = tm["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
    return {"workload": "slam-dense-mag N=32768 (1/8 of N=262144) T=3000 m=1024 (nLin=1027), fp32 covariance storage / fp64 arithmetic, "
                        "filter, 1 GPU, single bank in place, lazy_depth 2",
            "value": 32768 * K / dt_s, "unit": "particle-steps/s", "steps": K, "warmup": W, "ms_per_step": dt_s / K * 1e3,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": tm["bytes_per_launch"]},
            "parity": "fp32 storage agrees with the fp64 oracle to 2e-5 over short runs (tests/test_gpu_filter.py), not to 1e-9"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=8192, help="particles per GPU")
    ap.add_argument("--m", type=int, default=256)
    ap.add_argument("--T", type=int, default=3000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-smoother", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the N=65536 single-GPU legs (configs[2] filter, large smoothers)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lazy-depth", type=int, default=3, help="rewrite the covariances every C-th step only (0/1: every step)")
    ap.add_argument("--inplace", type=int, default=0, help="single covariance bank rewritten in place: 1 on, -1 off, 0 automatic (when two banks do not fit)")
    ap.add_argument("--storage", default="fp64", choices=["fp64", "fp32"], help="precision the covariance banks are STORED in (arithmetic is fp64)")
    ap.add_argument("--force-sharded", action="store_true", help="use the sharded session even at --gpus 1 (testing)")
    args = ap.parse_args()

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
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    datagen = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    K, W, T = args.steps, args.warmup, args.T
    if K + W > T:
        raise SystemExit(f"--steps + --warmup must be <= T={T}")
    Q = q_mag()
    data = datagen.bean_6D(T, Q, THETA_MAG, 0.01, seed=args.seed)           # data seed 1
    model, x0_lin, P0, R = pkg.dense_mag_prior(args.m, data["LL"], THETA_MAG)
    N_local = args.particles

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1 or args.force_sharded:
        mg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.multigpu")
        sess = mg.ShardedFilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R,
                                       N_local, 0.01, rng=pkg.PhiloxRNG(args.seed), rank=rank, world=world,
                                       lazy_depth=args.lazy_depth, storage=args.storage)
    else:
        sess = pkg.FilterSession(model, data["dx"], data["y"], data["initState"], x0_lin, P0, Q, R, N_local, 0.01,
                                 rng=pkg.PhiloxRNG(args.seed), keep_history=False, lazy_depth=args.lazy_depth,
                                 inplace=args.inplace, storage=args.storage)     # filter seed 1
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
    if dist is not None:
        tt = torch.tensor([dt_s], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_s = float(tt.item())
    chk = sess.finish(want=("traj_mean",))
    if not np.all(np.isfinite(chk["traj_mean"][:, :W + K])):
        raise SystemExit("non-finite filter output")
    shard_stats = getattr(sess, "stats", None)
    sess.close()

    if rank == 0:
        n = model.nLin
        N_total = N_local * world
        value = N_total * K / dt_s
        avg_ms = tm["ms"] / max(tm["launches"], 1)
        achieved = tm["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
        traffic = None
        tr_file = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tr_file):
            try:
                rec = json.load(open(tr_file))
                if rec.get("N_P") == N_local and rec.get("m") == args.m and rec.get("lazy_depth", 0) == args.lazy_depth:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "particle-steps/s (filter)", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt_s / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.storage == "fp64" else "f64 arithmetic, f32 covariance storage", "data": "synthetic",
            "config": {"workload": f"slam-dense-mag N={N_total} T={T} m={args.m} (nLin={n}) fp64 filter only "
                                   f"(BASELINE.json configs[1] x {world} GPU)",
                       "particles_per_gpu": N_local, "rng": "device Philox4x32-10", "data_seed": args.seed,
                       "lazy_depth": args.lazy_depth, "inplace": args.inplace, "storage": args.storage,
                       "filter_seed": args.seed},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "step_kernel", "avg_launch_ms": avg_ms, "launches": tm["launches"],
                         "algorithmic_bytes_per_launch": tm["bytes_per_launch"],
                         "note": ("algorithmic bytes = the reference's one read + one write of every covariance per step "
                                  "(SURVEY 8d); with lazy_depth C the stored covariances are rewritten every C-th step only, so "
                                  "the measured HBM traffic is below the algorithmic figure and frac can exceed 1"
                                  if args.lazy_depth >= 2 else "")},
        }
        if shard_stats:
            st = dict(shard_stats)
            ph = st.pop("phase_s", {})
            st["phase_ms_per_step"] = {k: round(v / max(st.get("steps", 1), 1) * 1e3, 4) for k, v in ph.items()}
            line["config"]["sharding"] = st
        if world == 1 and not args.no_smoother and not args.force_sharded:
            try:
                line["smoother"] = smoother_wallclock(pkg, datagen)
            except Exception as exc:
                line["smoother"] = {"error": str(exc)}
            try:
                line["smoother"]["kernel_roofline"] = smoother_kernel_roofline(pkg)
            except Exception as exc:
                line["smoother"]["kernel_roofline"] = {"error": str(exc)}
            if not args.no_large:
                try:
                    line["smoother"]["large"] = smoother_large(pkg, datagen)
                except Exception as exc:
                    line["smoother"]["large"] = {"error": str(exc)}
        if world == 1 and not args.no_large and not args.force_sharded and N_local == 8192 and args.m == 256:
            try:
                free_b, _ = torch.cuda.mem_get_info()
                if free_b > 170e9:
                    line["config2_filter"] = config2_filter(pkg, datagen, args)
                else:
                    line["config2_filter"] = {"skipped": f"only {free_b / 1e9:.0f} GB free"}
            except Exception as exc:
                line["config2_filter"] = {"error": str(exc)}
            try:
                free_b, _ = torch.cuda.mem_get_info()
                if free_b > 170e9:
                    line["config4_share_filter"] = config4_share_filter(pkg, datagen, args)
                else:
                    line["config4_share_filter"] = {"skipped": f"only {free_b / 1e9:.0f} GB free"}
            except Exception as exc:
                line["config4_share_filter"] = {"error": str(exc)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(pkg, data, model, x0_lin, P0, R, args.m)
            except Exception as exc:                                  # report, never hide
                line["cpu_baseline"] = {"value": None, "unit": "particle-steps/s", "cores": None, "kind": "port",
                                        "sample": f"failed: {exc}"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
