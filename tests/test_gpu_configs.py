"""GPU tests at the sizes of BASELINE.json's configurations (configs[2], configs[3], configs[4]) -- the sizes everything
else is graded on.  The numpy oracle cannot run there, so these use (a) size-independent properties of the filter
(determinism, normalised weights, ancestors in range, symmetric positive-definite shrinking covariances), (b) equality of
two independent schedules of the same arithmetic (single bank in place vs ping-pong banks; one GPU vs the sharded session),
(c) oracle parity at the full matrix size with few particles, and (d) the plain-C restatement (oracle/rbpf_oracle_c.c) of
the filter and of both smoothers at N ~ 1000, T ~ 100."""
import importlib
import os
import sys

import numpy as np
import pytest

import cases
import oracle_c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def mag_inputs(rbpf, T, m, seed=1):
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=seed)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    return d, mdl, x0, P0, R


def run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, **kw):
    with rbpf.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rbpf.PhiloxRNG(5),
                            keep_history=True, trace=True, **kw) as s:
        s.advance(steps)
        s.sync()
        return s.finish(want=want)


def check_filter_properties(out, N, steps, P0):
    w, ai = out["trace_w"], out["trace_ai"]
    assert np.all(np.isfinite(w)) and np.all(w >= 0)
    np.testing.assert_allclose(w.sum(axis=0), 1.0, rtol=0, atol=1e-12)          # particleFilter.m:154-156
    assert ai[:, 1:].min() >= 0 and ai[:, 1:].max() < N                          # tools/sample.m:32
    assert np.all(np.isfinite(out["traj_mean"][:, :steps])) and np.all(np.isnan(out["traj_mean"][:, steps:]))
    P = out["P_max"]
    assert rel(P, P.T) < 1e-12                                                   # plain form, no symmetrisation: still symmetric
    ev = np.linalg.eigvalsh(0.5 * (P + P.T))
    assert ev.min() > 0                                                          # positive definite
    assert np.all(np.diag(P) <= np.diag(P0) * (1 + 1e-12)) and np.trace(P) < np.trace(P0)   # information only shrinks it


@pytest.mark.parametrize("lazy_depth", [3, 4])
def test_configs2_filter_single_bank_in_place(rbpf, lazy_depth):
    """BASELINE.json configs[2], filter: N = 65 536, m = 512 (nLin = 515), fp64, lazy_depth 3 / 4 (bench.py's default) -- one
    139 GB covariance bank rewritten in place (chosen automatically: two banks do not fit).  Two runs are bit-identical;
    properties hold."""
    N, steps = 65536, 10
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=lazy_depth, inplace=0)
    check_filter_properties(a, N, steps, P0)
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=lazy_depth, inplace=0)
    for k in want:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("lazy_depth", [3, 4])
def test_in_place_equals_ping_pong_at_the_largest_size_with_two_banks(rbpf, lazy_depth):
    """N = 32 768, m = 512: two 69.5 GB banks still fit, so the in-place flush (children moved to dead entries, first
    children overwritten last) can be compared with the ping-pong schedule: every output bit for bit."""
    N, steps = 32768, 10
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean", "trace_logw")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=lazy_depth, inplace=1)
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=lazy_depth, inplace=-1)
    check_filter_properties(a, N, steps, P0)
    for k in want:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_configs4_share_fp32_storage(rbpf):
    """Per-GPU share of BASELINE.json configs[4]: N = 32 768, m = 1024 (nLin = 1027: four row chunks per wave, whole
    columns, the x4 unroll), covariance banks stored in fp32 -- one 138 GB bank in place, lazy_depth 2."""
    N, steps = 32768, 6
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 1024)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=2, inplace=0, storage="fp32")
    check_filter_properties(a, N, steps, P0)
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=2, inplace=0, storage="fp32")
    for k in want:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("storage,lazy_depth,tol", [("fp64", 0, 1e-9), ("fp64", 2, 1e-9), ("fp32", 0, 2e-5), ("fp32", 2, 2e-5)])
def test_m1024_filter_matches_oracle(rbpf, storage, lazy_depth, tol):
    """nLin = 1027 (the RS = 4 / CS = 1 wave decomposition, with the extra unroll for float storage) against the numpy oracle:
    1e-9 with fp64 banks (north_star), 2e-5 with fp32 STORAGE of the covariance banks (arithmetic stays fp64; the stored
    matrix carries 24-bit mantissas, so 1e-9 is out of reach by construction -- stated in the bench line as well)."""
    c = cases.mag_case(6, 5, 1024, seed=29)
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                              rng=cases.device_rng(rbpf, c), extras=True, lazy_depth=lazy_depth, storage=storage)
    ex, tr = out[8], ref["trace"]
    if storage == "fp64":
        np.testing.assert_array_equal(ex["ai"][1:], tr["ai"][1:])
    assert rel(ex["w"], tr["w"]) <= max(tol, RTOL) * (1 if storage == "fp64" else 50)   # weights: exp() of log-weights of size ~1e2
    for got, want_ in ((out[0], ref["traj_max"]), (out[1], ref["traj_mean"]), (out[2], ref["xl_max"]), (out[4], ref["P_max"]),
                       (ex["xl"], tr["xl"]), (ex["P"], tr["P"])):
        if storage == "fp64" or np.array_equal(ex["ai"][1:], tr["ai"][1:]):
            assert rel(got, want_) <= tol


@pytest.mark.parametrize("chol_refresh", [1, 0])
def test_information_form_smoother_matches_oracle_at_m512(rbpf, chol_refresh):
    """particleSmootherInformationForm at the metric's matrix size nLin = 515 (33 row tiles: the 8-wave shape of the
    64-column factorisation, the step kernel's 4-chunk layout) against the numpy oracle, N_K = 3 -- the reference's arithmetic
    (chol_refresh = 1: a factorisation per particle and step) and the library default (0: carried factors)."""
    import test_gpu_smoother as ts
    c = cases.mag_case(6, 5, 512, seed=37, N_K=3)
    ref, out = ts.run_both(rbpf, c, info_form=True, chol_refresh=chol_refresh)
    ts.check(ref, out, 3)


def test_configs3_radio_smoother_one_gpu_equals_sharded_world1(rbpf):
    """BASELINE.json configs[3]: slam-dense-radio, N = 65 536, particleSmootherInformationForm, N_K = 2 -- the single-GPU
    entry point against the sharded session at world 1 (the code path 8 GPUs run, device planner included): bit for bit."""
    mg = importlib.import_module(rbpf.__name__ + ".multigpu")
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T, N_K = 65536, 6, 2
    Qr = dg.radio_Q(T, "square_3D")
    th = [0.25, 2.0, 0.01]
    d = dg.planar_heading(T, Qr, th, 1.0, seed=1, nLL=4, traj="square_3D")
    mdl, x0, P0, R = rbpf.dense_radio_prior(128, d["LL"], th)
    XNK, XLK, PK, ex = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],
                                                            x0, P0, Qr, R, N, N_K, 1.0, rng=rbpf.PhiloxRNG(9), extras=True)
    assert np.all(np.isfinite(XNK)) and np.all(np.isfinite(PK))
    w = ex["w"]
    np.testing.assert_allclose(w.sum(axis=2), 1.0, rtol=0, atol=1e-12)
    assert ex["ai"][:, 1:].min() >= 0 and ex["ai"][:, 1:].max() < N
    pa = ex["paNt"][1, 1:]
    np.testing.assert_allclose(pa.sum(axis=1), 1.0, rtol=0, atol=1e-12)          # particleSmootherInformationForm.m:243-245
    with mg.ShardedSmootherSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, Qr, R, N, N_K, 1.0, rng=rbpf.PhiloxRNG(9), rank=0,
                                   world=1) as s:
        X2, L2, P2 = s.run()
        aks = list(s.aks)
    np.testing.assert_array_equal(X2, XNK)
    np.testing.assert_array_equal(L2, XLK)
    np.testing.assert_array_equal(P2, PK)
    assert aks == [int(a) for a in ex["ak"]]


# ---- the second, independently structured restatement (plain C) at sizes the numpy oracle cannot reach -------------------
@pytest.fixture(scope="module")
def c_radio_case():
    return cases.radio_case(1024, 100, 128, seed=43, N_K=2, traj="square_3D")


@pytest.mark.parametrize("info_form", [True, False])
def test_radio_smoothers_against_the_c_restatement(rbpf, c_radio_case, info_form):
    """slam-dense-radio, N = 1024, T = 100, m = 128, N_K = 2, both forms (the covariance form stacks up to 99 future
    measurements): ancestors and trajectory draws bit-exact; weights, ancestor probabilities and outputs to 1e-9 over 100
    steps of 1024 draws."""
    c = c_radio_case
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    ref, secs = oracle_c.particle_smoother(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], 2, c["dt"],
                                           cases.device_rng(rbpf, c), info_form)
    f = rbpf.particleSmootherInformationForm if info_form else rbpf.particleSmoother
    XNK, XLK, PK, ex = f(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                         c["N_P"], 2, c["dt"], rng=cases.device_rng(rbpf, c), extras=True)
    np.testing.assert_array_equal(ex["ak"], ref["ak"])
    np.testing.assert_array_equal(ex["ai"][:, 1:], ref["ai"][:, 1:])
    assert rel(ex["w"], ref["w"]) <= RTOL
    a, b = ex["paNt"][1, 1:], ref["paNt"][1, 1:]
    assert np.max(np.abs(a - b)) <= RTOL * max(1.0, np.max(np.abs(b)))
    assert rel(XNK, ref["XNK"]) <= RTOL and rel(XLK, ref["XLK"]) <= RTOL and rel(PK, ref["PK"]) <= RTOL


def test_mag_information_form_smoother_against_the_c_restatement(rbpf):
    """slam-dense-mag, N = 1024, T = 100, m = 128 (nLin = 131), N_K = 2, information form, against the C restatement."""
    c = cases.mag_case(1024, 100, 128, seed=47, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    ref, secs = oracle_c.particle_smoother(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], 2, c["dt"],
                                           cases.device_rng(rbpf, c), True)
    XNK, XLK, PK, ex = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"],
                                                            c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], 2, c["dt"],
                                                            rng=cases.device_rng(rbpf, c), extras=True)
    np.testing.assert_array_equal(ex["ak"], ref["ak"])
    np.testing.assert_array_equal(ex["ai"][:, 1:], ref["ai"][:, 1:])
    assert rel(ex["w"], ref["w"]) <= RTOL
    assert rel(XNK, ref["XNK"]) <= RTOL and rel(XLK, ref["XLK"]) <= RTOL and rel(PK, ref["PK"]) <= RTOL


def test_radio_filter_against_the_c_restatement(rbpf):
    """slam-dense-radio filter, N = 4096, T = 200, m = 128 against the C restatement (the numpy oracle stops at a few dozen)."""
    c = cases.radio_case(4096, 200, 128, seed=51, traj="square_3D")
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    ref, secs = oracle_c.particle_filter(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                                         cases.device_rng(rbpf, c), want_full=False)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                              rng=cases.device_rng(rbpf, c), want_xn_traj=False)
    assert rel(out[0], ref["traj_max"]) <= RTOL and rel(out[1], ref["traj_mean"]) <= RTOL
    assert rel(out[2], ref["xl_max"]) <= RTOL and rel(out[3], ref["xl_mean"]) <= RTOL


@pytest.mark.parametrize("kind,N,T,m", [("radio", 65536, 8, 128), ("mag", 8192, 8, 512)])
def test_carried_factors_and_lazy_update_at_the_configuration_sizes(rbpf, kind, N, T, m):
    """The options the bench's second smoother number uses (lazy_depth = 3, chol_refresh) at configs[3]'s size and at the
    per-GPU share of configs[2] (N_P = 8192, nLin = 515) against the reference's arithmetic (chol_refresh = 1) on the same Philox streams: same
    ancestors and trajectory draws, ancestor probabilities within 1e-9, outputs within 1e-9."""
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    if kind == "radio":
        Q = dg.radio_Q(T, "square_3D")
        th = [0.25, 2.0, 0.01]
        d = dg.planar_heading(T, Q, th, 1.0, seed=1, nLL=4, traj="square_3D")
        mdl, x0, P0, R = rbpf.dense_radio_prior(m, d["LL"], th)
        dt = 1.0
    else:
        Q, dt = cases.Q_MAG, 0.01
        d = dg.bean_6D(T, Q, cases.THETA_MAG, dt, seed=1)
        mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    run = lambda **kw: rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],   # noqa: E731
                                                            x0, P0, Q, R, N, 2, dt, rng=rbpf.PhiloxRNG(9), extras=True, **kw)
    a = run(chol_refresh=1)                               # from scratch at every step (0 is automatic since r05)
    b = run(lazy_depth=3, chol_refresh=3)
    np.testing.assert_array_equal(a[3]["ai"][:, 1:], b[3]["ai"][:, 1:])
    np.testing.assert_array_equal(a[3]["ak"], b[3]["ak"])
    pa, pb = a[3]["paNt"][1, 1:], b[3]["paNt"][1, 1:]
    assert np.max(np.abs(pa - pb)) <= 1e-9
    assert rel(b[3]["w"], a[3]["w"]) <= 1e-9
    assert rel(b[0], a[0]) <= RTOL and rel(b[1], a[1]) <= RTOL and rel(b[2], a[2]) <= RTOL


def test_two_ranks_carried_factors_at_the_share_size_equal_one_gpu(rbpf):
    """The N-GPU smoother's code path at a realistic per-rank size: two ranks of 4096 particles (m = 512, nLin = 515) sharing the
    card, lazy covariance update + carried factors (records with factors and base matrices cross ranks), against the single-GPU
    smoother with 8192 particles and the same options on the same Philox streams: the same trajectory draws, outputs to 1e-9."""
    import socket
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sharded_rehearsal as sr
    import bench
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    o = dict(world=2, n_local=4096, m=512, T=36, N_K=2, lazy_depth=3, chol_refresh=16, exchange_capacity=0)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=sr._worker, args=(r, 2, port, o, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r["rank"])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert "error" not in res[0] and "error" not in res[1], (res[0].get("error"), res[1].get("error"))
    Q = bench.q_mag()
    d = dg.bean_6D(o["T"], Q, bench.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(o["m"], d["LL"], bench.THETA_MAG)
    XNK, XLK, PK, ex = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],
                                                            x0, P0, Q, R, 8192, o["N_K"], 0.01, rng=rbpf.PhiloxRNG(1), extras=True,
                                                            lazy_depth=3, chol_refresh=16)
    for r in res:
        assert r["finite"] and r["migrated"] > 0 and r["sent_records"] > 0
        np.testing.assert_array_equal(np.asarray(r["aks"]), ex["ak"])
        np.testing.assert_allclose(r["XNK"], XNK, rtol=1e-9, atol=1e-11)
    assert res[0]["refreshes"] == 1 + (o["T"] - 2) // 16
