"""GPU edge cases the reference's semantics define: chol jitter retry / failure, per-particle x0_lin,
time-varying Q and dt, N_T = 1, Philox <-> replay equivalence, forced resampling ties, and
size-independent properties at the benchmark size."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def run_filter(rbpf, c, **kw):
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    return rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"],
                               kw.get("x0_lin", c["x0_lin"]), kw.get("P0_lin", c["P0_lin"]), kw.get("Q", c["Q"]),
                               kw.get("R", c["R"]), c["N_P"], kw.get("dt", c["dt"]),
                               rng=kw.get("rng", cases.device_rng(rbpf, c)), extras=True)


def test_jitter_retry_path_matches_oracle(rbpf, oracle):
    """S = H P H' + R not positive definite on the first try -> chol(S + 1e-3 I) (particleFilter.m:145-148);
    the downdate still uses the un-jittered S (:198)."""
    c = cases.radio_case(6, 5, 16, seed=12)
    c = dict(c, P0_lin=c["P0_lin"] * 1e-9, R=np.array([[-2e-4]]))
    ref = cases.oracle_filter(c)
    out = run_filter(rbpf, c, P0_lin=c["P0_lin"], R=c["R"])
    ex = out[8]
    np.testing.assert_array_equal(ex["ai"][1:], ref["trace"]["ai"][1:])
    assert rel(ex["w"], ref["trace"]["w"]) <= RTOL
    assert rel(ex["P"], ref["trace"]["P"]) <= 1e-8
    assert rel(ex["xl"], ref["trace"]["xl"]) <= 1e-8


def test_second_cholesky_failure_is_an_error(rbpf):
    c = cases.radio_case(6, 4, 16, seed=12)
    with pytest.raises(rbpf.RBPFError) as ei:
        run_filter(rbpf, c, P0_lin=c["P0_lin"] * 1e-9, R=np.array([[-1.0]]))
    assert ei.value.status == rbpf.RBPF_ERR_CHOL_FAILED       # MATLAB: "Matrix must be positive definite"


def test_per_particle_x0_lin_and_time_varying_q_dt(rbpf, oracle):
    c = cases.mag_case(7, 6, 16, seed=13)
    rs = np.random.RandomState(0)
    x0 = rs.standard_normal((c["model"].nLin, c["N_P"])) * 0.1          # n x N_P  (particleFilter.m:60-61)
    Qs = np.stack([c["Q"] * (1.0 + 0.3 * t) for t in range(c["y"].shape[0] - 1)], axis=2)
    dts = 0.01 * (1.0 + 0.1 * np.arange(c["y"].shape[0] - 1))
    c2 = dict(c, x0_lin=x0, Q=Qs, dt=dts)
    ref = cases.oracle_filter(c2)
    out = run_filter(rbpf, c2, x0_lin=x0, Q=Qs, dt=dts)
    ex = out[8]
    np.testing.assert_array_equal(ex["ai"][1:], ref["trace"]["ai"][1:])
    assert rel(ex["w"], ref["trace"]["w"]) <= RTOL
    assert rel(out[4], ref["P_max"]) <= RTOL and rel(out[3], ref["xl_mean"]) <= RTOL


def test_single_time_step(rbpf):
    c = cases.radio_case(5, 1, 16, seed=14)
    ref = cases.oracle_filter(c)
    out = run_filter(rbpf, c)
    assert rel(out[8]["w"], ref["trace"]["w"]) <= RTOL
    assert rel(out[4], ref["P_max"]) <= RTOL
    assert out[7].shape == (3, 5, 1)


def test_philox_run_equals_its_own_replay(rbpf):
    """The device generator's numbers, dumped by rbpf_philox_fill, fed back in replay mode give the same run."""
    c = cases.mag_case(16, 7, 130, seed=15)
    prng = rbpf.PhiloxRNG(seed=20240611)
    a = run_filter(rbpf, c, rng=prng)
    rep = prng.replay(c["N_P"], c["y"].shape[0], 6)
    assert np.all((rep.U > 0) & (rep.U < 1)) and abs(rep.Z.mean()) < 0.2 and 0.7 < rep.Z.std() < 1.3
    b = run_filter(rbpf, c, rng=rep)
    np.testing.assert_array_equal(a[8]["ai"], b[8]["ai"])
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[4], b[4])
    # and the oracle agrees on that replayed stream
    c2 = dict(c, rng=__import__("rbpf_oracle").ReplayRNG(rep.U, rep.Z, rep.Ufin))
    ref = cases.oracle_filter(c2)
    np.testing.assert_array_equal(a[8]["ai"][1:], ref["trace"]["ai"][1:])
    assert rel(a[4], ref["P_max"]) <= RTOL


def test_forced_ties_on_cumsum_edges(rbpf, oracle):
    """Uniforms placed exactly on / one ulp around values of cumsum(w): strict '<' of tools/sample.m:32."""
    c = cases.radio_case(8, 4, 16, seed=16)
    ref = cases.oracle_filter(c)
    U = c["rng"].U.copy()
    wc = np.cumsum(ref["trace"]["w"][0])
    U[0, 0, :3] = [wc[2], np.nextafter(wc[2], 2.0), np.nextafter(wc[2], -1.0)]
    c2 = dict(c, rng=oracle.ReplayRNG(U, c["rng"].Z, c["rng"].Ufin))
    ref2 = cases.oracle_filter(c2)
    out = run_filter(rbpf, c2, rng=rbpf.ReplayRNG(U, c["rng"].Z, c["rng"].Ufin))
    np.testing.assert_array_equal(out[8]["ai"][1:], ref2["trace"]["ai"][1:])
    assert list(ref2["trace"]["ai"][1][:3]) == [2, 3, 2]


def test_properties_at_benchmark_size(rbpf):
    """BASELINE.json configs[1] sizes (N=8192, m=256; a short T): size-independent invariants --
    weights normalised, ancestors in range, the running Philox stream reproducible, covariance of the
    maximum-weight particle symmetric to rounding and shrinking (P0 - P_max positive semi-definite)."""
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    T, N, m = 12, 8192, 256
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1, m_sim=400)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)

    def go():
        with rbpf.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01,
                                rng=rbpf.PhiloxRNG(7), keep_history=False) as s:
            s.advance(T)
            s.sync()
            return s.finish(want=("traj_max", "traj_mean", "xl_max", "P_max", "xl_mean"))
    a, b = go(), go()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])                 # deterministic reductions, fixed order
    P = a["P_max"]
    assert np.all(np.isfinite(P)) and np.all(np.isfinite(a["traj_mean"]))
    assert np.max(np.abs(P - P.T)) <= 1e-9 * np.max(np.abs(P))
    ev = np.linalg.eigvalsh(0.5 * (P + P.T))
    assert ev.min() > -1e-8 * ev.max()
    ev2 = np.linalg.eigvalsh(P0 - 0.5 * (P + P.T))
    assert ev2.min() > -1e-8 * np.max(np.abs(np.diag(P0)))
    q = a["traj_mean"][3:7]
    assert np.all(np.abs(np.linalg.norm(q, axis=0) - 1.0) < 0.05)  # linear quaternion mean (quirk Q8) stays near unit


@pytest.mark.parametrize("kind", ["radio", "mag"])
def test_large_particle_count_uses_the_multi_workgroup_resample_pipeline(rbpf, kind):
    """N_P > 8192 switches normalisation / resampling to the multi-workgroup pipeline (rbpf_resample.hip);
    indices must still equal the strict-cumsum semantics of tools/sample.m bit for bit."""
    N = 9000
    c = cases.radio_case(N, 4, 16, seed=21) if kind == "radio" else cases.mag_case(N, 3, 16, seed=22)
    ref = cases.oracle_filter(c)
    out = run_filter(rbpf, c)
    ex = out[8]
    np.testing.assert_array_equal(ex["ai"][1:], ref["trace"]["ai"][1:])
    assert ex["iw_max"] == ref["iw_max"]
    assert rel(ex["w"], ref["trace"]["w"]) <= RTOL
    assert rel(out[1], ref["traj_mean"]) <= RTOL
    assert rel(out[4], ref["P_max"]) <= RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("lazy_depth", [0, 3])
def test_device_kalman_recursion_equals_batch_gp_posterior(rbpf, oracle, lazy_depth):
    """Known-answer test that does not go through the oracle's filter: with (numerically) no process noise every
    particle follows the odometry exactly, so the T per-step Kalman updates of the step kernel (particleFilter.m:184-198)
    along that fixed path must equal the batch reduced-rank GP posterior of tools/gp_scalar_potential_fast.m:190-192,
        mean = L'\\(L\\(Phi'*y)),  cov = sigma2 * inv(Phi'*Phi + diag(sigma2./k)),  L = chol(Phi'*Phi + diag(sigma2./k)),
    with Phi the stacked measurement Jacobians.  Benchmark basis size (m = 256, nLin = 259), 150 steps."""
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    T, m, N = 150, 256, 4
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=2, m_sim=400)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    Qtiny = 1e-24 * np.eye(6)
    rng = rbpf.ReplayRNG(np.full((1, T - 1, N), 0.5), np.zeros((1, T - 1, N, 6)))
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, Qtiny, R, N, 0.01, rng=rng,
                              extras=True, lazy_depth=lazy_depth)
    xn_traj, ex = out[7], out[8]
    path = xn_traj[:, 0, :]                                                     # every particle is on the same path
    assert np.max(np.abs(xn_traj - path[:, None, :])) == 0.0
    omdl = oracle.DenseMagModel(NN=mdl.NN.astype(np.int64), L=mdl.L)
    Phi = omdl.measModel(path).reshape(-1, mdl.nLin)                            # [3T x nLin]
    sigma2, k = R[0, 0], np.diag(P0)
    Lc = np.linalg.cholesky(Phi.T @ Phi + np.diag(sigma2 / k))
    mean = np.linalg.solve(Lc.T, np.linalg.solve(Lc, Phi.T @ d["y"].reshape(-1)))
    cov = sigma2 * np.linalg.inv(Lc @ Lc.T)
    for i in range(N):
        np.testing.assert_allclose(ex["xl"][:, i], mean, rtol=1e-7, atol=1e-8 * np.max(np.abs(mean)))
        np.testing.assert_allclose(ex["P"][:, :, i], cov, rtol=1e-6, atol=1e-9 * np.max(np.abs(cov)))
    # ... and the importance weights (particleFilter.m:139-150, constants included): along a fixed path the unnormalised
    # log-weights are the one-step predictive log-densities, so their sum is the log marginal likelihood of all 3 T
    # measurements under the prior,  log N(y; Phi x0, Phi P0 Phi' + kron(I, R))
    yv = d["y"].reshape(-1)
    C = Phi @ P0 @ Phi.T + np.kron(np.eye(T), R)
    Lm = np.linalg.cholesky(C)
    v = np.linalg.solve(Lm, yv - Phi @ np.asarray(x0).reshape(-1))
    loglik = -0.5 * v @ v - np.sum(np.log(np.diag(Lm))) - 0.5 * yv.size * np.log(2.0 * np.pi)
    logw = np.asarray(ex["logw"])
    logw = logw if logw.shape[0] == N else logw.T
    for i in range(N):
        assert abs(float(np.sum(logw[i])) - loglik) <= 1e-8 * abs(loglik), (float(np.sum(logw[i])), loglik)


@pytest.mark.gpu
@pytest.mark.parametrize("lazy_depth", [0, 3])
def test_fix_p_mean_accumulates_over_particles(rbpf, oracle, lazy_depth):
    """rbpf_options.fix_p_mean = 1 (off by default): P_mean = sum_i w(i)*(P_i + (xl_mean - xl_i)(xl_mean - xl_i)'), the
    evident intent of particleFilter.m:228-230, instead of the reference's overwrite (quirk Q3, reproduced by default)."""
    c = cases.mag_case(40, 9, 130, seed=21)
    ref = oracle.particleFilter(c["model"], c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"], c["Q"], c["R"],
                                c["N_P"], c["dt"], c["rng"], fix_p_mean=True)
    q3 = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"],
                              c["dt"], rng=cases.device_rng(rbpf, c), lazy_depth=lazy_depth, fix_p_mean=True)
    scale = np.max(np.abs(ref["P_mean"]))
    assert np.max(np.abs(out[5] - ref["P_mean"])) <= 1e-9 * scale
    assert np.max(np.abs(ref["P_mean"] - q3["P_mean"])) > 1e-3 * scale          # and it is not the quirk's value
