"""rbpf_options.chol_refresh = K > 1: the ancestor-weight factors of particleSmootherInformationForm.m:224-236 are carried
along the lineages (ny rank-1 updates + ny rank-1 downdates per step, rbpf_chol_sweep.hpp) and recomputed from the exactly
carried Imat every K-th step, instead of a fresh chol(Imat_i + ImatAddt) per particle and step.  Same algebra, different
arithmetic.  Stated tolerance: ancestor probabilities paNt within 2e-9 (absolute, they are <= 1) of the fresh factorisation's;
the cases against the numpy oracle (which factorises from scratch as the reference does) hold 1e-9; every ancestor index and
every output equal to the oracle's as in the default mode.  K = 1000 never refreshes after the first step: the drift over the whole run stays inside the same bound."""
import numpy as np
import pytest

import cases
import test_gpu_smoother as ts

pytestmark = pytest.mark.gpu


def run(rbpf, c, **kw):
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    return rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0,
                                                c["Q"], R, c["N_P"], c["N_K"], c["dt"], rng=cases.device_rng(rbpf, c), extras=True, **kw)


@pytest.mark.parametrize("K", [2, 5, 1000])
@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 8, 14, 130), ("mag", 6, 12, 512), ("mag", 7, 10, 40), ("radio", 10, 16, 128),
                                            ("radio", 9, 12, 24)])
def test_carried_factors_match_the_oracle(rbpf, kind, N_P, N_T, m, K):
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=71, N_K=3)
    ref = cases.oracle_smoother(c, True)
    out = run(rbpf, c, chol_refresh=K)
    ts.check(ref, out, 3)                                  # ancestors bit-exact, paNt / weights / outputs to 1e-9


@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 64, 60, 130), ("radio", 96, 80, 128)])
def test_drift_of_the_carried_factors_over_a_long_run(rbpf, kind, N_P, N_T, m):
    """60 / 80 steps without a refresh against the fresh factorisation of every step (chol_refresh = 1) on the same random numbers: the
    ancestor probabilities stay within 1e-9, so every index and every output is identical or within 1e-9."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=73, N_K=2)
    a = run(rbpf, c, chol_refresh=1)                      # from scratch at every step (the r04 default; 0 is automatic since r05)
    b = run(rbpf, c, chol_refresh=1000)
    pa, pb = a[3]["paNt"][1, 1:], b[3]["paNt"][1, 1:]
    drift = float(np.max(np.abs(pa - pb)))
    assert drift <= 1e-9, drift
    np.testing.assert_array_equal(a[3]["ai"][:, 1:], b[3]["ai"][:, 1:])
    np.testing.assert_array_equal(a[3]["ak"], b[3]["ak"])
    assert ts.rel(b[0], a[0]) <= 1e-9 and ts.rel(b[2], a[2]) <= 1e-9


def test_carried_factors_with_the_lazy_covariance_update(rbpf):
    c = cases.mag_case(7, 11, 256, seed=75, N_K=3)
    ref = cases.oracle_smoother(c, True)
    out = run(rbpf, c, chol_refresh=4, lazy_depth=3)
    ts.check(ref, out, 3)


def test_refresh_in_chunks_of_factor_workspaces(rbpf):
    """r05: with carried factors the refresh factorises in chunks of 4096 particles over one set of workspaces (the 2.2 MB per
    particle N_P = 32 768 has no room for).  N_P = 4100 = one full chunk + 4: the default (K = 32, refreshes at t = 1 and t = 33)
    against the from-scratch factorisation of every step on the same Philox streams -- same ancestors and draws, paNt 1e-9."""
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    T, N = 36, 4100
    Q = dg.radio_Q(T, "square_3D")
    th = [0.25, 2.0, 0.01]
    d = dg.planar_heading(T, Q, th, 1.0, seed=1, nLL=4, traj="square_3D")
    mdl, x0, P0, R = rbpf.dense_radio_prior(128, d["LL"], th)
    assert rbpf.chol_refresh_in_use(mdl, 0) == 32
    go = lambda **kw: rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],   # noqa: E731
                                                           x0, P0, Q, R, N, 2, 1.0, rng=rbpf.PhiloxRNG(9), extras=True, **kw)
    a = go(chol_refresh=1)
    b = go()
    np.testing.assert_array_equal(a[3]["ai"][:, 1:], b[3]["ai"][:, 1:])
    np.testing.assert_array_equal(a[3]["ak"], b[3]["ak"])
    assert np.max(np.abs(a[3]["paNt"][1, 1:] - b[3]["paNt"][1, 1:])) <= 1e-9
    assert ts.rel(b[0], a[0]) <= 1e-9 and ts.rel(b[1], a[1]) <= 1e-9 and ts.rel(b[2], a[2]) <= 1e-9


@pytest.mark.parametrize("kind,N_P,N_T,m,opts", [("mag", 7, 12, 512, dict(storage="fp64sym", lazy_depth=3, inplace=1)),
                                                  ("mag", 9, 10, 256, dict(lazy_depth=3)), ("radio", 10, 14, 128, dict(lazy_depth=2, inplace=1)),
                                                  ("mag", 8, 11, 130, dict())])
def test_refresh_free_smoother_matches_the_oracle(rbpf, kind, N_P, N_T, m, opts):
    """r05: chol_refresh >= N_T - 1 never refactorises after the first step and stores no information matrix (the first
    factorisation runs over a chunk buffer); with inplace = 1 the information-form smoother keeps ONE covariance bank rewritten in
    place.  Against the numpy oracle like every other mode: ancestors bit-exact, paNt / weights / outputs 1e-9."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=79, N_K=3)
    ref = cases.oracle_smoother(c, True)
    out = run(rbpf, c, chol_refresh=10 ** 6, **opts)
    ts.check(ref, out, 3)


@pytest.mark.parametrize("kind,N_P,N_T,m,K,opts", [("mag", 7, 80, 512, 33, dict(storage="fp64sym", lazy_depth=3, inplace=1)),
                                                    ("mag", 9, 45, 130, 7, dict(lazy_depth=2)), ("radio", 10, 75, 128, 36, dict(lazy_depth=3, inplace=1)),
                                                    ("mag", 6, 12, 256, 2, dict())])
def test_refresh_from_the_origin_matches_the_oracle(rbpf, kind, N_P, N_T, m, K, opts):
    """rbpf_options.info_rebuild = 1: no information matrix is stored; every K-th step rebuilds them from Imat0 along the whole
    ancestral path in segments of 32 generations (paths of 33 .. 79 generations: two and three segments, a one-generation tail) and
    refactorises.  Against the numpy oracle: ancestors bit-exact, paNt / weights / outputs 1e-9."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=83, N_K=2)
    ref = cases.oracle_smoother(c, True)
    out = run(rbpf, c, chol_refresh=K, info_rebuild=1, **opts)
    ts.check(ref, out, 2)


def test_refresh_free_and_in_place_equal_the_default_at_many_chunks(rbpf):
    """N_P = 9000 (three chunks of the first factorisation: 4096 + 4096 + 808), dense-radio m = 128, T = 40: the refresh-free in-place
    run against the library default (K = 32, two banks) on the same Philox streams: same ancestors and draws, outputs 1e-9."""
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    T, N = 40, 9000
    Q = dg.radio_Q(T, "square_3D")
    th = [0.25, 2.0, 0.01]
    d = dg.planar_heading(T, Q, th, 1.0, seed=1, nLL=4, traj="square_3D")
    mdl, x0, P0, R = rbpf.dense_radio_prior(128, d["LL"], th)
    go = lambda **kw: rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],   # noqa: E731
                                                           x0, P0, Q, R, N, 2, 1.0, rng=rbpf.PhiloxRNG(9), extras=True, **kw)
    a = go(lazy_depth=3)
    b = go(lazy_depth=3, chol_refresh=10 ** 6, inplace=1)
    b2 = go(lazy_depth=3, chol_refresh=17, info_rebuild=1, inplace=1)      # refreshes from the origin at t = 18, 35 over three chunks
    for other in (b2,):
        np.testing.assert_array_equal(a[3]["ai"][:, 1:], other[3]["ai"][:, 1:])
        assert np.max(np.abs(a[3]["paNt"][1, 1:] - other[3]["paNt"][1, 1:])) <= 1e-9 and ts.rel(other[0], a[0]) <= 1e-9
    np.testing.assert_array_equal(a[3]["ai"][:, 1:], b[3]["ai"][:, 1:])
    np.testing.assert_array_equal(a[3]["ak"], b[3]["ak"])
    assert np.max(np.abs(a[3]["paNt"][1, 1:] - b[3]["paNt"][1, 1:])) <= 1e-9
    assert ts.rel(b[0], a[0]) <= 1e-9 and ts.rel(b[1], a[1]) <= 1e-9 and ts.rel(b[2], a[2]) <= 1e-9


def test_info_rebuild_is_refused_without_carried_factors(rbpf):
    c = cases.mag_case(5, 5, 130, seed=3, N_K=2)
    with pytest.raises(rbpf.RBPFError) as ei:
        run(rbpf, c, chol_refresh=1, info_rebuild=1)
    assert ei.value.status == rbpf.RBPF_ERR_UNSUPPORTED


def test_in_place_needs_the_lazy_update(rbpf):
    c = cases.mag_case(5, 5, 130, seed=3, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError) as ei:
        rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                                             c["N_P"], 2, c["dt"], rng=cases.device_rng(rbpf, c), inplace=1)      # lazy_depth < 2
    assert ei.value.status == rbpf.RBPF_ERR_UNSUPPORTED


def test_carried_factors_are_refused_where_they_do_not_apply(rbpf):
    c = cases.mag_case(4, 4, 600, seed=1, N_K=2)             # nLin = 603 > 575
    with pytest.raises(rbpf.RBPFError) as ei:
        run(rbpf, c, chol_refresh=8)
    assert ei.value.status == rbpf.RBPF_ERR_UNSUPPORTED


_fresh_T3000 = {}                                            # the from-scratch run, shared by the two parametrisations


@pytest.mark.parametrize("opts", [dict(chol_refresh=32), dict(chol_refresh=10 ** 6, inplace=1)])
def test_drift_at_the_bench_configuration_over_the_full_horizon(rbpf, opts):
    """The options of the metric's smoother number (lazy_depth = 3, chol_refresh = 32 = what 0 resolves to) at the metric's matrix size
    (slam-dense-mag m = 512, nLin = 515) over the metric's T = 3000 time steps -- 94 refresh cycles, device Philox, the product's own data
    generator -- against the fresh factorisation on the same streams: ancestor probabilities within the stated 2e-9 at every
    step (measured 8.8e-10; it does not grow with the refresh period, tools/drift_study.py), every ancestor index and trajectory
    draw identical."""
    import importlib
    import bench
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T, N_K = 128, 3000, 2
    Q = bench.q_mag()
    d = dg.bean_6D(T, Q, bench.THETA_MAG, 0.01, seed=5)
    mdl, x0, P0, R = rbpf.dense_mag_prior(512, d["LL"], bench.THETA_MAG)

    def go(**kw):
        return rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, Q, R,
                                                    N, N_K, 0.01, rng=rbpf.PhiloxRNG(17), extras=True, **kw)
    a = _fresh_T3000.get("a") or go(lazy_depth=3, chol_refresh=1)
    _fresh_T3000["a"] = a
    b = go(lazy_depth=3, **opts)                            # K = 32: the default; 10^6: refresh-free (2999 sweeps on end), one bank in place
    pa, pb = a[3]["paNt"][1, 1:], b[3]["paNt"][1, 1:]
    assert np.all(np.isfinite(pb))
    drift = np.max(np.abs(pa - pb), axis=1)                 # per time step
    print("max |paNt difference| over 3000 steps:", float(drift.max()), "at step", int(drift.argmax()) + 1)
    assert float(drift.max()) <= 2e-9, (float(drift.max()), int(drift.argmax()))
    np.testing.assert_array_equal(a[3]["ai"][:, 1:], b[3]["ai"][:, 1:])
    np.testing.assert_array_equal(a[3]["ak"], b[3]["ak"])
    assert ts.rel(b[0], a[0]) <= 1e-9 and ts.rel(b[2], a[2]) <= 1e-9
