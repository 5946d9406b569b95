"""GPU parity: HIP particleFilter (through the C ABI) vs the numpy oracle on the same seeded inputs.

Tolerances: ancestor indices bit-exact; fp64 states / weights / covariances within 1e-9 relative
(BASELINE.json north_star)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def run_both(rbpf, c, lazy_depth=0, inplace=0):
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    np.testing.assert_array_equal(mdl.NN, c["model"].NN.astype(np.int32))
    np.testing.assert_allclose(P0, c["P0_lin"], rtol=1e-14)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                              c["N_P"], c["dt"], rng=cases.device_rng(rbpf, c), extras=True, lazy_depth=lazy_depth,
                              inplace=inplace)
    return ref, out


def check_filter(ref, out):
    traj_max, traj_mean, xl_max, xl_mean, P_max, P_mean, traj_sample, xn_traj, ex = out
    tr = ref["trace"]
    np.testing.assert_array_equal(ex["ai"][1:], tr["ai"][1:])                  # bit-exact resample indices
    assert ex["iw_max"] == ref["iw_max"]
    # log-weights: compare after removing the particle-independent offset scale
    assert np.max(np.abs(ex["logw"] - tr["logw"])) <= RTOL * max(1.0, np.max(np.abs(tr["logw"])))
    assert rel(ex["w"], tr["w"]) <= RTOL
    assert rel(traj_max, ref["traj_max"]) <= RTOL
    assert rel(traj_mean, ref["traj_mean"]) <= RTOL
    assert rel(xl_max, ref["xl_max"]) <= RTOL
    assert rel(xl_mean, ref["xl_mean"]) <= RTOL
    assert rel(P_max, ref["P_max"]) <= RTOL
    assert rel(P_mean, ref["P_mean"]) <= RTOL                                   # incl. quirk Q3
    assert rel(traj_sample, ref["traj_sample_iwmax"]) <= RTOL
    assert rel(xn_traj, ref["xn_traj"]) <= RTOL
    assert rel(ex["xl"], tr["xl"]) <= RTOL
    assert rel(ex["P"], tr["P"]) <= RTOL
    assert rel(ex["xn"], tr["xn"]) <= RTOL


@pytest.mark.parametrize("N_P,N_T,m", [(8, 6, 16),      # n = 19  : border-only layout (mc = 0)
                                         (12, 8, 125),    # n = 128 : one core chunk, no border
                                         (9, 7, 130),     # n = 133 : core + 5 border rows
                                         (8, 6, 256)])    # n = 259 : the C2 layout (2 chunks + 3 border)
def test_dense_mag_filter_matches_oracle(rbpf, N_P, N_T, m):
    c = cases.mag_case(N_P, N_T, m, seed=3)
    ref, out = run_both(rbpf, c)
    check_filter(ref, out)


@pytest.mark.parametrize("N_P,N_T,m", [(10, 12, 32), (16, 10, 128), (7, 9, 140)])
def test_dense_radio_filter_matches_oracle(rbpf, N_P, N_T, m):
    c = cases.radio_case(N_P, N_T, m, seed=5)
    ref, out = run_both(rbpf, c)
    check_filter(ref, out)


@pytest.mark.parametrize("lazy_depth", [2, 3, 4])
@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 12, 11, 125), ("mag", 9, 10, 130), ("mag", 8, 9, 256),
                                            ("radio", 16, 12, 128), ("radio", 7, 11, 140)])
def test_multi_step_lazy_update_matches_oracle(rbpf, kind, N_P, N_T, m, lazy_depth):
    """lazy_depth = C: the stored covariances are rewritten every C-th step only, with up to C pending rank-ny
    downdates applied on the fly (same algebra as particleFilter.m:198 every step, so same results to rounding)."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=17)
    ref, out = run_both(rbpf, c, lazy_depth=lazy_depth)
    check_filter(ref, out)


@pytest.mark.parametrize("lazy_depth", [2, 3, 4])
@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 24, 14, 125), ("mag", 300, 9, 130), ("radio", 33, 13, 128)])
def test_single_bank_inplace_flush_is_bit_identical(rbpf, kind, N_P, N_T, m, lazy_depth):
    """inplace=1: ONE covariance bank; at a flush the siblings of every stored matrix are written to dead entries
    first and its first child then overwrites it (two launches of the step kernel).  Only the placement of the
    matrices in the bank changes, so every output equals the ping-pong run bit for bit (and the oracle to 1e-9)."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=23)
    ref, out1 = run_both(rbpf, c, lazy_depth=lazy_depth, inplace=1)
    check_filter(ref, out1)
    _, out0 = run_both(rbpf, c, lazy_depth=lazy_depth, inplace=-1)
    for a, b in zip(out1[:8], out0[:8]):
        np.testing.assert_array_equal(a, b)
    for k in ("ai", "logw", "w", "xl", "P", "xn"):
        np.testing.assert_array_equal(out1[8][k], out0[8][k])


@pytest.mark.parametrize("lazy_depth,inplace", [(0, -1), (3, -1), (3, 1)])
@pytest.mark.parametrize("N_P,N_T,m", [(24, 10, 125), (10, 8, 256)])
def test_fp32_storage_of_the_covariance_banks(rbpf, N_P, N_T, m, lazy_depth, inplace):
    """storage="fp32" (BASELINE.json configs[4]): the covariance banks hold float, all arithmetic stays fp64.  Every
    stored element carries a 6e-8 relative rounding per rewrite, so the run agrees with the fp64 oracle to ~1e-5, not to
    1e-9: tolerance 2e-5 on normalised weights / states / covariances over these few steps, and -- for these seeds --
    the same resampling indices (a weight perturbation of 1e-6 can move a draw that lands that close to a bin edge)."""
    c = cases.mag_case(N_P, N_T, m, seed=41)
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                              c["N_P"], c["dt"], rng=cases.device_rng(rbpf, c), extras=True, lazy_depth=lazy_depth,
                              inplace=inplace, storage="fp32")
    ex, tr = out[8], ref["trace"]
    np.testing.assert_array_equal(ex["ai"][1:], tr["ai"][1:])
    TOL = 2e-5
    assert rel(ex["w"], tr["w"]) <= TOL
    assert rel(out[1], ref["traj_mean"]) <= TOL and rel(out[2], ref["xl_max"]) <= TOL and rel(out[4], ref["P_max"]) <= TOL
    assert rel(ex["xl"], tr["xl"]) <= TOL and rel(ex["P"], tr["P"]) <= TOL
    assert rel(ex["P"], tr["P"]) > 1e-12                                        # it really is a different storage precision


def test_fp32_storage_is_rejected_where_it_is_not_implemented(rbpf):
    c = cases.radio_case(6, 5, 16, seed=1, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError):                                         # dense-radio (ny = 1): fp64 only
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 6, 1.0,
                            rng=cases.device_rng(rbpf, c), storage="fp32")


@pytest.fixture(scope="module")
def reduced_c2(rbpf, tmp_path_factory):
    """BASELINE.md section 3 "reduced C2": slam-dense-mag N=1024, T=300, m=256, fp64, replayed random numbers, and the
    plain-C restatement's outputs on it (all host cores; computed once for the tests below)."""
    import importlib
    import oracle_c
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T, m = 1024, 300, 256
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    rs = np.random.RandomState(77)
    rng = rbpf.ReplayRNG(rs.random_sample((1, T - 1, N)), rs.standard_normal((1, T - 1, N, 6)))
    import bench                                                             # usable_cores(): affinity mask / cgroup quota
    lib = oracle_c.build(native_dir=str(tmp_path_factory.mktemp("oracle_native")))      # -O3 -march=native -fopenmp
    ref, _ = oracle_c.particle_filter(rbpf, mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng,
                                      n_threads=bench.usable_cores(), want_full=False, lib_path=lib)
    return dict(d=d, mdl=mdl, x0=x0, P0=P0, R=R, rng=rng, N=N, ref=ref)


@pytest.mark.parametrize("lazy_depth", [0, 3])
def test_reduced_c2_against_the_c_restatement(rbpf, reduced_c2, lazy_depth):
    """Too large for the numpy oracle, so the HIP filter is compared with the plain-C restatement (itself pinned to the
    numpy oracle, tests/test_oracle_c.py) on the same replayed random numbers: 300 steps x 1024 draws of identical
    resampling (any differing index would show up in every later trajectory summary) and states / maps to 1e-9."""
    c = reduced_c2
    d, mdl, ref = c["d"], c["mdl"], c["ref"]
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], c["x0"], c["P0"], cases.Q_MAG, c["R"],
                              c["N"], 0.01, rng=c["rng"], want_xn_traj=False, lazy_depth=lazy_depth)
    assert rel(out[0], ref["traj_max"]) <= RTOL and rel(out[1], ref["traj_mean"]) <= RTOL
    assert rel(out[2], ref["xl_max"]) <= RTOL and rel(out[3], ref["xl_mean"]) <= RTOL


@pytest.mark.parametrize("kind,N_P,N_T,m,lazy_depth", [("mag", 10, 8, 130, 0), ("mag", 12, 9, 125, 3), ("radio", 9, 7, 128, 0)])
def test_generic_host_callback_path_matches_oracle(rbpf, oracle, kind, N_P, N_T, m, lazy_depth):
    """Arbitrary dynModel / measModel callables (here: plain Python closures over the oracle's model objects, unknown
    to the library) run through the generic family: the host evaluates the handles every step, the device does weights,
    normalisation, resampling and the Kalman update.  Same answers as the oracle's particleFilter."""
    c = (cases.mag_case if kind == "mag" else cases.radio_case)(N_P, N_T, m, seed=19)
    ref = cases.oracle_filter(c)
    mdl, Z = c["model"], c["rng"].Z
    calls = {"n": 0}

    def dynModel(xn, dx, dt, Q):                                  # called for t = 1.., i = 0..N-1 in that order (:104-109)
        t, i = divmod(calls["n"], N_P)
        calls["n"] += 1
        return mdl.dynModel(xn, dx, dt, Q, Z[0, t, i])[0]

    def measModel(xn):
        return mdl.measModel(xn)

    out = rbpf.particleFilter(dynModel, measModel, c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"], c["Q"], c["R"],
                              N_P, c["dt"], rng=cases.device_rng(rbpf, c), extras=True, lazy_depth=lazy_depth)
    assert calls["n"] == (N_T - 1) * N_P
    check_filter(ref, out)


def test_filter_makePlots_hook_sees_every_step(rbpf):
    """particleFilter.m:215-217: makePlots(xn, xl_max, P_max, traj_max, yhattraj, xn_traj, traj_mean, xl, P) after every time
    step, driven by the library's on_step hook; the final call's arguments equal the filter's outputs."""
    c = cases.mag_case(9, 7, 20, seed=41)
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    calls = []

    def makePlots(xn, xl_max, P_max, traj_max, yhattraj, xn_traj, traj_mean, xl, P):
        calls.append(dict(xn=xn.copy(), xl_max=xl_max.copy(), P_max=P_max.copy(), traj_max=traj_max.copy(), xl=xl.copy(), P=P.copy(),
                          shapes=(yhattraj.shape, xn_traj.shape)))

    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                              False, makePlots, rng=cases.device_rng(rbpf, c), extras=True)
    check_filter(ref, out)
    assert len(calls) == 7
    assert calls[0]["shapes"] == ((3, 7), (7, 9, 7))
    last = calls[-1]
    assert rel(last["xl_max"], ref["xl_max"]) <= RTOL and rel(last["P_max"], ref["P_max"]) <= RTOL
    assert rel(last["xn"], ref["trace"]["xn"]) <= RTOL and rel(last["P"], ref["trace"]["P"]) <= RTOL
    for t, cl in enumerate(calls):                       # traj_max is NaN beyond the step just finished (particleFilter.m:92)
        assert np.all(np.isfinite(cl["traj_max"][:, :t + 1])) and np.all(np.isnan(cl["traj_max"][:, t + 1:]))
        assert rel(cl["traj_max"][:, :t + 1], ref["traj_max"][:, :t + 1]) <= RTOL

    def bad(*a):
        raise ZeroDivisionError("plot failed")
    with pytest.raises(ZeroDivisionError):
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                            False, bad, rng=cases.device_rng(rbpf, c))


def test_caller_driven_generic_api_equals_the_callback_form(rbpf):
    """rbpf_filter_ancestors / rbpf_filter_step_external (a binding that cannot hand callbacks over) against
    rbpf_callbacks inside rbpf_filter_advance: same numbers."""
    c = cases.radio_case(9, 7, 24, seed=19)
    mdl, Z = c["model"], c["rng"].Z

    def make():
        calls = {"n": 0}

        def dynModel(xn, dx, dt, Q):
            t, i = divmod(calls["n"], 9)
            calls["n"] += 1
            return mdl.dynModel(xn, dx, dt, Q, Z[0, t, i])[0]
        return dynModel
    meas = lambda xn: mdl.measModel(xn)                                   # noqa: E731
    a = rbpf.particleFilter(make(), meas, c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"], c["Q"], c["R"], 9, c["dt"],
                            rng=cases.device_rng(rbpf, c))
    b = rbpf.particle_filter_external(make(), meas, c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"], c["Q"], c["R"], 9,
                                      c["dt"], rng=cases.device_rng(rbpf, c))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[4], b[3])


def test_exception_inside_a_model_handle_surfaces(rbpf):
    c = cases.radio_case(6, 4, 16, seed=2)

    def dyn(xn, dx, dt, Q):
        raise FloatingPointError("handle failed")
    with pytest.raises(FloatingPointError):
        rbpf.particleFilter(dyn, lambda xn: c["model"].measModel(xn), c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"],
                            c["Q"], c["R"], 6, c["dt"], rng=cases.device_rng(rbpf, c))


@pytest.mark.parametrize("m", [381, 400, 438, 445, 500, 508, 650, 765, 900])
def test_every_row_chunk_count(rbpf, m):
    """Basis sizes between the benchmark's: the covariance stream splits into 128-row chunks, and how the four waves share them
    depends on their number -- three chunks (384 <= nLin < 512: whole columns per wave, CPL = 3), five to seven (one round of four
    plus a remainder chunk).  r04 found the three-chunk sizes WRONG once the per-column LDS records pass 72 KB (nLin >= 441; the
    information form from 384 on): the launch sized the LDS for the blocked plan and ran the plain kernel (step_use_blocked).
    Filter and both smoothers against the numpy oracle at every chunk count the benchmark sizes do not cover."""
    import test_gpu_smoother as ts
    c = cases.mag_case(6, 5, m, seed=31, N_K=2)
    ref, out = run_both(rbpf, c)
    check_filter(ref, out)
    for info_form in (True, False):
        ref, out = ts.run_both(rbpf, c, info_form=info_form)
        ts.check(ref, out, 2)
