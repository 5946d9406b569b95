"""GPU parity of the sparseFeatures = true branch (examples/slam-sparse-visual: per-particle EKF linearisation, NaN =
not observed; src/particleFilter.m:127-137,165-181, src/particleSmoother.m:194-217,267-277,306-321) against the oracle,
on synthetic cases and on the reference's own data file curve-x2.mat."""
import os

import numpy as np
import pytest

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def run_filter(rbpf, c):
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    return rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                               c["N_P"], c["dt"], True, rng=cases.device_rng(rbpf, c), extras=True)


def check_filter(ref, out):
    traj_max, traj_mean, xl_max, xl_mean, P_max, P_mean, traj_sample, xn_traj, ex = out
    tr = ref["trace"]
    np.testing.assert_array_equal(ex["ai"][1:], tr["ai"][1:])                  # bit-exact resample indices
    assert ex["iw_max"] == ref["iw_max"]
    assert np.max(np.abs(ex["logw"] - tr["logw"])) <= RTOL * max(1.0, np.max(np.abs(tr["logw"])))
    assert rel(ex["w"], tr["w"]) <= RTOL
    for got, key in ((traj_max, "traj_max"), (traj_mean, "traj_mean"), (xl_max, "xl_max"), (xl_mean, "xl_mean"),
                     (P_max, "P_max"), (P_mean, "P_mean"), (traj_sample, "traj_sample_iwmax"), (xn_traj, "xn_traj")):
        assert rel(got, ref[key]) <= RTOL, key
    assert rel(ex["xl"], tr["xl"]) <= RTOL and rel(ex["P"], tr["P"]) <= RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("N_P,N_T,nLand", [(12, 10, 6), (40, 14, 20), (7, 9, 3)])
def test_sparse_filter_matches_oracle(rbpf, N_P, N_T, nLand):
    c = cases.sparse_case(N_P, N_T, nLand, seed=5)
    assert np.all(np.isnan(c["y"][N_T // 2]))                                 # one step without any observation
    check_filter(cases.oracle_filter(c), run_filter(rbpf, c))


@pytest.mark.gpu
def test_sparse_filter_on_reference_data_file(rbpf):
    c = cases.sparse_curve_case(rbpf, 30, 40)
    check_filter(cases.oracle_filter(c), run_filter(rbpf, c))


def run_smoother(rbpf, c):
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    return rbpf.particleSmoother(mdl.dynModel, mdl.measModel, [], c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                                 c["N_P"], c["N_K"], c["dt"], True, rng=cases.device_rng(rbpf, c), extras=True)


def check_smoother(ref, out, N_K):
    XNK, XLK, PK, ex = out
    tr = ref["trace"]
    np.testing.assert_array_equal(ex["ak"], tr["ak"])
    np.testing.assert_array_equal(ex["ai"][:, 1:], tr["ai"][:, 1:])            # incl. the ancestor of the reference slot
    assert rel(ex["w"], tr["w"]) <= RTOL
    for k in range(1, N_K):
        a, b = ex["paNt"][k, 1:], tr["paNt"][k, 1:]
        assert np.max(np.abs(a - b)) <= RTOL * max(1.0, np.max(np.abs(b)))
    assert rel(XNK, ref["XNK"]) <= RTOL and rel(XLK, ref["XLK"]) <= RTOL and rel(PK, ref["PK"]) <= RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("N_P,N_T,nLand", [(8, 9, 5), (10, 12, 20)])
def test_sparse_smoother_matches_oracle(rbpf, N_P, N_T, nLand):
    c = cases.sparse_case(N_P, N_T, nLand, seed=6, N_K=3)
    check_smoother(cases.oracle_smoother(c, False), run_smoother(rbpf, c), 3)


@pytest.mark.gpu
def test_sparse_smoother_on_reference_data_file(rbpf):
    c = cases.sparse_curve_case(rbpf, 10, 30, N_K=3)                           # psslam.m: N_P = 10
    check_smoother(cases.oracle_smoother(c, False), run_smoother(rbpf, c), 3)


@pytest.mark.gpu
def test_sparse_flag_must_match_the_family(rbpf):
    c = cases.sparse_case(6, 5, 4, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError):                                        # dense calling convention on a sparse model
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 6, 1.0, False)
    with pytest.raises(rbpf.RBPFError, match="dense features"):                # particleSmootherInformationForm.m:77-80
        rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, [], c["odometry"], c["y"], c["x0_nonLin"], x0,
                                             P0, c["Q"], R, 6, 2, 1.0, True)


# ---- CPU: anchors of the oracle's sparse branch ---------------------------------------------------------------------
def test_pinhole_jacobian_matches_finite_differences(oracle):
    """measurement.m:61-81 (analytic derivatives w.r.t. the landmarks) against central differences of the projection
    measurement.m:49-52 -- an anchor that does not depend on this repository's reading of the derivative lines."""
    rs = np.random.RandomState(0)
    m = oracle.SparseVisualModel(nLand=7)
    for _ in range(5):
        xn = np.array([rs.uniform(-1, 1), rs.uniform(-1, 1), rs.uniform(-0.4, 0.4)])
        xl = np.vstack((rs.uniform(-2, 2, 7), rs.uniform(3, 7, 7))).T.reshape(-1)
        y, dy = m.measModel(xn, xl)
        J = np.zeros_like(dy)
        for j in range(xl.size):
            e = np.zeros(xl.size)
            e[j] = 1e-6
            J[:, j] = (m.measModel(xn, xl + e)[0] - m.measModel(xn, xl - e)[0]) / 2e-6
        np.testing.assert_allclose(dy, J, rtol=1e-6, atol=1e-8)


def test_sparse_update_with_no_observation_is_identity(oracle):
    """ind = ~isnan(yt) all false (particleFilter.m:134-136): empty innovation, log-weight 0, state untouched."""
    c = cases.sparse_case(5, 6, 4, seed=2)
    r = cases.oracle_filter(c)
    t = 6 // 2
    assert np.all(np.isnan(c["y"][t]))
    np.testing.assert_allclose(r["trace"]["w"][t], 1.0 / 5, rtol=1e-15)


def test_oracle_reproduces_sparse_golden(oracle):
    g = np.load(os.path.join(HERE, "golden", "sparse_curve_n40.npz"))
    c = cases.sparse_curve_case(None, int(g["N_P"]), int(g["N_T"]), N_K=int(g["N_K"]))
    np.testing.assert_array_equal(c["y"], g["y"])                              # loader + data file reproduce the inputs
    np.testing.assert_array_equal(c["odometry"], g["odometry"])
    r = cases.oracle_filter(c)
    np.testing.assert_array_equal(r["trace"]["ai"], g["filter_ai"])
    np.testing.assert_allclose(r["traj_mean"], g["filter_traj_mean"], rtol=1e-12)
    np.testing.assert_allclose(r["xl_mean"], g["filter_xl_mean"], rtol=1e-10, atol=1e-12)
    s = cases.oracle_smoother(c, False)
    np.testing.assert_array_equal(s["trace"]["ak"], g["smoother_ak"])
    np.testing.assert_allclose(s["XNK"], g["smoother_XNK"], rtol=1e-12)


@pytest.mark.gpu
def test_hip_reproduces_sparse_golden(rbpf):
    g = np.load(os.path.join(HERE, "golden", "sparse_curve_n40.npz"))
    c = cases.sparse_curve_case(rbpf, int(g["N_P"]), int(g["N_T"]), N_K=int(g["N_K"]))
    out = run_filter(rbpf, c)
    np.testing.assert_array_equal(out[8]["ai"][1:], g["filter_ai"][1:])
    assert rel(out[1], g["filter_traj_mean"]) <= RTOL and rel(out[3], g["filter_xl_mean"]) <= RTOL
    XNK, XLK, PK, ex = run_smoother(rbpf, c)
    np.testing.assert_array_equal(ex["ak"], g["smoother_ak"])
    assert rel(XNK, g["smoother_XNK"]) <= RTOL and rel(XLK, g["smoother_XLK"]) <= RTOL
