"""The C-ABI library loads on a CPU-only machine, exports every symbol include/rbpf.h declares, and
the product path fails loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rbpf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rbpf_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_match_export_list(rbpf):
    assert declared_functions() == sorted(rbpf.EXPORTS)


def test_library_exports_every_declared_symbol(rbpf):
    lib = rbpf.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.rbpf_abi_version() == 4
    assert lib.rbpf_status_string(0) == b"ok"
    assert b"positive definite" in lib.rbpf_status_string(rbpf.RBPF_ERR_CHOL_FAILED)


def test_no_cpu_fallback_without_a_device(rbpf):
    if rbpf.device_count() > 0:
        pytest.skip("a GPU is visible")
    c = cases.radio_case(4, 3, 8, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError) as ei:
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                            c["N_P"], c["dt"], rng=cases.device_rng(rbpf, c))
    assert ei.value.status == rbpf.RBPF_ERR_NO_DEVICE
    with pytest.raises(rbpf.RBPFError):
        rbpf.sample(np.ones(4) / 4, [0.3])


def test_arbitrary_callables_are_not_emulated_on_the_cpu(rbpf):
    """Unrecognised handles take the generic family (host evaluates the handles, device does the rest of the step):
    without a device that is an error like every other entry point, never a CPU emulation."""
    if rbpf.device_count() > 0:
        pytest.skip("a GPU is visible: covered by test_generic_host_callback_path_matches_oracle")
    c = cases.radio_case(4, 3, 8, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError) as ei:
        rbpf.particleFilter(lambda xn, dx, dt, Q: xn, lambda xn: np.zeros((xn.shape[1], 8)), c["odometry"], c["y"],
                            c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"])
    assert ei.value.status == rbpf.RBPF_ERR_NO_DEVICE


def test_host_side_basis_selection_matches_oracle(rbpf, oracle):
    for m, d, LL in [(16, 3, [[-11.8, -7.4, -2.4], [11.8, 7.4, 2.4]]), (256, 3, [[-12.02, -12.02, -2.4], [12.02, 12.02, 2.4]]),
                     (128, 2, [[-1.0, -2.5], [1.0, 2.5]]), (40, 2, [[-3.0, -3.0], [3.0, 3.0]])]:
        L1, NN1 = rbpf.domain_cartesian_dx(m, d, LL)
        L2, NN2 = oracle.domain_cartesian_dx(m, d, np.array(LL))
        np.testing.assert_array_equal(NN1, NN2.astype(np.int32))
        np.testing.assert_allclose(L1, L2, rtol=0)
        np.testing.assert_allclose(rbpf.eigenval(NN1, L1), oracle.eigenval(NN2, L2), rtol=1e-15)


def test_priors_match_oracle(rbpf, oracle):
    LL = np.array([[-12.0, -11.0, -2.4], [12.0, 11.0, 2.4]])
    _, x0a, P0a, Ra = rbpf.dense_mag_prior(64, LL, cases.THETA_MAG)
    _, x0b, P0b, Rb = oracle.dense_mag_prior(64, LL, cases.THETA_MAG)
    np.testing.assert_allclose(P0a, P0b, rtol=1e-14)
    np.testing.assert_array_equal(Ra, Rb)
    LL2 = np.array([[-1.0, -2.5], [1.0, 2.5]])
    _, _, P0a, Ra = rbpf.dense_radio_prior(32, LL2, cases.THETA_RADIO)
    _, _, P0b, Rb = oracle.dense_radio_prior(32, LL2, cases.THETA_RADIO)
    np.testing.assert_allclose(P0a, P0b, rtol=1e-14)
    np.testing.assert_array_equal(Ra, Rb)


def test_product_datagen_matches_oracle_generator(rbpf, oracle):
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    a = dg.bean_6D(30, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=5, m_sim=80)
    b = oracle.generate_bean_6D(30, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=5, m_sim=80)
    for k in ("dx", "initState", "y", "LL"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("traj,N_T", [("line_3D", 32), ("square_3D", 48)])
def test_product_radio_datagen_matches_oracle_generator(rbpf, oracle, traj, N_T):
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    Q = dg.radio_Q(N_T, traj)
    if traj == "line_3D":
        np.testing.assert_array_equal(Q, cases.radio_case(2, N_T, 8)["Q"])
    a = dg.planar_heading(N_T, Q, cases.THETA_RADIO, 1.0, seed=7, m_sim=90, nLL=4, traj=traj)
    b = oracle.generate_line_3D(N_T, Q, cases.THETA_RADIO, 1.0, seed=7, m_sim=90, nLL=4, traj=traj)
    for k in ("dx", "initState", "y", "LL"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-11, atol=1e-12)


def test_replay_rng_shape_validation(rbpf):
    c = cases.radio_case(4, 3, 8, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    bad = rbpf.ReplayRNG(np.zeros((1, 2, 3)), np.zeros((1, 2, 3, 1)))
    with pytest.raises(ValueError):
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                            c["N_P"], c["dt"], rng=bad)


def test_procrustes_recovers_a_similarity_transform(rbpf):
    """metrics.procrustes (MATLAB `procrustes`, used by calc_rmses.m:37) recovers scale, rotation and translation."""
    import importlib
    mt = importlib.import_module(rbpf.__name__ + ".metrics")
    rs = np.random.RandomState(0)
    Y = rs.standard_normal((20, 2))
    th = 0.7
    T = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    X = 1.7 * Y @ T + np.array([3.0, -2.0])
    d, Z, tr = mt.procrustes(X, Y)
    assert d < 1e-12 and abs(tr["b"] - 1.7) < 1e-12
    np.testing.assert_allclose(Z, X, atol=1e-12)
    np.testing.assert_allclose(tr["T"], T, atol=1e-12)
    rp, rm = mt.calc_rmses(X, Y, X[:5], Y[:5])
    assert rp < 1e-12 and rm < 1e-12
